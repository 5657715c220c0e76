from oracle.pygcl_ops import JSD, InfoNCE  # noqa: F401
