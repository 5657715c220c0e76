from oracle.pygcl_ops import DualBranchContrast, SingleBranchContrast  # noqa: F401
