from oracle.pyg_ops import glorot_ as glorot, uniform_ as uniform  # noqa: F401
