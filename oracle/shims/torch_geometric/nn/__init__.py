from oracle.pyg_ops import GAE, GCNConv, RGATConv, RGCNConv  # noqa: F401
from . import inits  # noqa: F401
