from .decoder import ComplEx, DistMult, TransE
from .encoder import RGAT, RGCN, GCNEncoder
from .gcl import DGI, GGD, GRACE

__all__ = ["TransE", "DistMult", "ComplEx", "RGAT", "RGCN", "GCNEncoder", "DGI", "GRACE", "GGD"]
