"""KGE decoders behind the reference's class names (``biomedkg/model/decoder.py``); the scoring and
its backward run in ``csrc/kge.hip``."""
from __future__ import annotations

import math

import torch

from .. import ops


class Decoder(torch.nn.Module):
    """``rel_emb [R, d]`` xavier-uniform (reference decoder.py:7-18)."""

    def __init__(self, num_relations, hidden_channels):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.rel_emb = torch.nn.Parameter(torch.empty(num_relations, hidden_channels))
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self.rel_emb)

    def forward(self, z, edge_index, edge_type):
        raise NotImplementedError


class TransE(Decoder):
    kind = ops.KGE_TRANSE

    def reset_parameters(self):
        """U(-6/sqrt(d), 6/sqrt(d)) then row L2-normalise (reference decoder.py:25-29)."""
        bound = 6.0 / math.sqrt(self.hidden_channels)
        with torch.no_grad():
            self.rel_emb.uniform_(-bound, bound)
            self.rel_emb.div_(self.rel_emb.norm(p=2, dim=-1, keepdim=True).clamp_min(1e-12))

    def forward(self, z, edge_index, edge_type):
        return ops.kge_score(z, self.rel_emb, edge_index, edge_type, ops.KGE_TRANSE)


class DistMult(Decoder):
    kind = ops.KGE_DISTMULT

    def forward(self, z, edge_index, edge_type):
        return ops.kge_score(z, self.rel_emb, edge_index, edge_type, ops.KGE_DISTMULT)


class ComplEx(Decoder):
    """Unsupported: the reference's ComplEx assigns a Parameter before ``Module.__init__`` and raises
    on construction (``decoder.py:52-58``).  Raises the same way here."""

    def __init__(self, num_relations, hidden_channels):
        raise AttributeError("cannot assign parameters before Module.__init__() call (ComplEx is broken in the "
                             "reference: decoder.py:52-58)")
