"""Restatement of the torch_geometric 2.5.3 ops on the BioMedKG hot path (test infrastructure).

torch_geometric is pinned by the reference (``/root/reference/pyproject.toml:6``) but its source
is not under ``/root/reference`` and it is not installable here, so this follows the published
semantics summarised in SURVEY.md Appendix A.  PARITY UNPINNED for these ops (no reference test
or fixture covers them); see ``tests/test_oracle_known_answers.py`` for the hand-derived checks.

Reference call sites: ``biomedkg/model/encoder.py:22-40,138-143`` (RGCNConv / GCNConv),
``biomedkg/model/gcl.py:3-4,40-43,75-76`` (uniform, mask_feature, dropout_edge),
``biomedkg/kge_module.py:93`` (negative_sampling), ``biomedkg/factory.py:47`` (GAE).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import rand as _rand


# --------------------------------------------------------------------------- A.1 / A.2  GCNConv
def add_remaining_self_loops(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """A.1: drop every existing self loop, append exactly one (i,i) per node in id order."""
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index[:, mask], torch.stack([loop, loop])], dim=1)


def gcn_norm(edge_index: torch.Tensor, num_nodes: int, dtype=torch.float32):
    """A.2: symmetric normalisation with degree counted at the TARGET (``edge_index[1]``)."""
    ei = add_remaining_self_loops(edge_index, num_nodes)
    row, col = ei[0], ei[1]
    w = torch.ones(ei.size(1), dtype=dtype, device=ei.device)
    deg = torch.zeros(num_nodes, dtype=dtype, device=ei.device).scatter_add_(0, col, w)
    dis = deg.pow(-0.5)
    dis.masked_fill_(dis == float("inf"), 0.0)
    return ei, dis[row] * w * dis[col]


def gcn_conv(x, edge_index, weight, bias=None):
    """A.2: ``out[c] = sum_{e=(r->c)} w_e (x W^T)[r] + b``; ``weight`` is ``[out, in]``."""
    n = x.size(0)
    ei, w = gcn_norm(edge_index, n, x.dtype)
    xw = x @ weight.t()
    out = torch.zeros(n, weight.size(0), dtype=x.dtype, device=x.device)
    out.index_add_(0, ei[1], xw.index_select(0, ei[0]) * w.unsqueeze(1))
    if bias is not None:
        out = out + bias
    return out


# --------------------------------------------------------------------------- A.3  RGCNConv
def rgcn_conv(x, edge_index, edge_type, weight, root, bias):
    """A.3 (CPU / no pyg-lib path): per-relation scatter-mean then ``@ weight[r]``, r ascending.

    ``weight`` ``[R, in, out]``, ``root`` ``[in, out]``, ``bias`` ``[out]``.
    """
    n = x.size(0)
    out = torch.zeros(n, weight.size(2), dtype=x.dtype, device=x.device)
    for r in range(weight.size(0)):
        sel = edge_type == r
        src, dst = edge_index[0, sel], edge_index[1, sel]
        s = torch.zeros(n, x.size(1), dtype=x.dtype, device=x.device).index_add_(0, dst, x.index_select(0, src))
        cnt = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(
            0, dst, torch.ones(dst.numel(), dtype=x.dtype, device=x.device)
        )
        h = s / cnt.clamp(min=1).unsqueeze(1)
        out = out + h @ weight[r]
    out = out + x @ root
    if bias is not None:
        out = out + bias
    return out


# --------------------------------------------------------------------------- A.7  augmentations
def mask_feature(x, p: float = 0.5, mode: str = "all"):
    """A.7: zero each element w.p. ``p`` (no rescale); always active (training=True default)."""
    assert mode == "all"
    keep = _rand.current().feat_keep(x, p).to(x.device)
    return x.masked_fill(~keep, 0.0), keep


def dropout_edge(edge_index, p: float = 0.5):
    """A.7: keep each edge w.p. ``1-p``; always active."""
    keep = _rand.current().edge_keep(edge_index.size(1), p).to(edge_index.device)
    return edge_index[:, keep], keep


def dropout(x, p: float, training: bool = True):
    """``F.dropout`` restated over an explicit keep mask (``encoder.py:54-55,157-158``)."""
    if not training or p == 0.0:
        return x
    keep = _rand.current().drop_keep(x, p).to(x.device)
    return x * keep.to(x.dtype) * (1.0 / (1.0 - p))


def uniform_(size: int, tensor: torch.Tensor):
    """A.7 ``inits.uniform``: U(-1/sqrt(size), 1/sqrt(size))."""
    bound = 1.0 / math.sqrt(size)
    with torch.no_grad():
        tensor.uniform_(-bound, bound)


def glorot_(tensor: torch.Tensor):
    """PyG ``inits.glorot``: U(-a, a), a = sqrt(6 / (size(-2) + size(-1)))."""
    a = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
    with torch.no_grad():
        tensor.uniform_(-a, a)


# --------------------------------------------------------------------------- A.4  negative_sampling
def edge_index_to_vector(edge_index, num_nodes):
    """PyG 2.5.3 ``utils/_negative_sampling.py::edge_index_to_vector``, non-bipartite and not ``force_undirected``
    (the form ``negative_sampling(edge_index, num_neg_samples=k)`` takes for the call at ``biomedkg/kge_module.py:93``):
    self loops are removed, the column index is compacted past the diagonal, ``idx = row (N-1) + col``,
    ``population = N^2 - N``."""
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    row, col = row[mask], col[mask]              # copies: the caller's edge_index is not touched
    col = col - (row < col).to(col.dtype)
    return row * (num_nodes - 1) + col, num_nodes * num_nodes - num_nodes


def vector_to_edge_index(idx, num_nodes):
    """PyG 2.5.3 ``vector_to_edge_index`` (same form): ``row = idx // (N-1)``, ``col = idx % (N-1)``,
    ``col[row <= col] += 1`` -- the diagonal is skipped, so ``row != col`` always."""
    row = torch.div(idx, num_nodes - 1, rounding_mode="floor")
    col = idx % (num_nodes - 1)
    col = col + (row <= col).to(col.dtype)
    return torch.stack([row, col], dim=0)


def negative_sampling(edge_index, num_nodes=None, num_neg_samples=None):
    """PyG 2.5.3 ``negative_sampling`` with ``method="sparse"``, ``force_undirected=False``: uniform off-diagonal
    (h, t) pairs not among the positives, <= 3 over-sampling rounds of ``random.sample`` + ``np.isin``.
    (SURVEY Appendix A.4 described the pre-2.0 ``N*N`` population; this follows the pinned version.)"""

    def run():
        n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
        idx_t, population = edge_index_to_vector(edge_index.cpu(), n)
        idx = idx_t.numpy()
        if idx.size >= population:
            return edge_index.new_empty((2, 0))
        k = num_neg_samples if num_neg_samples is not None and num_neg_samples else edge_index.size(1)
        prob = 1.0 - idx.size / population         # probability that a draw is a negative
        sample_size = int(1.1 * k / prob)
        neg = None
        for _ in range(3):
            if population <= sample_size:            # PyG `sample`: the whole range when over-asked
                rnd = np.arange(population, dtype=np.int64)
            else:
                rnd = _rand.current().sample(population, sample_size).numpy()
            mask = np.isin(rnd, idx)
            if neg is not None:
                mask |= np.isin(rnd, neg)
            rnd = rnd[~mask]
            neg = rnd if neg is None else np.concatenate([neg, rnd])
            if neg.size >= k:
                neg = neg[:k]
                break
        return vector_to_edge_index(torch.from_numpy(neg), n).to(edge_index.device)

    return _rand.current().neg_edges(run)


# --------------------------------------------------------------------------- Modules (PyG names)
class GCNConv(torch.nn.Module):
    """A.2 defaults: add_self_loops, normalize, bias; ``lin.weight [out,in]`` glorot, bias zeros."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = torch.nn.Linear(in_channels, out_channels, bias=False)
        self.bias = torch.nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.lin.weight)
        torch.nn.init.zeros_(self.bias)

    def forward(self, x, edge_index):
        return gcn_conv(x, edge_index, self.lin.weight, self.bias)


class RGCNConv(torch.nn.Module):
    """A.3 defaults: aggr="mean", root_weight, bias, no bases/blocks."""

    def __init__(self, in_channels: int, out_channels: int, num_relations: int):
        super().__init__()
        self.in_channels, self.out_channels, self.num_relations = in_channels, out_channels, num_relations
        self.weight = torch.nn.Parameter(torch.empty(num_relations, in_channels, out_channels))
        self.root = torch.nn.Parameter(torch.empty(in_channels, out_channels))
        self.bias = torch.nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        glorot_(self.weight)
        glorot_(self.root)
        torch.nn.init.zeros_(self.bias)

    def forward(self, x, edge_index, edge_type):
        return rgcn_conv(x, edge_index, edge_type, self.weight, self.root, self.bias)


class RGATConv(torch.nn.Module):
    """Placeholder: the reference's RGAT wiring raises TypeError before use."""

    def __init__(self, in_channels, out_channels, num_relations, heads=1):
        super().__init__()
        raise NotImplementedError("RGATConv is out of scope (broken in the reference)")


class GAE(torch.nn.Module):
    """A.8: ``encode = encoder(...)``, ``decode = decoder(...)``."""

    def __init__(self, encoder, decoder=None):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder

    def encode(self, *args, **kwargs):
        return self.encoder(*args, **kwargs)

    def decode(self, *args, **kwargs):
        return self.decoder(*args, **kwargs)

    def forward(self, *args, **kwargs):
        return self.encoder(*args, **kwargs)


__all__ = [
    "add_remaining_self_loops", "gcn_norm", "gcn_conv", "rgcn_conv", "mask_feature", "dropout_edge",
    "dropout", "uniform_", "glorot_", "negative_sampling", "edge_index_to_vector", "vector_to_edge_index", "GCNConv", "RGCNConv", "RGATConv", "GAE", "F",
]
