"""Negative-edge sampling on the device.

Semantics follow PyG 2.5.3 ``negative_sampling(edge_index, num_neg_samples=k)`` as the reference
calls it (``biomedkg/kge_module.py:93``): uniform (head, tail) pairs from the
N x N grid (N = max id + 1) that are not positives, all distinct, at most 3 over-sampling rounds of
``int(1.1 k / (1 - E/N^2))`` candidates, truncated to k (fewer may come back).  Upstream draws with
python ``random.sample`` + ``np.isin`` on the host; here candidates come from the device generator
and the membership / distinctness tests are sort + binary search on the GPU.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .graph import _cached


def _positive_keys(edge_index):
    def build():
        n = int(edge_index.max().item()) + 1 if edge_index.numel() else 0
        keys = torch.unique(edge_index[0] * n + edge_index[1]) if n else edge_index.new_empty(0)
        return n, keys  # torch.unique returns them sorted
    return _cached(edge_index, "pos_keys", (), build)


def _distinct_first(cand):
    """Mask keeping the first occurrence of every value of ``cand`` (original order preserved)."""
    s, order = torch.sort(cand, stable=True)
    dup_sorted = torch.zeros_like(s, dtype=torch.bool)
    dup_sorted[1:] = s[1:] == s[:-1]
    keep = torch.ones_like(dup_sorted)
    keep[order] = ~dup_sorted
    return keep


def _isin_sorted(values, sorted_keys):
    if sorted_keys.numel() == 0:
        return torch.zeros_like(values, dtype=torch.bool)
    pos = torch.searchsorted(sorted_keys, values).clamp_(max=sorted_keys.numel() - 1)
    return sorted_keys[pos] == values


def negative_sampling(edge_index, num_nodes=None, num_neg_samples=None):
    L.require_cuda(edge_index)
    dev = edge_index.device
    n, pos = _positive_keys(edge_index)
    if num_nodes is not None:
        if num_nodes != n:
            n = int(num_nodes)
            pos = torch.unique(edge_index[0] * n + edge_index[1])
    population = n * n
    e = edge_index.size(1)  # upstream counts duplicate positives too (idx.numel())
    if e >= population:
        return edge_index.new_empty((2, 0))
    k = num_neg_samples if num_neg_samples else e
    prob = 1.0 - e / population
    sample_size = int(1.1 * k / prob)
    neg = None
    for _ in range(3):
        if population <= sample_size:
            rnd = torch.arange(population, device=dev)
        else:
            rnd = torch.randint(population, (sample_size,), device=dev)
            rnd = rnd[_distinct_first(rnd)]          # random.sample draws without replacement
        bad = _isin_sorted(rnd, pos)
        if neg is not None:
            bad |= _isin_sorted(rnd, torch.sort(neg).values)
        rnd = rnd[~bad]
        neg = rnd if neg is None else torch.cat([neg, rnd])
        if neg.numel() >= k:
            neg = neg[:k]
            break
    return torch.stack([neg // n, neg % n], dim=0)
