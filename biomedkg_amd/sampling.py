"""Negative-edge sampling on the device.

Semantics follow PyG 2.5.3 ``negative_sampling(edge_index, num_neg_samples=k)`` as the reference
calls it (``biomedkg/kge_module.py:93``), i.e. ``edge_index_to_vector`` / ``vector_to_edge_index`` in their
non-bipartite, directed form: the population is the N(N-1) OFF-DIAGONAL cells of the grid (N = max id + 1),
self-loop positives are dropped, cell (r, c) has the code ``r (N-1) + c - (r < c)`` and the diagonal is re-opened
on the way out (``c += (r <= c)``) -- no negative is ever a self pair.  ``prob = 1 - E'/(N^2 - N)`` with E' the
loop-free positive count (duplicates counted, as ``idx.numel()`` does), at most 3 over-sampling rounds of
``int(1.1 k / prob)`` candidates, truncated to k (fewer may come back); ``k`` defaults to ``edge_index.size(1)``.
Upstream draws with python ``random.sample`` + ``np.isin`` on the host; here the candidates are a prefix of a keyed
random permutation of the population (distinct by construction), membership in the positives is a hash-set lookup
(table built once per graph) and the survivors are compacted in candidate order (``csrc/negsample.hip``): no sort,
no atomics, the result is a pure function of (graph, seed).  Graphs so dense that the whole population is
enumerated take the sort / binary-search form below (torch glue over the same codes).
"""
from __future__ import annotations

import torch

from . import _lib as L
from .graph import _cached


def _offdiag_code(row, col, n):
    """PyG ``edge_index_to_vector`` (non-bipartite, directed) on loop-free pairs."""
    return row * (n - 1) + col - (row < col).to(col.dtype)


def _positive_keys(edge_index, n):
    def build():
        row, col = edge_index[0], edge_index[1]
        keep = row != col
        return torch.unique(_offdiag_code(row[keep], col[keep], n))  # torch.unique returns them sorted
    return _cached(edge_index, "pos_keys", (int(n),), build)


def _loop_free_count(edge_index):
    """E' = number of positives with head != tail (duplicates counted): PyG's ``idx.numel()``."""
    return _cached(edge_index, "loop_free", (), lambda: int((edge_index[0] != edge_index[1]).sum().item()))


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


def _pow2_at_least(v):
    return 1 << max(10, int(v - 1).bit_length())


def _num_nodes_of(edge_index):
    return _cached(edge_index, "max_id", (), lambda: int(edge_index.max().item()) + 1 if edge_index.numel() else 0)


def _positive_table(edge_index, n):
    """Open-addressing hash set of the loop-free positives' off-diagonal codes (int64, empty = -1), cached per graph."""
    def build():
        ei = edge_index.contiguous()
        cap = _pow2_at_least(2 * ei.size(1))
        table = torch.full((cap,), -1, dtype=torch.int64, device=ei.device)
        L.call("bkg_hashset_insert_offdiag", L.ptr(ei[0]), L.ptr(ei[1]), n, ei.size(1), L.ptr(table), cap, L.stream())
        return table
    return _cached(edge_index, "pos_table", (int(n),), build)


_WATCH = {"host": None, "slots": [None] * 8, "next": 0}   # ONE pinned buffer, made once; (event, k) per slot in flight
SHORTFALLS = 0           # calls whose single round came up short (expected: never; ~1e-9 per call under the guard)


def _watch_shortfall(count, k):
    """The "sure" path never READS its survivor count inside the step (that would be a host round trip); the count is
    copied to pinned memory asynchronously and looked at by a LATER call, once its event has completed -- a shortfall
    (PyG would have gone on to rounds two and three; here the missing slots hold (0, 0) pairs) is reported, late but
    loudly, instead of passing silently.  Skipped while a stream is being captured.  The pinned buffer is allocated once
    (a hipHostMalloc per call showed up as a 60 ms stall inside a timed bench step)."""
    global SHORTFALLS
    if torch.cuda.is_current_stream_capturing():
        return
    w = _WATCH
    if w["host"] is None:
        w["host"] = torch.zeros(len(w["slots"]), dtype=torch.int32).pin_memory()
    for i, item in enumerate(w["slots"]):
        if item is not None and item[0].query():
            w["slots"][i] = None
            if int(w["host"][i]) < item[1]:
                SHORTFALLS += 1
                import warnings
                warnings.warn(f"negative_sampling: an earlier call kept {int(w['host'][i])} of {item[1]} negatives in its "
                              "single round (the rest are (0, 0) pairs); PyG would have drawn further rounds", RuntimeWarning)
    i = w["next"]
    if w["slots"][i] is None:                       # (all slots busy: this call goes unwatched)
        w["host"][i:i + 1].copy_(count, non_blocking=True)
        event = torch.cuda.Event()
        event.record()
        w["slots"][i] = (event, int(k))
        w["next"] = (i + 1) % len(w["slots"])


def _negative_sampling_hashed(edge_index, n, k, sample_size, prob, seed):
    population = n * (n - 1)
    dev = edge_index.device
    pos_table = _positive_table(edge_index, n)
    # PyG's first round keeps ~1.1 k of its int(1.1 k / prob) candidates (binomial, sigma^2 <= 1.1 k (1 - prob)).  When
    # that 10 % margin is worth six sigma the round cannot fall short of k in practice, and the survivor count need not
    # be READ: no host round trip in the middle of the step (zero-filled result: a shortfall -- ~1e-9 per call -- would
    # leave (0, 0) pairs, never stale memory).  Otherwise: count, and go on to rounds two and three like PyG.
    sure = sample_size <= population and 0.1 * k >= 6.0 * (1.1 * k * max(1.0 - prob, 0.0)) ** 0.5 and prob > 0.9
    out = (torch.zeros if sure else torch.empty)(2, k, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    have, first, draw = 0, 0, min(sample_size, population)
    if sure:
        keys = torch.empty(draw, dtype=torch.int64, device=dev)
        flag = torch.empty(2, draw, dtype=torch.int32, device=dev)
        ws = torch.empty(L.lib().bkg_negative_sample_workspace_bytes(draw), dtype=torch.uint8, device=dev)
        L.call("bkg_negative_sample", n, 0, draw, seed, L.ptr(pos_table), pos_table.numel(), L.ptr(keys),
               L.ptr(flag[0]), L.ptr(flag[1]), L.ptr(out[0]), L.ptr(out[1]), 0, k, L.ptr(count), L.ptr(ws), ws.numel(),
               L.stream())
        _watch_shortfall(count, k)
        return out
    for _ in range(3):                               # PyG: at most three over-sampling rounds
        keys = torch.empty(draw, dtype=torch.int64, device=dev)
        flag = torch.empty(2, draw, dtype=torch.int32, device=dev)
        ws = torch.empty(L.lib().bkg_negative_sample_workspace_bytes(draw), dtype=torch.uint8, device=dev)
        L.call("bkg_negative_sample", n, first, draw, seed, L.ptr(pos_table), pos_table.numel(), L.ptr(keys),
               L.ptr(flag[0]), L.ptr(flag[1]), L.ptr(out[0]), L.ptr(out[1]), have, k, L.ptr(count), L.ptr(ws), ws.numel(),
               L.stream())
        have = min(have + int(count.item()), k)      # the one host sync of the sampler
        first += draw
        if have >= k:
            return out
        draw = min(int(1.1 * (k - have) / prob) + 64, population - first)   # later rounds continue the permutation's prefix
        if draw <= 0:
            break
    return out[:, :have].contiguous()


def negative_sampling(edge_index, num_nodes=None, num_neg_samples=None, seed=None):
    L.require_cuda(edge_index)
    dev = edge_index.device
    n = _num_nodes_of(edge_index) if num_nodes is None else int(num_nodes)
    population = n * n - n                          # off-diagonal cells only
    e = _loop_free_count(edge_index)                # upstream counts duplicate positives too (idx.numel()), never self loops
    if e >= population:
        return edge_index.new_empty((2, 0))
    k = num_neg_samples if num_neg_samples else edge_index.size(1)
    prob = 1.0 - e / population
    sample_size = int(1.1 * k / prob)
    if population > sample_size and n < (1 << 31):
        if seed is None:
            seed = int(torch.randint(0, 1 << 62, (1,)).item())    # CPU generator: follows torch.manual_seed, no device sync
        return _negative_sampling_hashed(edge_index, n, k, sample_size, prob, seed)
    pos = _positive_keys(edge_index, n)
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
    row, col = neg // (n - 1), neg % (n - 1)         # vector_to_edge_index: re-open the diagonal
    return torch.stack([row, col + (row <= col).to(col.dtype)], dim=0)
