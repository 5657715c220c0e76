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
    # be READ: no host round trip in the middle of the step (the kernel zeroes the slots its survivors do not reach: a
    # shortfall -- ~1e-9 per call -- leaves (0, 0) pairs, never stale memory).  Otherwise: count, and go on to rounds two
    # and three like PyG.
    sure = sample_size <= population and 0.1 * k >= 6.0 * (1.1 * k * max(1.0 - prob, 0.0)) ** 0.5 and prob > 0.9
    out = torch.empty(2, k, dtype=torch.int64, device=dev)
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
        out._bkg_exchangeable = True                 # the first k survivors of a keyed permutation: any order is as likely
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
            out._bkg_exchangeable = True
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


# ----------------------------------------------------------------------------- node-partitioned training
class BlockNegativeSampler:
    """Negatives for a node-partitioned KGE step (biomedkg_amd/partition.py, both-owner scoring; csrc/negsample.hip
    ``bkg_negative_sample_blocks``).  The N x N cell grid is cut by the partition's node ranges into world x world blocks
    (head owner a, tail owner b).  Block (a, b) draws ``k_ab`` of the ``k`` negatives -- proportional to its FREE cells
    (off-diagonal cells that are not positives; counted here, once per graph, from the edge list every rank holds), so
    the union over the blocks is k distinct non-edges, uniform over the free cells, stratified by block with proportional
    allocation (PyG draws a simple random sample: same marginals) -- from its own keyed permutation seeded by
    ``(seed, a, b)``.  Rank r draws the blocks of its row band (it owns the heads: these triplets count in its loss) and
    of its column band (tails local, heads remote: scored for the gradient of its own rows only); the other owner of a
    block draws the very same triplets, so the scorer's backward needs no exchange at all.

    Relation ids (the reference's ``edge_type.repeat(neg_ratio)[randperm(k)]``, kge_module.py:97-100): the multiset
    ``edge_type.repeat(neg_ratio)`` is dealt to the blocks ONCE, by a fixed low-discrepancy bijection of the k positions
    (global position p holds the ``(p * stride) mod k``-th id of the relation-sorted list, stride ~ 0.618 k coprime to k:
    every block's contiguous range samples all relations in proportion), and laid out relation-grouped inside a block.
    The negatives a block pairs them with arrive in exchangeable order (the first survivors of a keyed permutation of the
    block's cells), so this fixed arrangement IS a uniformly random pairing within the block -- the argument of
    ``ops.grouped_types`` block by block -- the exact multiset over all blocks, and every rank derives the same table.
    What it buys: the ids, their relation order and the scorer's wave tables are constants of the (partition, k) pair --
    no per-step permutation, no per-step sort.

    ``draw(seed)`` -> ``index [2, k_r]`` (node ids, or table SLOTS when the sampler was given ``n_max``) in the fixed
    layout [block (r, r) | blocks (r, b != r) | blocks (a != r, r)]; ``types`` / ``types_own`` / ``types_foreign``: the
    static ids of that layout; ``entries``: positions in ``cat(head, tail)`` of the incidences with a local node (known
    by construction: no mask, no count read inside the step).  All ranks must pass the same seed."""

    def __init__(self, edge_index, edge_type, bounds, rank, k, num_relations=None, n_max=None):
        L.require_cuda(edge_index, edge_type)
        dev = edge_index.device
        world = len(bounds) - 1
        n = int(bounds[-1])
        self.n, self.k, self.rank, self.world = n, int(k), int(rank), world
        e = int(edge_type.numel())
        if self.k <= 0 or e == 0 or self.k % e:
            raise ValueError("k must be a positive multiple of the positives' count (neg_ratio * E)")
        self.table = _positive_table(edge_index, n)
        sizes = [bounds[i + 1] - bounds[i] for i in range(world)]
        # free cells per block: population minus the distinct loop-free positives inside it
        bt = torch.tensor(bounds[1:-1], device=dev, dtype=torch.int64)
        keys = _positive_keys(edge_index, n)                                  # distinct off-diagonal codes, sorted
        row = keys // (n - 1)
        col = keys % (n - 1)
        col = col + (row <= col).to(col.dtype)
        blk = torch.bucketize(row, bt, right=True) * world + torch.bucketize(col, bt, right=True)
        used = torch.bincount(blk, minlength=world * world).tolist()          # (one host read, once per graph)
        pop = [sizes[a] * (sizes[b] - 1 if a == b else sizes[b]) for a in range(world) for b in range(world)]
        free = [p - u for p, u in zip(pop, used)]
        total_free = sum(free)
        if self.k > total_free:
            raise ValueError(f"{self.k} negatives asked of a graph with {total_free} non-edges")
        # proportional allocation, largest remainders (every rank computes the same table)
        quota = [self.k * f // total_free for f in free]
        rem = sorted(range(world * world), key=lambda i: (-(self.k * free[i] % total_free), i))
        for i in rem[: self.k - sum(quota)]:
            quota[i] += 1
        gpos, acc = [], 0
        for q in quota:
            gpos.append(acc)
            acc += q
        r = self.rank
        order = [r * world + r] + [r * world + b for b in range(world) if b != r] + [a * world + r for a in range(world) if a != r]
        rows, cand0, out0 = [], 0, 0
        self.n_own = 0
        layout = []                                           # (block id, out0, k) of the rank's blocks, in output order
        shift = (lambda owner: owner * int(n_max) - bounds[owner]) if (n_max is not None and world > 1) else (lambda owner: 0)
        for bid in order:
            a, b = divmod(bid, world)
            kq = quota[bid]
            if kq == 0:
                layout.append((bid, out0, 0))
                continue
            p = free[bid] / pop[bid]
            # candidates: enough that fewer than kq survivors is a six-sigma event (hypergeometric, variance below the
            # binomial's); the whole block when that is most of it anyway
            m = int((kq + 6.0 * (kq * max(1.0 - p, 0.0)) ** 0.5 + 8.0) / p) + 1
            m = min(max(m, kq), pop[bid])
            rows.append([bounds[a], sizes[a], bounds[b], sizes[b], pop[bid], cand0, m, out0, kq, gpos[bid], bid, shift(a), shift(b)])
            layout.append((bid, out0, kq))
            cand0 += -(-m // 256) * 256
            out0 += kq
            if a == r:
                self.n_own += kq
        self.layout, self.k_rank, self.n_cand = layout, out0, cand0
        self.n_blocks = len(rows)
        self.slots = n_max is not None
        assert L.lib().bkg_negative_blocks_struct_bytes() == 13 * 8
        self.blocks = torch.tensor(rows if rows else [[0] * 13], dtype=torch.int64, device=dev)
        self.ws_bytes = L.lib().bkg_negative_sample_workspace_bytes(max(cand0, 1))
        # incidences with a local node, by construction: heads of the own blocks, tails of block (r, r), tails of the
        # foreign blocks -- positions in cat(head [k_r], tail [k_r])
        k_rr = quota[r * world + r]
        ar = lambda lo, hi: torch.arange(lo, hi, device=dev, dtype=torch.int64)          # noqa: E731
        self.entries = torch.cat([ar(0, self.n_own), self.k_rank + ar(0, k_rr), self.k_rank + ar(self.n_own, self.k_rank)])
        self.count = None
        # ---- the relation ids of the rank's blocks: constants of (partition, k)
        et = edge_type.contiguous()
        nrel = int(num_relations) if num_relations is not None else int(et.max().item()) + 1
        self.num_relations = nrel
        sorted_ids = torch.sort(et).values                                   # the relation-sorted positives' ids ...
        ratio = self.k // e                                                  # ... each standing for `ratio` consecutive entries
        stride = int(self.k * 0.6180339887498949) | 1
        from math import gcd
        while gcd(stride, self.k) != 1:
            stride += 2
        parts = []
        for bid, o0, kq in layout:
            if kq == 0:
                continue
            p = gpos[bid] + ar(0, kq)
            parts.append(torch.sort(sorted_ids[((p * stride) % self.k) // ratio]).values)
        self.types = torch.cat(parts) if parts else et.new_empty(0)
        self.types_own = self.types[: self.n_own].clone()
        self.types_foreign = self.types[self.n_own:].clone()
        for t in (self.types_own, self.types_foreign):                       # their relation order, computed once
            if t.numel():
                perm = torch.argsort(t, stable=True).to(torch.int32)
                ptr = torch.searchsorted(t[perm.long()].contiguous(), torch.arange(nrel + 1, device=dev)).to(torch.int32)
                t._bkg_cache = {"rel_order": ((t._version, nrel), (perm, ptr))}

    def draw(self, seed):
        dev = self.blocks.device
        kr = self.k_rank
        out = torch.empty(2, kr, dtype=torch.int64, device=dev)
        self.count = torch.zeros(max(self.n_blocks, 1), dtype=torch.int32, device=dev)
        if self.n_blocks and self.n_cand:
            keys = torch.empty(self.n_cand, dtype=torch.int64, device=dev)
            flag = torch.empty(2, self.n_cand, dtype=torch.int32, device=dev)
            ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
            L.call("bkg_negative_sample_blocks", L.ptr(self.blocks), self.n_blocks, self.n_cand, self.n, int(seed) & ((1 << 64) - 1),
                   self.k, L.ptr(self.table), self.table.numel(), None, 0, L.ptr(keys),
                   L.ptr(flag[0]), L.ptr(flag[1]), L.ptr(out[0]), L.ptr(out[1]), None, L.ptr(self.count), L.ptr(ws),
                   ws.numel(), L.stream())
        return out

    def shortfall(self):
        """Negatives the last draw's blocks fell short by (a host read: tests / diagnostics only)."""
        if self.count is None or not self.n_blocks:
            return 0
        want = self.blocks[:, 8].to(torch.int32)
        return int((want - self.count[: self.n_blocks]).clamp_min(0).sum().item())
