"""oracle/samplers.py -- the CPU restatement of the reference's two loaders (NeighborLoader([30] * 3),
GraphSAINTRandomWalkSampler(walk_length=10): /root/reference/biomedkg/data_module.py:89-107) -- against hand-derived known
answers and brute-force checks.  The reference has no tests for its loaders and PyG is not installable here: the
restatement follows the published pyg-lib / torch-sparse algorithms ("parity unpinned", see the module's header); what is
pinned here is that it IS those algorithms.  The HIP samplers are held to it draw for draw in tests/test_hip_samplers.py."""
import itertools
import math

import torch

from oracle import samplers as S


def _word_for(t, n):
    """A 32-bit word that the multiply-shift maps to t in [0, n)."""
    w = -(-t * (1 << 32) // n)
    assert S._below(w, n) == t
    return w


def _as_i32(words):
    return torch.tensor([[w - (1 << 32) if w >= (1 << 31) else w for w in row] for row in words], dtype=torch.int32)


def test_below_is_the_multiply_shift_on_the_raw_word_whatever_the_storage_sign():
    for word in (0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0xDEADBEEF):
        signed = word - (1 << 32) if word >= (1 << 31) else word
        for n in (1, 2, 3, 30, 31, 2999, (1 << 31) - 1):
            assert S._below(signed, n) == (word * n) >> 32 == S._below(word, n) < n


def test_floyd_known_answer_and_exact_uniformity():
    # deg 4, k 2: steps j = 2, 3.  t0 = 1 -> take 1; t1 = 1 is taken -> take j = 3
    assert S.floyd_positions(4, 2, [_word_for(1, 3), _word_for(1, 4)]) == [1, 3]
    assert S.floyd_positions(4, 2, [_word_for(2, 3), _word_for(0, 4)]) == [2, 0]
    # every k-subset of [0, deg) comes out of the same number of equally likely draw sequences
    for deg, k in ((5, 2), (6, 3), (4, 4)):
        ranges = [range(j + 1) for j in range(deg - k, deg)]
        counts = {}
        for ts in itertools.product(*ranges):
            words = [_word_for(t, j + 1) for t, j in zip(ts, range(deg - k, deg))]
            pos = S.floyd_positions(deg, k, words)
            assert len(set(pos)) == k and all(0 <= p < deg for p in pos)
            counts[frozenset(pos)] = counts.get(frozenset(pos), 0) + 1
        assert len(counts) == math.comb(deg, k) and len(set(counts.values())) == 1


def test_neighbor_sample_known_answer():
    # in-edges of 0: from 1, 2, 3, 4 (edge ids 0..3); of 2: from 5 (edge 4) and 0 (edge 5); of 4: from 4 (a self loop, edge 6)
    ei = torch.tensor([[1, 2, 3, 4, 5, 0, 4], [0, 0, 0, 0, 2, 2, 4]])
    tape = torch.zeros(7, S.TAPE_STRIDE, dtype=torch.int32)
    tape[0, :2] = _as_i32([[_word_for(1, 3), _word_for(1, 4)]])[0]        # node 0, fan-out 2 of 4: positions [1, 3]
    out = S.neighbor_sample(ei, 7, torch.tensor([0]), [2, 2], tape)
    # hop 1: edges (2 -> 0), (4 -> 0); new nodes 2, 4 in discovery order.  hop 2: node 2 takes both in-edges (5, 0 -- 0 is
    # known), node 4 its self loop; new node 5
    assert out["n_id"].tolist() == [0, 2, 4, 5]
    assert out["hops"] == [[0], [2, 4], [5]]
    assert out["e_id"].tolist() == [1, 3, 4, 5, 6]
    assert out["edge_index"].tolist() == [[1, 2, 3, 0, 2], [0, 0, 1, 1, 2]]


def test_neighbor_sample_structure_on_a_random_graph():
    g = torch.Generator().manual_seed(0)
    n, e = 300, 9000
    ei = torch.randint(0, n, (2, e), generator=g)
    ei[1, :400] = 7                                      # a hub
    seeds = torch.tensor([7, 3, 99, 250])
    out = S.neighbor_sample(ei, n, seeds, [30, 30, 30], S.make_tape(n, S.TAPE_STRIDE, 1))
    n_id, sub, e_id = out["n_id"], out["edge_index"], out["e_id"]
    assert n_id[:4].tolist() == seeds.tolist() and n_id.unique().numel() == n_id.numel()
    assert torch.equal(n_id[sub[0]], ei[0, e_id]) and torch.equal(n_id[sub[1]], ei[1, e_id])
    assert e_id.unique().numel() == e_id.numel()
    indeg = torch.bincount(ei[1], minlength=n)
    per_dst = torch.bincount(n_id[sub[1]], minlength=n)
    expanded = per_dst > 0
    assert bool((per_dst[expanded] == indeg[expanded].clamp(max=30)).all())
    assert [len(h) for h in out["hops"]][0] == 4 and sum(len(h) for h in out["hops"]) == n_id.numel()
    # the last hop's discoveries are never expanded: they are sources only
    last = set(out["hops"][-1])
    assert not (set(n_id[sub[1]].tolist()) & last)


def test_random_walk_and_saint_batch():
    ei = torch.tensor([[0, 0, 1, 2, 2, 4], [1, 2, 2, 0, 3, 4]])        # node 3: a sink; node 4: only a self loop; 5: isolated
    start = torch.tensor([0, 3, 4, 5])
    tape = _as_i32([[_word_for(1, 2), _word_for(0, 2), _word_for(1, 2)],       # 0 -> 2 -> 0 -> 2
                    [0, 0, 0], [0, 0, 0], [0, 0, 0]])
    walks = S.random_walk(ei, 6, start, 3, tape)
    assert walks.tolist() == [[0, 2, 0, 2], [3, 3, 3, 3], [4, 4, 4, 4], [5, 5, 5, 5]]
    out = S.saint_random_walk_batch(ei, 6, start, 3, tape)
    assert out["n_id"].tolist() == [0, 2, 3, 4, 5]
    assert out["e_id"].tolist() == [1, 3, 4, 5]                     # every edge among {0, 2, 3, 4, 5}
    assert out["edge_index"].tolist() == [[0, 1, 1, 3], [1, 0, 2, 3]]
    # brute force on a random graph
    g = torch.Generator().manual_seed(3)
    n = 120
    ei = torch.randint(0, n, (2, 900), generator=g)
    start = torch.randint(0, n, (16,), generator=g)
    out = S.saint_random_walk_batch(ei, n, start, 10, S.make_tape(16, 10, 4))
    w = out["walks"]
    assert torch.equal(w[:, 0], start)
    edges = set(zip(ei[0].tolist(), ei[1].tolist()))
    outdeg = torch.bincount(ei[0], minlength=n)
    for a, b in zip(w[:, :-1].reshape(-1).tolist(), w[:, 1:].reshape(-1).tolist()):
        assert (a, b) in edges or (a == b and outdeg[a] == 0)
    inside = set(out["n_id"].tolist())
    want = [i for i, (s, d) in enumerate(zip(ei[0].tolist(), ei[1].tolist())) if s in inside and d in inside]
    assert out["e_id"].tolist() == want
    assert torch.equal(out["n_id"][out["edge_index"]], ei[:, out["e_id"]])
