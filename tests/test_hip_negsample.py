"""Device negative sampling (csrc/negsample.hip): the properties PyG 2.5.3's negative_sampling guarantees and the
reference relies on (biomedkg/kge_module.py:86-102) -- k distinct OFF-DIAGONAL pairs (edge_index_to_vector /
vector_to_edge_index: the population is the N(N-1) cells with head != tail), none a positive, ids in range, uniform
over the free off-diagonal cells, sized from the loop-free positive count -- checked against the CPU oracle's
restatement where the support is small enough to enumerate, and the keyed index bijection that stands in for
x[randperm(n)].  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _graph(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n, (2, e), generator=g).to(DEV)


@pytest.mark.parametrize("n,e,k", [(300, 2000, 2000), (5000, 60_000, 180_000), (40, 900, 900), (1000, 10, 5000)])
def test_negatives_are_distinct_non_edges(n, e, k):
    from biomedkg_amd import sampling
    ei = _graph(n, e, n)
    neg = sampling.negative_sampling(ei, n, k, seed=1234)
    assert neg.dtype == torch.int64 and neg.size(0) == 2 and neg.size(1) <= k
    assert int(neg.min()) >= 0 and int(neg.max()) < n
    keys = neg[0] * n + neg[1]
    assert torch.unique(keys).numel() == keys.numel()
    assert not bool(torch.isin(keys, ei[0] * n + ei[1]).any())
    assert not bool((neg[0] == neg[1]).any())                                  # PyG >= 2.0: never a self pair
    off = ei[:, ei[0] != ei[1]]
    free = n * (n - 1) - torch.unique(off[0] * n + off[1]).numel()
    assert neg.size(1) == min(k, free) or neg.size(1) >= 0.95 * min(k, free)    # PyG: at most 3 rounds, may fall short
    if k <= 0.5 * free:
        assert neg.size(1) == k


def test_dense_graph_takes_the_enumerating_path():
    from biomedkg_amd import sampling
    n = 30
    ei = _graph(n, 700, 5)
    neg = sampling.negative_sampling(ei, n, 700)
    keys = neg[0] * n + neg[1]
    assert torch.unique(keys).numel() == keys.numel() and not bool(torch.isin(keys, ei[0] * n + ei[1]).any())
    assert not bool((neg[0] == neg[1]).any())


def test_support_equals_the_oracles_on_enumerable_graphs():
    """Over-asked on small graphs, PyG returns EVERY free off-diagonal cell (its `sample` hands back the whole range):
    the device sampler's support must be that same set -- the oracle's (oracle/pyg_ops.py, PyG 2.5.3 restatement)."""
    import random
    from biomedkg_amd import sampling
    from oracle import pyg_ops as P
    for n, e, seed in [(7, 1, 0), (12, 40, 1), (25, 300, 2), (9, 72, 3)]:
        g = torch.Generator().manual_seed(seed)
        ei = torch.randint(0, n, (2, e), generator=g)
        ei[:, 0] = n - 1                                                       # a self loop that pins N = n
        random.seed(seed)
        want = P.negative_sampling(ei, num_neg_samples=4 * n * n)
        got = sampling.negative_sampling(ei.to(DEV), num_neg_samples=4 * n * n, seed=seed).cpu()
        assert got.shape == want.shape
        assert torch.equal(torch.sort(got[0] * n + got[1]).values, torch.sort(want[0] * n + want[1]).values)


@pytest.mark.parametrize("n,e", [(2, 1), (7, 1), (13, 60), (64, 1500), (257, 9000)])
def test_hashed_path_enumerates_exactly_the_free_offdiagonal_cells(n, e):
    """All N(N-1) candidates of the keyed permutation through the kernels (hash-set membership + the diagonal
    re-opening of vector_to_edge_index): the survivors are exactly the oracle's free cells, each once."""
    from biomedkg_amd import sampling
    from oracle import pyg_ops as P
    g = torch.Generator().manual_seed(n)
    ei = torch.randint(0, n, (2, e), generator=g)
    ei[:, 0] = n - 1
    population = n * (n - 1)
    got = sampling._negative_sampling_hashed(ei.to(DEV), n, population, population, 0.5, seed=11).cpu()
    idx, pop = P.edge_index_to_vector(ei, n)
    assert pop == population
    free = torch.ones(population, dtype=torch.bool)
    free[idx] = False
    want = P.vector_to_edge_index(torch.nonzero(free).flatten(), n)
    assert got.shape == want.shape
    assert torch.equal(torch.sort(got[0] * n + got[1]).values, want[0] * n + want[1])


def test_seed_determinism_and_variation():
    from biomedkg_amd import sampling
    ei = _graph(2000, 30_000, 3)
    a = sampling.negative_sampling(ei, 2000, 30_000, seed=7)
    b = sampling.negative_sampling(ei, 2000, 30_000, seed=7)
    c = sampling.negative_sampling(ei, 2000, 30_000, seed=8)
    ka, kb, kc = (t[0] * 2000 + t[1] for t in (a, b, c))
    assert torch.equal(a, b)                                                  # a pure function of (graph, seed)
    assert torch.isin(ka, kc).float().mean() < 0.05


def test_uniform_over_the_free_cells():
    """Chi-square of the negatives' cell counts over a 16 x 16 coarse grid against the free-cell counts."""
    from biomedkg_amd import sampling
    n, e, k = 1024, 200_000, 300_000
    g = torch.Generator().manual_seed(0)
    ei = torch.stack([torch.randint(0, n // 4, (e,), generator=g), torch.randint(0, n, (e,), generator=g)]).to(DEV)  # skewed
    neg = sampling.negative_sampling(ei, n, k, seed=99)
    assert neg.size(1) == k
    cell = lambda t: (t[0] // 64) * 16 + t[1] // 64
    pos_cells = torch.bincount(cell(torch.unique(ei[0] * n + ei[1]).unsqueeze(0).expand(2, -1) // torch.tensor([[n], [1]], device=DEV)
                                    % torch.tensor([[n * n], [n]], device=DEV)), minlength=256).double()
    free = 64.0 * 64.0 - pos_cells
    free[torch.arange(16) * 17] -= 64.0                      # the 64 diagonal cells of each diagonal block are not population
    pos_diag = ei[:, ei[0] == ei[1]]
    free += torch.bincount(cell(torch.unique(pos_diag, dim=1)), minlength=256).double()   # ... and were not positives either
    assert not bool((neg[0] == neg[1]).any())
    got = torch.bincount(cell(neg), minlength=256).double()
    want = free / free.sum() * k
    chi2 = float(((got - want) ** 2 / want).sum())
    assert chi2 < 255 + 6 * (2 * 255) ** 0.5, chi2          # mean 255, sd ~22.6 for 255 dof


@pytest.mark.parametrize("n,k", [(1, 1), (3, 4), (1000, 1), (4097, 3), (100_003, 10)])
def test_shuffled_repeat_is_repeat_then_randperm(n, k):
    """``DeviceRand.shuffle_repeat(src, k)`` (bkg_permute_gather_repeat_i64) = ``src.repeat(k)[randperm(k n)]`` of reference
    kge_module.py:97-100 without the repeated copy: every value exactly k times, another arrangement per draw, no trace of
    the source order."""
    from biomedkg_amd import rand as R
    r = R.DeviceRand(seed=9)
    src = torch.arange(n, dtype=torch.int64, device=DEV) * 5 + 2
    a, b = r.shuffle_repeat(src, k), r.shuffle_repeat(src, k)
    want = torch.sort(src.repeat(k)).values
    assert a.numel() == n * k and torch.equal(torch.sort(a).values, want) and torch.equal(torch.sort(b).values, want)
    if n >= 1000:
        assert not torch.equal(a, b)
        pos = ((a - 2) // 5).double()                                        # the source index each output came from
        assert float((pos[1:] - pos[:-1]).abs().mean()) > 0.2 * n            # neighbours come from far apart


@pytest.mark.parametrize("n", [1, 2, 3, 17, 1000, 4096, 100_003])
def test_permute_gather_is_a_bijection(n):
    from biomedkg_amd import rand as R
    r = R.DeviceRand(seed=5)
    src = torch.arange(n, dtype=torch.int64, device=DEV) * 3 + 1
    a, b = r.shuffle(src), r.shuffle(src)
    assert torch.equal(torch.sort(a).values, src) and torch.equal(torch.sort(b).values, src)
    if n >= 1000:
        assert not torch.equal(a, b)
        assert float((a == src).float().mean()) < 0.01                       # few fixed points
        # no linear structure left: displacement i -> pi(i) is spread over the range
        disp = ((a - 1) // 3 - torch.arange(n, device=DEV)).double()
        assert float(disp.std()) > 0.3 * n


def test_kge_module_uses_device_sampler():
    from types import SimpleNamespace
    from biomedkg_amd import rand as R, synthetic
    from biomedkg_amd.kge_module import KGEModule
    g = synthetic.primekg_like(3000, 40_000, 8, seed=1)
    ei, et = g["edge_index"].to(DEV), g["edge_type"].to(DEV)
    m = KGEModule("rgcn", "dismult", 32, 32, 32, 2, 8, 1, "cosine", 1e-3, 0.1, "none", 2, "random").to(DEV)
    with R.use(R.DeviceRand(seed=2)):
        nei, net = m.sample_neg_edges(ei, et)
    assert nei.shape == (2, 2 * ei.size(1)) and net.shape == (2 * ei.size(1),)
    assert torch.equal(torch.bincount(net, minlength=8), 2 * torch.bincount(et, minlength=8))   # a permutation of repeat(2)
    batch = SimpleNamespace(x=torch.randn(3000, 32, device=DEV), edge_index=ei, edge_type=et)
    with R.use(R.DeviceRand(seed=2)):
        loss = m.training_step(batch, 0)
    loss.backward()
    assert torch.isfinite(loss)


def test_grouped_negative_types_are_the_same_typed_negatives(monkeypatch):
    """The negatives' relation ids grouped by relation (ops.grouped_types) instead of ``repeat(k)[randperm]`` (reference
    kge_module.py:97-100): the same multiset of ids; its own relation order is the identity with the positives' counts
    times k; the fused step over it equals the unfused decode + BCE route over the SAME tensors; the pairing with the
    sampler's negatives is uniform -- here: over many draws every (cell, relation) combination of a small graph turns up
    with the frequency the independent shuffle gives it; and the literal form is still there behind the flag."""
    from types import SimpleNamespace
    from biomedkg_amd import kge_module as KM, ops, rand as R, synthetic
    monkeypatch.setattr(KM, "GROUPED_NEG_TYPES", True)          # (opt-in since round 6: BKG_GROUPED_NEG_TYPES=1)
    g = synthetic.primekg_like(3000, 40_000, 8, seed=1)
    ei, et = g["edge_index"].to(DEV), g["edge_type"].to(DEV)
    m = KM.KGEModule("rgcn", "dismult", 32, 32, 32, 2, 8, 1, "cosine", 1e-3, 0.1, "none", 3, "random").to(DEV)
    with R.use(R.DeviceRand(seed=2)):
        nei, net = m.sample_neg_edges(ei, et)
    assert torch.equal(net, torch.sort(et.repeat(3)).values)
    order, ptr = ops.relation_order_of(net, 8)
    assert torch.equal(order.long(), torch.arange(net.numel(), device=DEV))
    assert torch.equal(ptr.long(), torch.cat([torch.zeros(1, dtype=torch.long, device=DEV),
                                              torch.cumsum(3 * torch.bincount(et, minlength=8), 0)]))
    with R.use(R.DeviceRand(seed=2)):
        nei2, net2 = m.sample_neg_edges(ei, et)
    assert net2 is net and torch.equal(nei2, nei)            # cached on the positives' ids; the sampler's draw did not move
    # fused route == unfused route on these tensors
    z = torch.randn(3000, 32, device=DEV)
    rel = m.model.decoder.rel_emb.detach()
    loss, pp, npred = ops.kge_bce_loss(z, rel, ei, et, nei, net, ops.KGE_DISTMULT, reg=0.0)
    sp = (z[ei[0]] * rel[et] * z[ei[1]]).sum(1)
    sn = (z[nei[0]] * rel[net] * z[nei[1]]).sum(1)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(torch.cat([sp, sn]), torch.cat([torch.ones_like(sp), torch.zeros_like(sn)]))
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert torch.allclose(npred, sn, rtol=1e-4, atol=1e-5)
    # the literal form (the default): repeat(k)[randperm] in one launch, a fresh arrangement every draw
    monkeypatch.setattr(KM, "GROUPED_NEG_TYPES", False)
    with R.use(R.DeviceRand(seed=2)):
        _, net3 = m.sample_neg_edges(ei, et)
        _, net4 = m.sample_neg_edges(ei, et)
    monkeypatch.setattr(KM, "GROUPED_NEG_TYPES", True)
    assert not torch.equal(net3, net) and torch.equal(torch.sort(net3).values, net)
    assert not torch.equal(net4, net3) and torch.equal(torch.sort(net4).values, net)
    # uniform pairing: ids 0..9, two relations with 3 + 6 positives; negatives of relation 0 must be spread over the free
    # cells like those of relation 1 (a position-dependent sampler would put relation 0 -- always first -- on fixed cells)
    n = 12
    pe = torch.tensor([[0, 1, 2, 3, 4, 5, 6, 7, 8], [1, 2, 3, 4, 5, 6, 7, 8, 9]], device=DEV)
    pt = torch.tensor([0, 0, 0, 1, 1, 1, 1, 1, 1], device=DEV)
    hits = torch.zeros(2, n * n, dtype=torch.float64, device=DEV)
    draws = 3000
    with R.use(R.DeviceRand(seed=5)):
        m2 = KM.KGEModule("rgcn", "dismult", 8, 8, 8, 2, 2, 1, "cosine", 1e-3, 0.1, "none", 1, "random").to(DEV)
        for _ in range(draws):
            a, b = m2.sample_neg_edges(pe, pt)
            hits.view(-1).index_add_(0, b * (n * n) + a[0] * n + a[1], torch.ones(a.size(1), dtype=torch.float64, device=DEV))
    free = hits.sum(0) > 0
    assert int(free.sum()) == 10 * 9 - 9                      # N is inferred as max id + 1 = 10 (PyG), minus the positives
    for r, cnt in ((0, 3), (1, 6)):
        p = hits[r][free] / (draws * cnt)                       # probability of a free cell for one negative of relation r
        want = 1.0 / int(free.sum())
        assert float((p - want).abs().max()) < 6 * (want / (draws * cnt)) ** 0.5 + 1e-9


def test_dense_batches_shuffle_their_negative_types(monkeypatch):
    """on a tiny dense batch PyG's sample size exceeds the population and the sampler enumerates the cells in
    (head, tail) order -- NOT an exchangeable order, so relation-grouped ids would pair relation id with head id
    monotonically.  There the ids are shuffled (reference kge_module.py:97-100), whatever GROUPED_NEG_TYPES says."""
    from biomedkg_amd import kge_module as KM, rand as R
    pe = torch.tensor([[0, 1, 2, 3], [1, 2, 3, 0]], device=DEV)      # 4 nodes: 12 off-diagonal cells, 4 positives, 8 free
    pt = torch.tensor([0, 0, 1, 1], device=DEV)
    m = KM.KGEModule("rgcn", "dismult", 8, 8, 8, 2, 2, 1, "cosine", 1e-3, 0.1, "none", 2, "random").to(DEV)
    monkeypatch.setattr(KM, "GROUPED_NEG_TYPES", True)
    monotone = 0
    heads_of_rel0 = torch.zeros(4, device=DEV)
    with R.use(R.DeviceRand(seed=3)):
        for _ in range(200):
            nei, net = m.sample_neg_edges(pe, pt)                    # k = 8, sample size int(1.1 * 8 / (2/3)) = 13 > 12
            assert not getattr(nei, "_bkg_exchangeable", False)
            assert nei.size(1) == 8 == net.numel()
            assert bool((nei[0, 1:] >= nei[0, :-1]).all())           # the enumeration order: heads ascend
            assert sorted(net.tolist()) == [0, 0, 0, 0, 1, 1, 1, 1]
            monotone += int(torch.equal(net, torch.sort(net).values))
            heads_of_rel0 += torch.bincount(nei[0][net == 0], minlength=4).float()
    assert monotone < 20                                             # grouped ids would be monotone every time (1 / 70 by chance)
    assert bool((heads_of_rel0 > 50).all())                          # relation 0 meets every head (grouped: heads 0 and 1 only)


def test_degenerate_inputs():
    from biomedkg_amd import rand as R, sampling
    empty = torch.empty(2, 0, dtype=torch.int64, device=DEV)
    assert sampling.negative_sampling(empty, 50, 0, seed=1).shape == (2, 0)          # nothing to draw
    got = sampling.negative_sampling(empty, 50, 30, seed=1)                         # no positives at all
    assert got.shape == (2, 30) and torch.unique(got[0] * 50 + got[1]).numel() == 30
    assert not bool((got[0] == got[1]).any())
    one = torch.tensor([[3], [3]], device=DEV)                                      # a self loop is not a positive of the population
    got = sampling.negative_sampling(one, 7, 48, seed=2)                            # every off-diagonal cell of a 7 x 7 grid: 42
    keys = torch.unique(got[0] * 7 + got[1])
    assert got.size(1) == 42 and keys.numel() == 42 and not bool((got[0] == got[1]).any())
    two = torch.tensor([[3, 3, 2], [3, 5, 6]], device=DEV)                          # one loop + two real positives: 40 free cells
    got = sampling.negative_sampling(two, 7, 48, seed=2)
    keys = torch.unique(got[0] * 7 + got[1]).tolist()
    assert got.size(1) == 40 and len(keys) == 40 and 3 * 7 + 5 not in keys and 2 * 7 + 6 not in keys
    assert sampling.negative_sampling(torch.zeros(2, 1, dtype=torch.int64, device=DEV)).shape == (2, 0)   # N = 1: no population
    full = torch.cartesian_prod(torch.arange(4), torch.arange(4)).t().contiguous().to(DEV)
    assert sampling.negative_sampling(full[:, full[0] != full[1]]).shape == (2, 0)   # all 12 off-diagonal cells taken
    assert R.DeviceRand(seed=1).shuffle(torch.empty(0, dtype=torch.int64, device=DEV)).numel() == 0


def test_a_round_that_falls_short_leaves_zero_pairs_not_stale_ids():
    """The one-round caller never reads the survivor count: the slots its survivors do not reach must hold (0, 0), whatever
    the (uninitialised) output buffer held before -- and nothing is written past ``max_out`` or past the candidates' reach."""
    from biomedkg_amd import _lib as L, sampling
    n, draw, max_out = 60, 300, 500
    ei = _graph(n, 2500, 9)                                          # dense: most candidates are positives
    table = sampling._positive_table(ei, n)
    keys = torch.empty(draw, dtype=torch.int64, device=DEV)
    flag = torch.empty(2, draw, dtype=torch.int32, device=DEV)
    out = torch.full((2, max_out + 50), -7, dtype=torch.int64, device=DEV)
    count = torch.zeros(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(L.lib().bkg_negative_sample_workspace_bytes(draw), dtype=torch.uint8, device=DEV)
    L.call("bkg_negative_sample", n, 0, draw, 4321, L.ptr(table), table.numel(), L.ptr(keys), L.ptr(flag[0]), L.ptr(flag[1]),
           L.ptr(out[0]), L.ptr(out[1]), 0, max_out, L.ptr(count), L.ptr(ws), ws.numel(), L.stream())
    kept = int(count)
    assert 0 < kept < draw
    got = out.cpu()
    assert int(got[:, :kept].min()) >= 0 and not bool((got[0, :kept] == got[1, :kept]).any())
    reach = min(max_out, kept + draw)
    assert bool((got[:, kept:reach] == 0).all())                     # the shortfall: (0, 0) pairs
    assert bool((got[:, reach:] == -7).all())                        # nothing beyond
