"""Device negative sampling (csrc/negsample.hip): the properties PyG's negative_sampling guarantees and the
reference relies on (biomedkg/kge_module.py:86-102) -- k distinct pairs, none a positive, ids in range, uniform
over the non-edges -- and the keyed index bijection that stands in for x[randperm(n)].  Needs an MI355X."""
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
    free = n * n - torch.unique(ei[0] * n + ei[1]).numel()
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
    got = torch.bincount(cell(neg), minlength=256).double()
    want = free / free.sum() * k
    chi2 = float(((got - want) ** 2 / want).sum())
    assert chi2 < 255 + 6 * (2 * 255) ** 0.5, chi2          # mean 255, sd ~22.6 for 255 dof


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


def test_degenerate_inputs():
    from biomedkg_amd import rand as R, sampling
    empty = torch.empty(2, 0, dtype=torch.int64, device=DEV)
    assert sampling.negative_sampling(empty, 50, 0, seed=1).shape == (2, 0)          # nothing to draw
    got = sampling.negative_sampling(empty, 50, 30, seed=1)                         # no positives at all
    assert got.shape == (2, 30) and torch.unique(got[0] * 50 + got[1]).numel() == 30
    one = torch.tensor([[3], [3]], device=DEV)
    got = sampling.negative_sampling(one, 7, 48, seed=2)                            # every free cell of a 7 x 7 grid
    keys = torch.unique(got[0] * 7 + got[1])
    assert keys.numel() == got.size(1) and 24 not in keys.tolist()
    assert R.DeviceRand(seed=1).shuffle(torch.empty(0, dtype=torch.int64, device=DEV)).numel() == 0
