"""GGD's / DGI's two encoder passes side by side in the EXACT fp32 mode (ops._GcnConvViewsF32, round 4): the corruption is a
row permutation, every product is row-independent, so ``x[perm] W^T = (x W^T)[perm]`` holds bit for bit and the two views
can share the first product and every launch.  Against the two separate passes: same loss, same gradients (the aggregation
adds a row's entries in another lane grouping at twice the width: fp32 rounding only).  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _fp32_mode():
    from biomedkg_amd import ops
    was = ops.get_matmul_precision()
    ops.set_matmul_precision("fp32")
    yield
    ops.set_matmul_precision(was)


def _problem(n=6000, e=100_000, seed=12):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    ei[1, : e // 5] = 5                                  # a hub target: long rows
    return ei.to(DEV), torch.randn(n, 768, generator=g).to(DEV)


@pytest.mark.parametrize("kind", ["ggd", "dgi"])
@pytest.mark.parametrize("aug", [False, True])
def test_fp32_views_equal_two_passes(kind, aug):
    from biomedkg_amd import ops, rand as R
    from biomedkg_amd.gcl_module import DGIModule, GGDModule

    class TwoPass(R.DeviceRand):
        device_draws = False

    ei, x = _problem()
    n = x.size(0)
    torch.manual_seed(0)
    cls = GGDModule if kind == "ggd" else DGIModule
    m = cls(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(DEV).train()
    m.model.encoder.drop_out = False                     # (the two forms draw their dropout masks in different shapes)
    if kind == "ggd":
        m.model.p = 1.0 if aug else 0.0
    elif aug:
        pytest.skip("DGI has no augmentation")
    assert m.model.encoder.views_ok(n)
    res = []
    for prov in (TwoPass(seed=1), R.DeviceRand(seed=1)):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(99)                            # same permutation / masks from torch's generator
        calls = []
        orig = ops.gcn_conv_views
        ops.gcn_conv_views = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            with R.use(prov):
                loss = m.calculate_loss(x, ei)
        finally:
            ops.gcn_conv_views = orig
        assert (len(calls) == 4) == prov.device_draws    # the side-by-side path ran exactly when asked
        loss.backward()
        res.append((float(loss.detach()), [p.grad.clone() for p in m.model.parameters() if p.grad is not None]))
    (la, ga), (lb, gb) = res
    assert abs(la - lb) <= 2e-6 * abs(la), (la, lb)
    assert len(ga) == len(gb)
    for a, b in zip(ga, gb):
        assert float((a - b).norm() / a.norm().clamp_min(1e-12)) < 2e-5


def test_the_shared_first_product_is_bitwise_the_permuted_product():
    from biomedkg_amd import ops
    _, x = _problem(n=5000)
    w = torch.randn(256, 768, device=DEV) * 0.05
    perm = torch.randperm(x.size(0), device=DEV)
    assert torch.equal(ops.xw_table(x, w)[perm], ops.xw_table(x[perm].contiguous(), w))
    assert ops.xw_table(x, w).dtype == torch.float32


def test_fp32_views_train_with_dropout_and_augmentation():
    from biomedkg_amd import rand as R
    from biomedkg_amd.gcl_module import GGDModule
    ei, x = _problem()
    torch.manual_seed(0)
    m = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(DEV).train()
    m.model.p = 0.5
    opt = torch.optim.Adam(m.model.parameters(), lr=1e-3)
    losses = []
    with R.use(R.DeviceRand(seed=4)):
        for _ in range(12):
            opt.zero_grad(set_to_none=True)
            loss = m.calculate_loss(x, ei)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
    assert all(torch.isfinite(torch.tensor(losses))) and min(losses[-3:]) < losses[0]


@pytest.mark.parametrize("m", [16384, 16385 + 64 * 3, 51_402, 25_691, 16384 + 64 * 130 + 7, 130_000])
def test_tall_fp32_products_with_a_split_last_round(m):
    """``gemm_xw32_kernel``: when the last round of 64-row tiles would leave more than half of the 256 blocks idle its tiles
    go out as 32-row halves (51 k rows: 803 tiles = 3 rounds + 35 -> 70 halves).  Every row against fp64, whatever item
    computed it; trans_w both ways; with bias + ReLU."""
    from biomedkg_amd import ops
    g = torch.Generator().manual_seed(m)
    a = torch.randn(m, 256, generator=g).to(DEV)
    w = (torch.randn(256, 256, generator=g) * 0.1).to(DEV)
    b = torch.randn(256, generator=g).to(DEV)
    ref = a.double() @ w.double().t()
    got = ops.gemm(a, w, trans_b=True)
    torch.testing.assert_close(got.double(), ref, rtol=1e-5, atol=1e-4)
    got2 = ops.gemm(a, w.t().contiguous())
    assert torch.equal(got, got2) or float((got - got2).abs().max()) < 1e-4
    lin = ops.linear(a, w, b, act=ops.ACT_RELU)
    torch.testing.assert_close(lin.double(), (ref + b.double()).relu(), rtol=1e-5, atol=1e-4)
    # rows of a product do not depend on where they sit (what the shared first product of the views relies on)
    perm = torch.randperm(m, device=DEV)
    assert torch.equal(ops.gemm(a[perm].contiguous(), w, trans_b=True), got[perm])
