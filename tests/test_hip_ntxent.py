"""NT-Xent v2 (csrc/ntxent2.hip) against the CPU oracle's restatement of PyGCL's DualBranchContrast(InfoNCE, L2L,
intraview_negs=True) -- reference biomedkg/gcl_module.py:171-173,186-190: exact-fp32 MFMA path (loss |d| <= 1e-5, the
bar of BASELINE configs[1]) over sizes that hit every tiling edge (one partial panel, several panels, partial last
j-tile, widths that are not multiples of 32, both backward workgroup shapes), the bf16-operand path against the exact
one, and bitwise repeatability of both directions (the sums are reduced in a fixed order, no atomics).  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _pair(n, h, seed, corr=0.6):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(n, h, generator=g)
    return a, corr * a + (1 - corr) * torch.randn(n, h, generator=g)


@pytest.mark.parametrize("n,h", [(2, 4), (3, 5), (31, 32), (65, 33), (130, 96), (257, 128), (300, 129), (700, 256),
                                 (1500, 160), (1029, 64)])
def test_fp32_matches_oracle(n, h):
    from biomedkg_amd import ops
    from oracle import pygcl_ops as G
    a, b = _pair(n, h, n + h)
    ao, bo = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = G.dual_branch_l2l_intraview(ao.double(), bo.double(), 0.2)
    ref.backward()
    ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    loss = ops.ntxent_loss(ad, bd, 0.2, operands="fp32")
    (2.5 * loss).backward()                                              # a non-trivial upstream gradient
    assert abs(float(loss) - float(ref)) <= 1e-5, (float(loss), float(ref))
    for got, want in ((ad.grad, ao.grad), (bd.grad, bo.grad)):
        err = float((got.cpu().double() - 2.5 * want.double()).norm() / (2.5 * want.double().norm()))
        assert err < 1e-5, err


# (h > 128 takes the software-pipelined backward and the 64-anchors-per-wave forward: the small sizes are its one-, two- and
# three-tile blocks, odd and even tile counts, partial last tiles and panels)
@pytest.mark.parametrize("n,h", [(3000, 128), (2000, 256), (513, 100), (20, 256), (40, 200), (65, 256), (97, 130), (129, 256),
                                 (1111, 256)])
def test_bf16_operands_track_the_exact_path(n, h):
    from biomedkg_amd import ops
    a, b = _pair(n, h, 7)
    res = {}
    for mode in ("fp32", "bf16"):
        ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        loss = ops.ntxent_loss(ad, bd, 0.2, operands=mode)
        loss.backward()
        res[mode] = (float(loss), torch.cat([ad.grad, bd.grad]))
    # (the loss is a mean over 2 n rows: bf16 rounding of the operands averages out as 1 / sqrt(n))
    tol = 5e-5 * max(1.0, (3000.0 / n) ** 0.5)
    assert abs(res["bf16"][0] - res["fp32"][0]) <= tol * abs(res["fp32"][0]), (res["bf16"][0], res["fp32"][0])
    err = float((res["bf16"][1] - res["fp32"][1]).norm() / res["fp32"][1].norm())
    assert err < 1e-2, err


@pytest.mark.parametrize("mode,h", [("fp32", 128), ("bf16", 128), ("bf16", 256)])
def test_bitwise_repeatable(mode, h):
    from biomedkg_amd import ops
    a, b = _pair(2500, h, 3)
    outs = []
    for _ in range(3):
        ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        loss = ops.ntxent_loss(ad, bd, 0.2, operands=mode)
        loss.backward()
        outs.append((loss.detach().clone(), ad.grad.clone(), bd.grad.clone()))
    for o in outs[1:]:
        assert all(torch.equal(x, y) for x, y in zip(o, outs[0]))


def test_precision_mode_selects_the_operands_and_width_limit():
    from biomedkg_amd import ops
    a, b = _pair(600, 128, 1)
    ad, bd = a.to(DEV), b.to(DEV)
    exact = float(ops.ntxent_loss(ad, bd, 0.2, operands="fp32"))
    fast = float(ops.ntxent_loss(ad, bd, 0.2, operands="bf16"))
    ops.set_matmul_precision("bf16")
    try:
        assert float(ops.ntxent_loss(ad, bd, 0.2)) == fast
    finally:
        ops.set_matmul_precision("fp32")
    assert float(ops.ntxent_loss(ad, bd, 0.2)) == exact
    with pytest.raises(ValueError):
        ops.ntxent_loss(torch.randn(8, 300, device=DEV), torch.randn(8, 300, device=DEV))
