"""ReDAF.forward as the epilogue of its product (``bkg_redaf_fused_xw``, reference ``biomedkg/utils/fusion.py:49-90``): the
one-launch form must give, bit for bit, what the product + ``bkg_redaf_combine_f32`` pair gives -- same accumulators, same
Philox masks, same order of the two modality terms -- and that pair is held to the reference's own class by the goldens
(``tests/test_hip_parity_golden.py``).  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(n, d, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(n, 2, d, device=DEV, generator=g)
    w = torch.randn(d, d, device=DEV, generator=g) / d ** 0.5
    b = torch.randn(d, device=DEV, generator=g) * 0.1
    scale = torch.rand(2, d, device=DEV, generator=g) + 0.25
    return x, w, b, scale


@pytest.mark.parametrize("n,d", [(130, 256), (2049, 256), (5000, 768), (4097, 512)])
@pytest.mark.parametrize("how", ["eval", "philox", "mask"])
def test_one_launch_equals_the_two_kernel_form(n, d, how):
    from biomedkg_amd import ops
    prev = ops.get_matmul_precision()
    ops.set_matmul_precision("bf16")
    try:
        x, w, b, scale = _inputs(n, d, 3)
        drop = None
        if how == "philox":
            drop = ops.Dropout(0.1, keep=None, seed=1234, stream_id=7)
        elif how == "mask":
            keep = torch.rand(n * 2, d, device=DEV) >= 0.1
            drop = ops.Dropout(0.1, keep=keep)
        outs = {}
        for one in (True, False):
            ops.REDAF_ONE_LAUNCH = one
            outs[one] = ops.redaf_fusion(x, w, b, scale, drop).clone()
        assert torch.equal(outs[True], outs[False])
        # and against plain torch on the bf16-rounded operands (fp32 accumulate): what both compute
        t = torch.relu(x.reshape(2 * n, d).bfloat16().float() @ w.bfloat16().float().t() + b) * scale.repeat(n, 1)
        if how == "mask":
            t = torch.where(keep, t / 0.9, torch.zeros(()).to(t))
        if how != "philox":
            ref = torch.relu(t).view(n, 2, d).mean(1)
            torch.testing.assert_close(outs[True], ref, rtol=2e-4, atol=2e-4)
        else:
            kept = float((outs[True] > 0).float().mean())
            assert 0.2 < kept < 0.9
    finally:
        ops.REDAF_ONE_LAUNCH = True
        ops.set_matmul_precision(prev)


def test_shapes_outside_the_streaming_product_keep_the_two_kernel_form():
    from biomedkg_amd import ops
    prev = ops.get_matmul_precision()
    try:
        for mode, d in (("fp32", 256), ("bf16", 96)):
            ops.set_matmul_precision(mode)
            x, w, b, scale = _inputs(300, d, 5)
            out = ops.redaf_fusion(x, w, b, scale, None)
            ref = torch.relu(torch.relu(x.reshape(600, d) @ w.t() + b) * scale.repeat(300, 1)).view(300, 2, d).mean(1)
            torch.testing.assert_close(out, ref, rtol=2e-2 if mode == "bf16" else 1e-4, atol=2e-2 if mode == "bf16" else 1e-4)
    finally:
        ops.set_matmul_precision(prev)


def _keep_mask(rows, d, p, seed, stream_id, row0=0):
    """The keep decisions of the in-kernel hash (csrc/common.h hash_keep4) for a [rows, d] input: bkg_redaf_combine_f32 with
    one modality over ones."""
    from biomedkg_amd import _lib as L
    t = torch.ones(rows, d, device=DEV)
    scale = torch.ones(1, d, device=DEV)
    out = torch.empty(rows, d, device=DEV)
    L.call("bkg_redaf_combine_f32", L.ptr(t), d, rows, 1, d, L.ptr(scale), None, p, seed, stream_id, row0, L.ptr(out), d,
           L.stream())
    kept = out > 0
    torch.testing.assert_close(out[kept], torch.full_like(out[kept], 1.0 / (1.0 - p)), rtol=1e-6, atol=0)      # kept values are scaled
    return kept


def test_keep_hash_statistics():
    """Round 6: ReDAF's masks come from a 32-bit mixer with 16-bit thresholds (Philox4x32-7 before).  Drop rate as asked;
    neighbours along a row (inside and across the 4-element groups one hash serves), down a column, between consecutive
    calls (stream ids) and between seeds are uncorrelated; a chunked call (row0) continues the whole input's mask."""
    rows, d, p = 4096, 768, 0.1
    k = _keep_mask(rows, d, p, 1234, 7)
    n = k.numel()
    rate = 1.0 - float(k.float().mean())
    assert abs(rate - p) < 4 * (p * (1 - p) / n) ** 0.5 + 1e-5, rate
    kf = k.float() - (1 - p)
    var = p * (1 - p)

    def corr(a, b):
        return float((a * b).mean()) / var

    tol = 5.0 / n ** 0.5
    for shift in (1, 2, 3, 4, 5, 8, 16):                       # along a row
        assert abs(corr(kf[:, :-shift], kf[:, shift:])) < tol, shift
    for shift in (1, 2, 3, 7):                                 # down a column (rows 2i / 2i + 1: the two modalities of a node)
        assert abs(corr(kf[:-shift], kf[shift:])) < tol, shift
    for other in (_keep_mask(rows, d, p, 1234, 8), _keep_mask(rows, d, p, 1235, 7), _keep_mask(rows, d, p, 1234 + (1 << 32), 7)):
        assert abs(corr(kf, other.float() - (1 - p))) < tol
    # per-column and per-row rates: no column or row is favoured (768 / 4096 binomials, 6 sigma)
    assert float((k.float().mean(0) - (1 - p)).abs().max()) < 6 * (var / rows) ** 0.5
    assert float((k.float().mean(1) - (1 - p)).abs().max()) < 6 * (var / d) ** 0.5
    # other probabilities, including the ends of the threshold's range
    for q in (0.5, 0.9, 1e-3):
        r = 1.0 - float(_keep_mask(rows, d, q, 99, 1).float().mean())
        assert abs(r - q) < 5 * (q * (1 - q) / n) ** 0.5 + 2e-5, (q, r)
    # a call over rows [1000, 1512) of the input draws that part of the whole input's mask
    assert torch.equal(_keep_mask(512, d, p, 1234, 7, row0=1000), k[1000:1512])


@pytest.mark.parametrize("rows,d,p,seed,stream_id,row0", [
    (2048, 768, 0.1, 1234, 7, 0), (513, 256, 0.5, (5 << 32) + 99, 0xFFFF0003, 1000), (64, 512, 0.25, 2 ** 63 + 5, 1, 30_000_000),
    (300, 64, 0.9, 0, 0, 2 ** 31)])
def test_keep_hash_is_the_restated_one(rows, d, p, seed, stream_id, row0):
    """The mask definition is pinned: ``oracle/rand.py redaf_keep_mask`` restates ``hash_keep4`` (64-bit seeds, stream ids
    with the replay counter's high bits set, global rows past 2^34 / d -- the element index's high word) and the kernel's
    decisions equal it bit for bit."""
    from oracle.rand import redaf_keep_mask
    got = _keep_mask(rows, d, p, seed, stream_id, row0=row0)
    assert torch.equal(got.cpu(), redaf_keep_mask(rows, d, p, seed, stream_id, row0))
