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
