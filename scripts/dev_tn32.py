#!/usr/bin/env python3
"""Exact-fp32 weight-gradient products (gemm_tn32.hip: LDS-DMA ring, v_mfma_f32_32x32x2_f32) against fp64, atomic and
deterministic forms, with HIP-event timings (each includes the zero fill and one launch gap).
python scripts/dev_tn32.py"""
import json
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
from biomedkg_amd import ops            # noqa: E402
from bench_gemm import timeit           # noqa: E402

dev = "cuda"
ops.set_matmul_precision("fp32")


def check():
    torch.manual_seed(0)
    err = {}
    for (k, m, n) in ((130_000, 256, 256), (9001, 256, 768), (40_000, 128, 128), (66_001, 384, 128), (70_000, 128, 256)):
        a, b = torch.randn(k, m, device=dev), torch.randn(k, n, device=dev)
        out = ops.gemm(a, b, trans_a=True)
        ref = a.double().t() @ b.double()
        err[f"tn {k}x{m}x{n}"] = float((out.double() - ref).abs().max() / ref.abs().max())
    s, g = 300_000, 30
    w = torch.arange(1, g + 1, dtype=torch.float64).pow(-1.2)
    sizes = (w / w.sum() * s).long()
    sizes[5] = 0
    sizes[7] = 1
    sizes[0] += s - sizes.sum()
    gp = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(torch.int32).to(dev)
    h, dy = torch.randn(s, 256, device=dev), torch.randn(s, 256, device=dev)
    out = ops.gemm_grouped_k(h, dy, gp, g)
    worst = 0.0
    for i in range(g):
        sl = slice(int(gp[i]), int(gp[i + 1]))
        ref = h[sl].double().t() @ dy[sl].double()
        worst = max(worst, float((out[i].double() - ref).abs().max() / ref.abs().max().clamp_min(1.0)))
    err["grouped"] = worst
    return err, (h, dy, gp, g)


def times(pack):
    h, dy, gp, g = pack
    r = {}
    for (m, n) in ((256, 256), (256, 768)):
        a, b = torch.randn(130_000, m, device=dev), torch.randn(130_000, n, device=dev)
        ms = timeit(lambda: ops.gemm(a, b, trans_a=True))
        r[f"tn {m}x{n}"] = {"us": round(ms * 1e3, 1), "TFLOPs": round(2.0 * 130_000 * m * n / ms / 1e9, 1)}
    ms = timeit(lambda: ops.gemm_grouped_k(h, dy, gp, g))
    r["grouped 300k"] = {"us": round(ms * 1e3, 1), "TFLOPs": round(2.0 * 300_000 * 256 * 256 / ms / 1e9, 1)}
    return r


for det in (False, True):
    ops.set_deterministic(det)
    err, pack = check()
    assert all(v < 1e-5 for v in err.values()), err
    print(json.dumps({"deterministic": det, "rel_err": err, "times": times(pack)}))
ops.set_deterministic(False)

# ---- exact-fp32 activation x weight products with K = N = 256 (gemm_xw32.hip)
torch.manual_seed(1)
w, bias = torch.randn(256, 256, device=dev) * 0.1, torch.randn(256, device=dev)
worst = 0.0
for m in (16_384, 16_385, 20_001, 130_000, 130_001):
    a = torch.randn(m, 256, device=dev)
    for tb in (False, True):
        for kw in ({}, {"bias": bias, "act": ops.ACT_RELU, "alpha": 0.5}, {"act": ops.ACT_ELU}):
            out = ops.gemm(a, w, trans_b=tb, **kw)
            ref = (a.double() @ (w.double().t() if tb else w.double())) * kw.get("alpha", 1.0)
            ref = ref + (kw["bias"].double() if "bias" in kw else 0.0)
            if kw.get("act") == ops.ACT_RELU:
                ref = ref.clamp_min(0)
            elif kw.get("act") == ops.ACT_ELU:
                ref = torch.where(ref > 0, ref, torch.expm1(ref))
            worst = max(worst, float((out.double() - ref).abs().max() / ref.abs().max()))
assert worst < 1e-5, worst
r = {"rel_err": worst}
for m in (130_000, 260_000):
    a = torch.randn(m, 256, device=dev)
    for tb in (False, True):
        ms = timeit(lambda: ops.gemm(a, w, trans_b=tb))
        r[f"M={m} tb={tb}"] = {"us": round(ms * 1e3, 1), "TFLOPs": round(2.0 * m * 65536 / ms / 1e9, 1)}
print(json.dumps({"xw32": r}))
