#!/usr/bin/env python3
"""Tall activation x weight products (gemm_xw.hip), K = N = 256, bf16 rows (the 8-wave register-weight kernel) against
an fp64 product of the same bf16 values, with every epilogue it takes; HIP-event timings (each includes the weight
conversion launch and one launch gap).  `python scripts/dev_xw8.py [M ...]`"""
import json
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
from biomedkg_amd import ops  # noqa: E402
from bench_gemm import timeit            # noqa: E402


def main():
    dev = "cuda"
    ops.set_matmul_precision("bf16")
    torch.manual_seed(0)
    w = torch.randn(256, 256, device=dev) * 0.1
    bias = torch.randn(256, device=dev)
    errs = {}
    for m in (4096, 4097, 4159, 4160, 4161, 130_000, 130_001, 260_000):
        a = torch.randn(m, 256, device=dev).bfloat16()
        ref = a.double() @ w.bfloat16().double().t()
        for cdt in (torch.bfloat16, torch.float32):
            for kw in ({}, {"bias": bias, "act": ops.ACT_RELU}, {"alpha": 0.5, "act": ops.ACT_ELU}):
                out = ops.gemm(a, w, trans_b=True, out_dtype=cdt, **kw)
                r = ref * kw.get("alpha", 1.0) + (kw["bias"].double() if "bias" in kw else 0.0)
                if kw.get("act") == ops.ACT_RELU:
                    r = r.clamp_min(0)
                elif kw.get("act") == ops.ACT_ELU:
                    r = torch.where(r > 0, r, torch.expm1(r))
                tol = 1e-2 if cdt == torch.bfloat16 else 1e-5
                e = float((out.double() - r).abs().max() / r.abs().max().clamp_min(1e-9))
                errs[f"{m} {str(cdt)[6:]} {sorted(kw)}"] = round(e, 7)
                assert e < tol, (m, cdt, kw, e)
    print(json.dumps({"worst_rel_err": max(errs.values())}))
    sizes = [int(x) for x in sys.argv[1:]] or [130_000, 260_000]
    for m in sizes:
        a = torch.randn(m, 256, device=dev).bfloat16()
        for cdt in (torch.bfloat16, torch.float32):
            ms = timeit(lambda: ops.gemm(a, w, trans_b=True, out_dtype=cdt))
            byt = m * 256 * 2 + m * 256 * (2 if cdt == torch.bfloat16 else 4)
            print(json.dumps({"M": m, "C": str(cdt)[6:], "us": round(ms * 1e3, 1), "GBs_io": round(byt / ms / 1e6)}))


if __name__ == "__main__":
    main()
