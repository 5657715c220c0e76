#!/usr/bin/env python3
"""Micro-benchmarks of the MFMA-bound kernels (HIP events, median of reps): NT-Xent forward / backward and the
GEMM tiles, against the fp32-input MFMA peak (155 TFLOP/s measured, MI355X_MICROARCH.md) or the bf16 peak.
    python scripts/bench_kernels.py            # prints one JSON object per line"""
import json
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs)
    return t[len(t) // 2]


def main():
    dev = "cuda"
    torch.manual_seed(0)
    for n, h in ((5000, 128), (16384, 128), (32768, 256)):
        h1, h2 = torch.randn(n, h, device=dev, requires_grad=True), torch.randn(n, h, device=dev, requires_grad=True)
        m = 2 * n
        fwd = timeit(lambda: ops.ntxent_loss(h1, h2, 0.2))
        loss = ops.ntxent_loss(h1, h2, 0.2)
        bwd = timeit(lambda: torch.autograd.grad(loss, (h1, h2), retain_graph=True))
        f_fwd, f_bwd = 2.0 * m * m * h, 4.0 * m * m * h
        print(json.dumps({"kernel": "ntxent", "N": n, "h": h, "fwd_ms": fwd, "bwd_ms": bwd,
                          "fwd_TFLOPs": f_fwd / fwd / 1e9, "bwd_TFLOPs": f_bwd / bwd / 1e9, "peak_TFLOPs_fp32_mfma": 155.0,
                          "note": "whole op incl. normalise / log glue; flops = full (2N)^2 Gram, no symmetry credit"}))
    for prec, peak in (("fp32", 155.0), ("bf16", 2500.0)):
        ops.set_matmul_precision(prec)
        for (mm, nn, kk, ta, tb) in ((130000, 256, 256, False, True), (130000, 256, 768, False, True),
                                     (256, 256, 130000, True, False), (8192, 8192, 8192, False, True)):
            a = torch.randn((kk, mm) if ta else (mm, kk), device=dev)
            b = torch.randn((nn, kk) if tb else (kk, nn), device=dev)
            ms = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb), reps=10)
            print(json.dumps({"kernel": "gemm", "precision": prec, "M": mm, "N": nn, "K": kk, "ta": ta, "tb": tb, "ms": ms,
                              "TFLOPs": 2.0 * mm * nn * kk / ms / 1e9, "peak_TFLOPs": peak,
                              "GBs_io_fp32": 4.0 * (mm * kk + kk * nn + mm * nn) / ms / 1e6}))
    ops.set_matmul_precision("fp32")


if __name__ == "__main__":
    main()
