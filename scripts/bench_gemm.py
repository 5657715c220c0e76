#!/usr/bin/env python3
"""GEMM shapes of the training step (HIP events, median of reps) with their operand + result traffic, next to a
device copy of the same byte count (what HBM gives a streaming kernel on this box).
    python scripts/bench_gemm.py [fp32|bf16]"""
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
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    ops.set_matmul_precision(prec)
    torch.manual_seed(0)
    n = 130_000
    src = torch.empty(n * 256, device=dev)
    dst = torch.empty_like(src)
    ms = timeit(lambda: dst.copy_(src))
    print(json.dumps({"kernel": "copy 133MB->133MB", "ms": ms, "GBs": 2 * src.numel() * 4 / ms / 1e6}))
    big_src = torch.empty(1 << 28, device=dev)
    big_dst = torch.empty_like(big_src)
    ms = timeit(lambda: big_dst.copy_(big_src))
    print(json.dumps({"kernel": "copy 1GiB->1GiB", "ms": ms, "GBs": 2 * big_src.numel() * 4 / ms / 1e6}))
    ms = timeit(lambda: big_src.sum())
    print(json.dumps({"kernel": "read 1GiB (sum)", "ms": ms, "GBs": big_src.numel() * 4 / ms / 1e6}))
    del big_src, big_dst
    if prec == "bf16":
        for (mm, nn, kk) in ((n, 256, 256), (n, 256, 768)):
            a = torch.randn(mm, kk, device=dev)
            b = torch.randn(nn, kk, device=dev)
            for adt, cdt in ((torch.float32, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16)):
                aa = a.to(adt)
                ms = timeit(lambda: ops.gemm(aa, b, trans_b=True, out_dtype=cdt))
                byt = mm * kk * aa.element_size() + mm * nn * (2 if cdt == torch.bfloat16 else 4)
                print(json.dumps({"kernel": "gemm_xw", "A": str(adt)[6:], "C": str(cdt)[6:], "M": mm, "N": nn, "K": kk,
                                  "ms": round(ms, 4), "GBs_io": round(byt / ms / 1e6)}))
    for (mm, nn, kk, ta, tb) in ((n, 256, 256, False, True), (n, 256, 256, False, False), (n, 256, 768, False, True),
                                 (n, 768, 256, False, False), (256, 256, n, True, False), (256, 768, n, True, False),
                                 (768, 256, n, True, False)):
        a = torch.randn((kk, mm) if ta else (mm, kk), device=dev)
        b = torch.randn((nn, kk) if tb else (kk, nn), device=dev)
        ms = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb))
        print(json.dumps({"kernel": "gemm", "precision": prec, "M": mm, "N": nn, "K": kk, "ta": ta, "tb": tb,
                          "ms": round(ms, 4), "TFLOPs": round(2.0 * mm * nn * kk / ms / 1e9, 1),
                          "GBs_io": round(4.0 * (mm * kk + kk * nn + mm * nn) / ms / 1e6)}))
    if prec == "bf16":
        for (mm, nn) in ((256, 256), (256, 768)):
            a16, b16 = torch.randn(n, mm, device=dev).bfloat16(), torch.randn(n, nn, device=dev).bfloat16()
            ms = timeit(lambda: ops.gemm(a16, b16, trans_a=True))
            print(json.dumps({"kernel": "gemm_tn bf16 x bf16", "M": mm, "N": nn, "K": n, "ms": round(ms, 4),
                              "GBs_io": round(2.0 * n * (mm + nn) / ms / 1e6)}))
    # R-GCN per-relation products over S ~ 3e5 (relation, target) segment rows, skewed group sizes
    s, g = 300_000, 30
    w = torch.arange(1, g + 1, dtype=torch.float64).pow(-1.2)
    sizes = (w / w.sum() * s).long()
    sizes[0] += s - sizes.sum()
    gptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(torch.int32).to(dev)
    h, wt, dy = torch.randn(s, 256, device=dev), torch.randn(g, 256, 256, device=dev), torch.randn(s, 256, device=dev)
    cases = [("grouped_rows", lambda: ops.gemm_grouped_rows(h, wt, gptr), 2.0 * s * 256 * 256, 8.0 * s * 256),
             ("grouped_rows_T", lambda: ops.gemm_grouped_rows(dy, wt, gptr, trans_b=True), 2.0 * s * 256 * 256, 8.0 * s * 256),
             ("grouped_k", lambda: ops.gemm_grouped_k(h, dy, gptr, g), 2.0 * s * 256 * 256, 8.0 * s * 256)]
    if prec == "bf16":
        h16, dy16 = h.bfloat16(), dy.bfloat16()
        cases.append(("grouped_k bf16 x bf16", lambda: ops.gemm_grouped_k(h16, dy16, gptr, g), 2.0 * s * 256 * 256,
                      4.0 * s * 256))
    for name, fn, flops, byt in cases:
        ms = timeit(fn)
        print(json.dumps({"kernel": name, "precision": prec, "S": s, "G": g, "ms": round(ms, 4),
                          "TFLOPs": round(flops / ms / 1e9, 1), "GBs_io": round(byt / ms / 1e6)}))


if __name__ == "__main__":
    main()
