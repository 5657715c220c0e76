#!/usr/bin/env python3
"""Dev: the per-relation products of the bench graph's R-GCN layers (gemm_xw8_grouped_kernel): contiguous rows (forward)
and gathered rows (backward), against the plain 8-wave product over the same number of rows."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from biomedkg_amd import ops, synthetic, graph as G
dev = "cuda"
ops.set_matmul_precision("bf16")
g = synthetic.primekg_like(130_000, 4_000_000, 30, seed=42)
ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
rg = G.rel_graph_of(ei, et, 130_000, 30)
S = rg.num_segments
print("segments", S, "rel_ptr", rg.rel_ptr.tolist()[:8], "...")
h = torch.randn(S, 256, device=dev).bfloat16()
w = torch.randn(30, 256, 256, device=dev) * 0.05
dout = torch.randn(130_000, 256, device=dev).bfloat16()
def t(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("grouped contiguous  %.1f us" % t(lambda: ops.gemm_grouped_rows(h, w, rg.rel_ptr, out_dtype=torch.bfloat16)))
print("grouped gathered    %.1f us" % t(lambda: ops.gemm_grouped_rows(dout, w, rg.rel_ptr, trans_b=True, a_rows=rg.seg_dst, out_dtype=torch.bfloat16)))
print("plain same rows     %.1f us" % t(lambda: ops.gemm(h, w[0], out_dtype=torch.bfloat16)))
sizes = (rg.rel_ptr[1:] - rg.rel_ptr[:-1]).tolist()
print("group sizes", sorted(sizes, reverse=True)[:6], "min", min(sizes))
