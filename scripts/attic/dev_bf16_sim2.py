"""Dev: one R-GCN layer of the bf16 path, piece by piece, against tests/bf16_model.py's rounding model."""
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import graph as G, ops, synthetic  # noqa: E402
from tests import bf16_model as B  # noqa: E402

DEV = "cuda"
n, r, d = 130_000, 30, 256
g = synthetic.primekg_like(n, 4_000_000, r, seed=42)
ei, et = g["edge_index"].to(DEV), g["edge_type"].to(DEV)
torch.manual_seed(42)
x = torch.randn(n, d, device=DEV)
weight = torch.randn(r, d, d, device=DEV) * 0.06
root = torch.randn(d, d, device=DEV) * 0.06
bias = torch.randn(d, device=DEV) * 0.05
ops.set_matmul_precision("bf16")
rg = G.rel_graph_of(ei, et, n, r)
x16 = ops.to_bf16(x)


def err(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


# pieces as ops._RgcnConv.forward runs them
h = ops.spmm(rg.seg_ptr, rg.src_sorted, rg.inv_cnt, x16, rg.num_segments, split=rg.seg_split, out_dtype=torch.bfloat16)
y = ops.gemm_grouped_rows(h, weight, rg.rel_ptr, out_dtype=torch.bfloat16)
xr16 = ops.gemm(x16, root, out_dtype=torch.bfloat16)
xr32 = ops.gemm(x16, root)
c = rg.dst_csr
out_h = ops.spmm(c.rowptr, c.col, None, y, n, addend=xr16, bias=bias, relu=True, split=c.split, out_dtype=torch.bfloat16)
out_l = ops.spmm(c.rowptr, c.col, None, y, n, addend=xr32, bias=bias, relu=False, split=c.split, out_dtype=torch.float32)

# the model
src, dst = ei[0], ei[1]
key = et * n + dst
uniq, seg_of_edge, cnt = torch.unique(key, return_inverse=True, return_counts=True)
seg_dst, seg_rel = uniq % n, uniq // n
inv_cnt = 1.0 / cnt[seg_of_edge].double()
xd = B.bf(x)
hm = B.bf(B.aggregate(xd, src, seg_of_edge, inv_cnt, None, uniq.numel()))
print("segments", rg.num_segments, uniq.numel(), "seg order equal", bool((rg.seg_dst.long() == seg_dst).all()))
print("h (segment means)", err(h, hm))
w16 = B.bf(weight)
counts = torch.bincount(seg_rel, minlength=r).tolist()
parts, a = [], 0
for rr, cc in enumerate(counts):
    if cc:
        parts.append(hm[a:a + cc] @ w16[rr])
    a += cc
ym = B.bf(torch.cat(parts))
print("y (per-relation products)", err(y, ym), " with the HIP h as input:", end=" ")
parts, a = [], 0
for rr, cc in enumerate(counts):
    if cc:
        parts.append(h.double()[a:a + cc] @ w16[rr])
    a += cc
print(err(y, B.bf(torch.cat(parts))))
xrm = xd @ B.bf(root)
print("xr bf16", err(xr16, B.bf(xrm)), "xr fp32", err(xr32, xrm))
iota = torch.arange(uniq.numel(), device=DEV)
ones = torch.ones(uniq.numel(), dtype=torch.float64, device=DEV)
pre_h = B.aggregate(ym, iota, seg_dst, ones, None, n) + B.bf(xrm) + bias.double()
pre_l = B.aggregate(ym, iota, seg_dst, ones, None, n) + xrm + bias.double()
print("out hidden", err(out_h, B.bf(torch.relu(pre_h))), "out last", err(out_l, pre_l))
pre_h2 = B.aggregate(y.double(), iota, seg_dst, ones, None, n) + xr16.double() + bias.double()
print("out hidden from the HIP y / xr", err(out_h, B.bf(torch.relu(pre_h2))))
