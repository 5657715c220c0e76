"""Dev: how fast would the node-side KGE backward be as a plain weighted segment reduce (the SpMM kernels over the
incidence lists, relation factor left out)?  Upper bound for a stream-based kge_bwd_nodes."""
import sys
sys.path.insert(0, ".")
import torch
from biomedkg_amd import ops, synthetic, graph as G
ops.set_matmul_precision("bf16")
dev = "cuda"
g = synthetic.primekg_like(130_000, 4_000_000, 30, seed=42)
ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
n, d = 130_000, 256
inc = ops.Incidence(ei, n)
t = ei.size(1)
perm = inc.perm.long()
other = torch.where(perm < t, ei[1][perm.clamp(max=t - 1)], ei[0][(perm - t).clamp(min=0)]).to(torch.int32)
w = torch.randn(2 * t, device=dev)
z16 = torch.randn(n, d, device=dev).bfloat16()
split = G.rowsplit_for(inc.rowptr, n, 2 * t, pack=True, col=other)


def tm(fn, reps=20):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print("spmm over the incidence lists, d=256 bf16 table, fp32 out: %.3f ms" % tm(lambda: ops.spmm(inc.rowptr, other, w, z16, n, split=split)))
