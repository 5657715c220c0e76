"""What do the hub rows cost a gather?  The bench graph's propagate (d = 256 / 512 bf16 tables) as it is, and with the edges
out of its top-K source nodes removed (they are 8 / 15 / 19 / 24 % of the edges for K = 16 / 64 / 128 / 256): if the time
falls by less than the edges do, the hubs' rows -- L2 hits -- are already the cheap part of the gather and keeping them in
LDS has little to win.  python scripts/dev_hub_share.py  (MI355X)"""
import sys, torch
sys.path.insert(0, ".")
from biomedkg_amd import ops, synthetic, graph as G
ops.set_matmul_precision("bf16")
dev = "cuda"
n = 130_000
g = synthetic.primekg_like(n, 4_000_000, 30, seed=1)
ei0 = g["edge_index"].to(dev)


def t(fn, reps=30):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


deg = torch.bincount(ei0[0], minlength=n)
order = deg.argsort(descending=True)
for d in (256, 512):
    x = torch.randn(n, d, device=dev).bfloat16()
    base = None
    for k in (0, 16, 64, 128, 256, 1024):
        hub = torch.zeros(n, dtype=torch.bool, device=dev)
        hub[order[:k]] = True
        ei = ei0[:, ~hub[ei0[0]]].contiguous()
        gr = G.Graph(ei, n)
        gw, c = gr.gcn_weights(), gr.by_dst
        us = t(lambda: ops.spmm(c.rowptr, c.col, gw.w_dst, x, n, self_w=gw.self_w, split=c.split, out_dtype=torch.bfloat16))
        base = base or (us, ei.size(1))
        print(f"d={d} without the top {k:5d} sources: {ei.size(1):8d} edges ({ei.size(1) / base[1]:.3f}), {us:7.1f} us ({us / base[0]:.3f})", flush=True)
