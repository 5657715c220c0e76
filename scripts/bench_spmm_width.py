import sys, torch
sys.path.insert(0, ".")
from biomedkg_amd import ops, synthetic, graph as G
ops.set_matmul_precision("bf16")
dev = "cuda"
g = synthetic.primekg_like(130_000, 4_000_000, 30, seed=42)
ei = g["edge_index"].to(dev)
gr = G.Graph(ei, 130_000)
gw = gr.gcn_weights()
c = gr.by_dst
def t(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for d in (64, 128, 256, 512, 1024):
    x = torch.randn(130_000, d, device=dev).bfloat16()
    us = t(lambda: ops.spmm(c.rowptr, c.col, gw.w_dst, x, 130_000, self_w=gw.self_w, split=c.split, out_dtype=torch.bfloat16))
    print(d, round(us, 1), "us", round(4e6 * d * 2 / us / 1e6, 2), "TB/s gathered")
