"""Dev: the fused scorer pass over the positives as they come (relation order = a permutation) against the same triplets
stored in relation order (order = identity, ids read in place)."""
import sys, torch
sys.path.insert(0, ".")
from biomedkg_amd import ops, synthetic
ops.set_matmul_precision("bf16")
dev = "cuda"
g = synthetic.primekg_like(130_000, 4_000_000, 30, seed=42)
ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
n, d, r = 130_000, 256, 30
z = (torch.randn(n, d, device=dev) * 0.3).requires_grad_()
rel = (torch.randn(r, d, device=dev) * 0.3).requires_grad_()
order = torch.sort(et, stable=True).indices
ei_s, et_s = ei[:, order].contiguous(), et[order].contiguous()
neg = torch.randint(0, n, (2, 1000), device=dev)
nt = torch.randint(0, r, (1000,), device=dev)


def run(a, b, reps=10):
    ops.KERNEL_TIMERS = {}
    for _ in range(reps):
        loss = ops.kge_bce_loss(z, rel, a, b, neg, nt, ops.KGE_DISTMULT)[0]
        loss.backward()
    torch.cuda.synchronize()
    out = {}
    for tag, rec in ops.KERNEL_TIMERS.items():
        ds = sorted(x.elapsed_time(y) for x, y in rec["events"])
        out[tag] = (round(ds[len(ds) // 2], 4), round(max(ds), 4), len(ds))
    ops.KERNEL_TIMERS = None
    return float(loss), out


for _ in range(2):
    run(ei, et, 3); run(ei_s, et_s, 3)
print("as given ", run(ei, et))
print("rel order", run(ei_s, et_s))
