"""Dev: the two KGE passes at bench scale by their HIP-event tags (ops.KERNEL_TIMERS)."""
import sys
import torch
sys.path.insert(0, ".")
from biomedkg_amd import ops, sampling, synthetic  # noqa: E402
ops.set_matmul_precision("bf16")
dev = "cuda"
g = synthetic.primekg_like(130_000, 4_000_000, 30, seed=42)
ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
torch.manual_seed(0)
z = (torch.randn(130_000, 256, device=dev) * 0.3).requires_grad_()
rel = (torch.randn(30, 256, device=dev) * 0.3).requires_grad_()
neg = sampling.negative_sampling(ei, 130_000, ei.size(1), seed=3)
nt = et[torch.randperm(et.numel(), device=dev)]
for it in range(4):
    loss, _, _ = ops.kge_bce_loss(z, rel, ei, et, neg, nt, 0)
    loss.backward(); z.grad = None; rel.grad = None
ops.KERNEL_TIMERS = {}
for it in range(10):
    loss, _, _ = ops.kge_bce_loss(z, rel, ei, et, neg, nt, 0)
    loss.backward()
    if it == 0:
        ref = z.grad.clone()
    z.grad = None; rel.grad = None
torch.cuda.synchronize()
for tag, rec in ops.KERNEL_TIMERS.items():
    d = [a.elapsed_time(b) for a, b in rec["events"]]
    print(tag, "pos / neg launches avg ms:", round(sum(d[0::2]) / len(d[0::2]), 4), round(sum(d[1::2]) / len(d[1::2]), 4), "checksum", float(ref.double().abs().sum()))
