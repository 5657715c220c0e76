"""Dev: NT-Xent forward / backward times at N = 130,000, h = 256, bf16 operands (and the C2 size in fp32), HIP events."""
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops  # noqa: E402


def ev(fn, reps):
    fn()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b))
    return sorted(out)[len(out) // 2]


for n, h, mode, reps in ((130_000, 256, "bf16", 3), (5000, 128, "fp32", 10)):
    torch.manual_seed(0)
    h1 = torch.randn(n, h, device="cuda", requires_grad=True)
    h2 = (0.6 * h1.detach() + 0.4 * torch.randn(n, h, device="cuda")).requires_grad_()
    fwd = ev(lambda: ops.ntxent_loss(h1, h2, 0.2, operands=mode), reps)
    loss = ops.ntxent_loss(h1, h2, 0.2, operands=mode)
    bwd = ev(lambda: torch.autograd.grad(loss, (h1, h2), retain_graph=True), reps)
    m = 2 * n
    peak = 2500.0 if mode == "bf16" else 155.0
    print(f"N={n} h={h} {mode}: fwd {fwd:.3f} ms = {m * m * h / fwd / 1e9 / peak:.3f} of peak, bwd {bwd:.3f} ms = "
          f"{4.0 * m * m * h / bwd / 1e9 / peak:.3f} of peak, loss {float(loss):.6f}", flush=True)
