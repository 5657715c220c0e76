#!/usr/bin/env python3
"""NT-Xent bf16 forward / backward at N = 130,000, h = 256 by HIP events (the figure bench.py reports), several rounds."""
import sys
import torch
sys.path.insert(0, ".")
from biomedkg_amd import ops  # noqa: E402
n, h = 130_000, 256
torch.manual_seed(0)
h1 = torch.randn(n, h, device="cuda", requires_grad=True)
h2 = (0.6 * h1.detach() + 0.4 * torch.randn(n, h, device="cuda")).requires_grad_()
fw, bw = [], []
for i in range(6):
    a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    a.record(); loss = ops.ntxent_loss(h1, h2, 0.2, operands="bf16"); b.record(); loss.backward(); c.record()
    torch.cuda.synchronize()
    if i:
        fw.append(a.elapsed_time(b)); bw.append(b.elapsed_time(c))
m = 2 * n
print(f"loss {float(loss):.6f} fwd {min(fw):.3f} ms ({m * m * h / min(fw) / 1e9 / 2500:.3f} of 2.5 PF)  bwd {min(bw):.3f} ms ({4 * m * m * h / min(bw) / 1e9 / 2500:.3f})")
