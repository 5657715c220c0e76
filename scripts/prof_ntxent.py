#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats -- python3 scripts/prof_ntxent.py N h mode reps`: repeated NT-Xent
forward + backward at one size (the per-kernel averages of that run are the profile committed under profiles/)."""
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops  # noqa: E402

n, h, mode, reps = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
torch.manual_seed(0)
h1 = torch.randn(n, h, device="cuda", requires_grad=True)
h2 = (0.6 * h1.detach() + 0.4 * torch.randn(n, h, device="cuda")).requires_grad_()
for _ in range(reps):
    loss = ops.ntxent_loss(h1, h2, 0.2, operands=mode)
    loss.backward()
torch.cuda.synchronize()
print("loss", float(loss))
