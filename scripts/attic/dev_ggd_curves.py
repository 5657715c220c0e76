#!/usr/bin/env python3
"""GGD loss curves over 20 Adam steps: exact fp32 (Linear + sum head), bf16 side-by-side path with the head as a dot
product per row, and the same with the head forced back to Linear + sum.  python scripts/dev_ggd_curves.py"""
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops, rand as R            # noqa: E402
from biomedkg_amd.gcl_module import GGDModule      # noqa: E402

dev = "cuda"
n, e = 6000, 100_000
g = torch.Generator().manual_seed(12)
ei = torch.randint(0, n, (2, e), generator=g).to(dev)
x = torch.randn(n, 768, generator=g).to(dev)
ops.set_deterministic(True)
ok = ops.rowsum_linear_ok
for name, mode, head in (("fp32", "fp32", True), ("bf16 rowdot", "bf16", True), ("bf16 linear+sum", "bf16", False)):
    ops.set_matmul_precision(mode)
    ops.rowsum_linear_ok = ok if head else (lambda *a: False)
    torch.manual_seed(0)
    m = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(dev).train()
    m.model.encoder.drop_out = False
    m.model.p = 0.0
    opt = torch.optim.Adam(m.model.parameters(), lr=1e-3)
    losses = []
    with R.use(R.DeviceRand(seed=5)):
        for _ in range(20):
            opt.zero_grad(set_to_none=True)
            loss = m.calculate_loss(x, ei)
            loss.backward()
            opt.step()
            losses.append(round(float(loss.detach()), 4))
    print(name, losses)
