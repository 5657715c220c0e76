#!/usr/bin/env python3
"""The KGE step at the reference's default neg_ratio = 10 on the bench graph (44 M triplets per step), a few steps (for a
rocprofv3 --kernel-trace run: scripts/dev_trace_kge_neg10.sh lists its kernels per step)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from biomedkg_amd import ops, synthetic  # noqa: E402
from biomedkg_amd.kge_module import KGEModule  # noqa: E402
ops.set_matmul_precision("bf16")
dev = torch.device("cuda", 0)
n, e, r, h = 130_000, 4_000_000, 30, 256
g = synthetic.primekg_like(n, e, r, seed=42)
b = bench.Batch()
b.x, b.edge_index, b.edge_type = torch.randn(n, h, device=dev), g["edge_index"].to(dev), g["edge_type"].to(dev)
torch.manual_seed(1234)
m = KGEModule("rgcn", "dismult", h, h, h, 2, g["num_relations"], 2, "cosine", 1e-3, 0.2, "none", 10, "random").to(dev).train()
step = bench.make_step(m, b, 1)
for _ in range(3 + 5):
    step()
torch.cuda.synchronize()
