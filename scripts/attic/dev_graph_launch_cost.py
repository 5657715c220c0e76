#!/usr/bin/env python3
"""Dev: what ONE replayed NeighborLoader GGD step costs the HOST when the GPU is idle (no queue back-pressure): the step's
host time with a synchronise before every call, against the free-running step time.  python3 scripts/dev_graph_launch_cost.py [fp32|bf16]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops, synthetic  # noqa: E402
from biomedkg_amd.gcl_module import GGDModule  # noqa: E402
from biomedkg_amd.sampled_gcl import NeighborGraphedGcl  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda:0")
ops.set_matmul_precision(mode)
n, e = 27_000, 640_000
ei = synthetic.gcl_small(n, e, seed=1).to(dev)
x = synthetic.lm_features(n, 2, 768, seed=2).to(dev)
torch.manual_seed(0)
m = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="attention").to(dev).train()
lp = NeighborGraphedGcl(m, x, ei, 128, (30, 30, 30), clip=1.0, seed=3)
for _ in range(120):
    lp.step()
torch.cuda.synchronize()
host = []
for _ in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lp.step()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    lp.step()
torch.cuda.synchronize()
free = (time.perf_counter() - t0) / 100
host.sort()
print(f"{mode}: host time of one step with an idle GPU: median {host[len(host) // 2] * 1e3:.3f} ms (min {host[0] * 1e3:.3f}); "
      f"free-running step {free * 1e3:.3f} ms = {1 / free:.0f} steps/s; stats {lp.stats}")
lp.close()
