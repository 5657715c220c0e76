#!/usr/bin/env python3
"""BASELINE configs[1]: the GRACE step on the 5k / 50k graph as hipGraph replays (for rocprofv3 traces).  Prints ms/step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biomedkg_amd import graphs, synthetic
from biomedkg_amd.gcl_module import GRACEModule
from types import SimpleNamespace
dev = torch.device("cuda:0")
b = SimpleNamespace(x=synthetic.lm_features(5000, 2, 768, seed=42).to(dev), edge_index=synthetic.gcl_small(5000, 50_000, seed=42).to(dev))
torch.manual_seed(1)
mod = GRACEModule(in_dim=768, hidden_dim=128, out_dim=128, num_hidden_layers=2, fuse_method="attention").to(dev).train()
g = graphs.GraphedStep(mod, b, torch.optim.Adam(mod.model.parameters(), lr=1e-3, fused=True), clip=1.0, warmup=3)
for _ in range(5):
    g()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for _ in range(n):
    g()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / n * 1e3)
