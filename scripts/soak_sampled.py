import sys, torch
sys.path.insert(0, "/root/repo")
from biomedkg_amd import synthetic, ops
from biomedkg_amd.kge_module import KGEModule
from biomedkg_amd.sampled import SaintGraphedKge
det, steps = sys.argv[1] == "1", int(sys.argv[2])
ops.set_deterministic(det)
dev = "cuda"
g = synthetic.primekg_like(130_000, 4_000_000, 30, seed=42)
ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
x = torch.randn(130_000, 256, device=dev) * 0.01
torch.manual_seed(42)
m = KGEModule("rgcn", "dismult", 256, 256, 256, 2, 30, 2, "cosine", 1e-3, 0.0, "none", 10, "random").to(dev).train()
walks = int(sys.argv[3]) if len(sys.argv) > 3 else 128
sg = SaintGraphedKge(m, x, ei, et, walks, 10, None, clip=1.0, seed=42)
import time
t0 = time.perf_counter()
for i in range(steps):
    l = sg.step()
    if i % 10 == 0: v = float(l)
torch.cuda.synchronize()
print("det", det, "steps", steps, "ms/step", round((time.perf_counter() - t0) / steps * 1e3, 3), "loss", v, dict(sg.stats))
