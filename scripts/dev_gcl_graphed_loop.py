"""Dev: the graphed NeighborLoader GGD loop alone (for rocprofv3 --kernel-trace --stats).  argv: precision steps graphs(0/1)"""
import sys, time
sys.path.insert(0, ".")
import torch
from biomedkg_amd import ops, synthetic
from biomedkg_amd.gcl_module import GGDModule
from biomedkg_amd.sampled_gcl import NeighborGraphedGcl
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mode = sys.argv[3] if len(sys.argv) > 3 else "1"       # 1: graphed, 0: eager padded, 2: the ordinary eager loop
graphs = mode == "1"
dev = torch.device("cuda", 0)
n, e = 27_000, 640_000
ops.set_matmul_precision(prec)
ei = synthetic.gcl_small(n, e, seed=1).to(dev)
x = synthetic.lm_features(n, 2, 768, seed=2).to(dev)
torch.manual_seed(0)
m = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="attention").to(dev).train()
if mode == "2":
    from biomedkg_amd.samplers import NeighborBatchLoader
    from biomedkg_amd.optim import clip_adam_step
    opt = torch.optim.Adam(list(m.model.parameters()), lr=1e-3, fused=True)
    it = iter(NeighborBatchLoader(x, ei, None, 128, (30, 30, 30), True, seed=3))

    def run(k):
        for _ in range(k):
            b = next(it)
            opt.zero_grad(set_to_none=True)
            m.training_step(b).backward()
            clip_adam_step(opt, 1.0)
    run(60)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    print(prec, "ordinary eager ms/step", (time.perf_counter() - t0) / steps * 1e3, flush=True)
    sys.exit(0)
lp = NeighborGraphedGcl(m, x, ei, 128, (30, 30, 30), clip=1.0, seed=3, graphs=graphs)
for _ in range(60):
    lp.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    lp.step()
torch.cuda.synchronize()
print(prec, "graphs" if graphs else "eager padded", "ms/step", (time.perf_counter() - t0) / steps * 1e3, lp.stats, flush=True)
lp.close()
