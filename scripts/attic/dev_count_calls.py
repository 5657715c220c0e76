"""Dev: library calls per bench step (by entry point)."""
import collections, sys, types
sys.path.insert(0, ".")
import torch
import bench
from biomedkg_amd import ops, _lib as L
args = types.SimpleNamespace(nodes=130_000, edges=4_000_000, relations=30, hidden=256, neg_ratio=1, precision="bf16")
dev = torch.device("cuda", 0)
ops.set_matmul_precision("bf16")
step_g, step_k, n, e, r = bench.build_replica_steps(args, 0, 1, dev)
for _ in range(3):
    step_g(); step_k()
torch.cuda.synchronize()
calls = collections.Counter()
orig = L.call


def counted(name, *a):
    calls[name] += 1
    return orig(name, *a)


L.call = counted
for m in (ops, __import__("biomedkg_amd.graph", fromlist=["x"]), __import__("biomedkg_amd.optim", fromlist=["x"]),
          __import__("biomedkg_amd.sampling", fromlist=["x"])):
    if hasattr(m, "L"):
        m.L.call = counted
step_g(); step_k()
torch.cuda.synchronize()
for k, v in calls.most_common(40):
    print(v, k)
