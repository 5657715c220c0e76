"""Dev: which call sites create zero-filled tensors in one bench step (torch.zeros / zeros_like / new_zeros / zero_ / fill_)."""
import collections, sys, traceback, types
sys.path.insert(0, ".")
import torch
import bench
from biomedkg_amd import ops
args = types.SimpleNamespace(nodes=130_000, edges=4_000_000, relations=30, hidden=256, neg_ratio=1, precision="bf16")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ops.set_matmul_precision("bf16")
step_g, step_k, n, e, r = bench.build_replica_steps(args, 0, 1, dev)
for _ in range(3):
    step_g(); step_k()
torch.cuda.synchronize()
sites = collections.Counter()


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "/biomedkg_amd/" in fr.filename or fr.filename.endswith("bench.py"):
            return f"{fr.filename.split('/')[-1]}:{fr.lineno} {fr.line[:70]}"
    return "?"


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        sites[(name, site())] += 1
        return orig(*a, **k)
    setattr(mod, name, f)


for nm in ("zeros", "zeros_like", "full", "ones", "ones_like", "full_like"):
    wrap(torch, nm)
for nm in ("zero_", "fill_", "new_zeros", "new_full", "new_ones"):
    orig = getattr(torch.Tensor, nm)

    def mk(orig=orig, nm=nm):
        def f(self, *a, **k):
            sites[(nm, site())] += 1
            return orig(self, *a, **k)
        return f
    setattr(torch.Tensor, nm, mk())
step_g(); step_k()
torch.cuda.synchronize()
for (nm, s), c in sites.most_common(40):
    print(c, nm, s)
