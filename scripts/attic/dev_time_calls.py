"""Dev: every library call of one bench step timed on its own (events around the call, a sync per call), with its integer
arguments -- which entry points carry the step, at which shapes.  usage: dev_time_calls.py [bf16|fp32] [fresh]"""
import collections, sys, types
sys.path.insert(0, ".")
import torch
import bench
from biomedkg_amd import ops, _lib as L
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
args = types.SimpleNamespace(nodes=130_000, edges=4_000_000, relations=30, hidden=256, neg_ratio=1, precision=prec)
dev = torch.device("cuda", 0)
ops.set_matmul_precision(prec)
step_g, step_k, n, e, r = bench.build_replica_steps(args, 0, 1, dev)
for _ in range(3):
    step_g(); step_k()
torch.cuda.synchronize()
rows = []
orig = L.call
which = ["?"]


def timed(name, *a):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = orig(name, *a)
    e1.record()
    torch.cuda.synchronize()
    ints = [int(getattr(x, "value", x)) for x in a if isinstance(x, (int, bool)) or hasattr(x, "value") and isinstance(getattr(x, "value", None), int)]
    rows.append((which[0], name, e0.elapsed_time(e1), tuple(v for v in ints if v < (1 << 40))))
    return rc


L.call = timed
for m in (ops, __import__("biomedkg_amd.graph", fromlist=["x"]), __import__("biomedkg_amd.optim", fromlist=["x"]),
          __import__("biomedkg_amd.sampling", fromlist=["x"]), __import__("biomedkg_amd.rand", fromlist=["x"])):
    if hasattr(m, "L"):
        m.L.call = timed
fresh = len(sys.argv) > 2 and sys.argv[2] == "fresh"      # a new edge_index per step, as bench.py's fresh-structures figure
for rep in range(3):
    rows.clear()
    if fresh:
        ei = step_k.batch.edge_index.clone()
        step_k.batch.edge_index = step_g.batch.edge_index = ei
        step_k.batch.edge_type = step_k.batch.edge_type.clone()
    which[0] = "gcl"; step_g()
    which[0] = "kge"; step_k()
torch.cuda.synchronize()
tot = collections.defaultdict(float)
for w, nm, ms, ints in rows:
    tot[w] += ms
print({k: round(v, 3) for k, v in tot.items()}, "ms in library calls (each timed alone)")
agg = collections.OrderedDict()
for w, nm, ms, ints in rows:
    k = (w, nm, ints)
    c, t = agg.get(k, (0, 0.0))
    agg[k] = (c + 1, t + ms)
for (w, nm, ints), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{w} {t:7.3f} ms {c:2d}x {nm} {ints}")
