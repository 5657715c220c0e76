#!/usr/bin/env python3
"""Degenerate-size sweep of the eager training steps: graphs of 2..60 nodes with 0..200 edges, 1..4 relations,
neg_ratio 0..50 (more negatives asked for than free cells), self loops, duplicate edges; KGE (DistMult / TransE) and
GGD / GRACE / DGI steps, fp32 and bf16 modes.  Allowed outcomes: a finite loss, NaN on a graph without edges (the mean
over no triplets, as in the reference), or a Python exception (the reference's own shape errors).  Not allowed: a GPU
memory fault (the process would die) or NaN / inf on a graph with edges.   python scripts/fuzz_tiny.py  (MI355X)"""
import random
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops, rand as R                                   # noqa: E402
from biomedkg_amd.gcl_module import DGIModule, GGDModule, GRACEModule     # noqa: E402
from biomedkg_amd.kge_module import KGEModule                             # noqa: E402

dev = "cuda"
random.seed(11)
bad = errs = 0
for it in range(120):
    n = random.choice([2, 3, 5, 10, 60])
    e = random.choice([0, 1, 2, 5, 20, 200])
    r = random.randint(1, 4)
    ratio = random.choice([0, 1, 3, 50])
    d = random.choice([16, 64])
    mode = random.choice(["fp32", "bf16"])
    g = torch.Generator().manual_seed(it)
    ei = torch.randint(0, n, (2, e), generator=g).to(dev)
    et = torch.randint(0, r, (e,), generator=g).to(dev)
    x = torch.randn(n, d, generator=g).to(dev)
    x3 = torch.randn(n, 2, d, generator=g).to(dev)
    ops.set_matmul_precision(mode)
    what = random.choice(["kge-dismult", "kge-transe", "ggd", "grace", "dgi"])
    try:
        torch.manual_seed(it)
        if what.startswith("kge"):
            m = KGEModule("rgcn", what[4:], d, d, d, 1, r, 2, "cosine", 1e-3, 0.0, "none", ratio, "random").to(dev).train()
            batch = SimpleNamespace(x=x, edge_index=ei, edge_type=et)
        else:
            cls = {"ggd": GGDModule, "grace": GRACEModule, "dgi": DGIModule}[what]
            m = cls(in_dim=d, hidden_dim=d, out_dim=d, num_hidden_layers=1, fuse_method="attention").to(dev).train()
            batch = SimpleNamespace(x=x3, edge_index=ei)
        with R.use(R.DeviceRand(seed=it)):
            loss = m.training_step(batch)
            loss.backward()
        torch.cuda.synchronize()
        val = float(loss)
        grads_ok = all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in m.model.parameters())
        ok = (val == val and abs(val) != float("inf") and grads_ok) or (e == 0 and what.startswith("kge"))
        note = f"loss {val:.4f}"
    except Exception as why:                                               # noqa: BLE001
        torch.cuda.synchronize()
        ok, note = True, "raised " + repr(why)[:90]
        errs += 1
    if not ok:
        bad += 1
    print("ok " if ok else "FAIL", it, what, dict(n=n, e=e, r=r, ratio=ratio, d=d, mode=mode), note, flush=True)
ops.set_matmul_precision("fp32")
print("fuzz_tiny done, failures:", bad, "raised:", errs)
