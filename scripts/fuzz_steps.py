#!/usr/bin/env python3
"""Differential sweep: random graphs (sizes around the bf16 fast-path thresholds, empty relations, isolated nodes),
KGE step (R-GCN + DistMult / TransE) and GGD step in bf16 mode against the exact-fp32 mode of the same build.
python scripts/fuzz_steps.py  (MI355X)"""
import random
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops, rand as R  # noqa: E402
from biomedkg_amd.gcl_module import GGDModule  # noqa: E402
from biomedkg_amd.kge_module import KGEModule  # noqa: E402

dev = "cuda"
random.seed(1)
bad = 0
for it in range(24):
    n = random.choice([700, 4095, 4096, 5000, 9000, 20000])
    e = n * random.randint(1, 25)
    r = random.randint(2, 12)
    d = random.choice([256, 256, 128])
    g = torch.Generator().manual_seed(it)
    ei = torch.randint(0, max(2, int(n * random.choice([1.0, 0.7]))), (2, e), generator=g).to(dev)   # 0.7: isolated tail nodes
    et = torch.randint(0, max(1, r - random.choice([0, 1])), (e,), generator=g).to(dev)              # maybe an empty relation
    xk = torch.randn(n, d, generator=g).to(dev)
    xg = torch.randn(n, 768, generator=g).to(dev)
    dec = random.choice(["dismult", "transe"])
    res = {}
    for mode in ("fp32", "bf16"):
        ops.set_matmul_precision(mode)
        torch.manual_seed(0)
        kge = KGEModule("rgcn", dec, d, d, d, 2, r, 1, "cosine", 1e-3, 0.1, "none", 1, "random").to(dev).train()
        gcl = GGDModule(in_dim=768, hidden_dim=d, out_dim=d, num_hidden_layers=2, fuse_method="none").to(dev).train()
        gcl.model.p = 0.0      # no augmentation: its feature mask is drawn by a different generator in bf16 mode
        gcl.model.encoder.drop_out = False   # bf16 mode runs both views side by side: other dropout element ids
        out = []
        for mod, batch in ((kge, SimpleNamespace(x=xk, edge_index=ei, edge_type=et)),
                           (gcl, SimpleNamespace(x=xg, edge_index=ei))):
            torch.manual_seed(123)       # same permutations / negatives' relation shuffle in both modes
            with R.use(R.DeviceRand(seed=5)):
                loss = mod.training_step(batch, 0)
            loss.backward()
            out.append((float(loss.detach()), [p.grad.clone() for p in mod.model.parameters() if p.grad is not None]))
        res[mode] = out
    for (lf, gf), (lb, gb), name in zip(res["fp32"], res["bf16"], ("kge", "ggd")):
        ok = abs(lf - lb) <= 3e-2 * max(1.0, abs(lf)) and all(torch.isfinite(t).all() for t in gb)
        # dropout masks differ between the two runs only through the hidden dtype (same Philox streams): compare norms
        rel = max(float((a - b).norm() / a.norm().clamp_min(1e-9)) for a, b in zip(gf, gb))
        tol = 0.35 if dec == "transe" and name == "kge" else 0.12
        if not ok or rel > tol:
            bad += 1
            print("FAIL", name, dict(n=n, e=e, r=r, d=d, dec=dec), lf, lb, rel)
print("fuzz_steps done, failures:", bad)
