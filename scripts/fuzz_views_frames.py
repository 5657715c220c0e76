#!/usr/bin/env python3
"""Differential sweep of round 4's structure changes: random graphs (isolated nodes, self loops, duplicate edges, empty
relations, hubs), a few Adam steps of the GGD step (augmentation always on) and of the KGE step (R-GCN + DistMult / TransE),
WITH the dropout_edge views' own CSRs (graph.KeptCSR), the planted relation order and the sorted frame of the positives
against the same build with all three switched off.  Exact-fp32 mode, fixed draws: the first step's losses agree to 2e-6, every step's to 2e-4;
the parameters after four Adam steps to 5e-3 (Adam divides by sqrt(v): an entry whose tiny gradient changes sign with the
summation order moves by the learning rate either way).
python scripts/fuzz_views_frames.py [cases]   (MI355X)"""
import random
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, ".")
from biomedkg_amd import graph as G, ops, rand as R  # noqa: E402
from biomedkg_amd.gcl_module import GGDModule  # noqa: E402
from biomedkg_amd.kge_module import KGEModule  # noqa: E402

dev = "cuda"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
random.seed(7)
bad = 0
for it in range(cases):
    n = random.choice([50, 700, 4095, 4096, 6000, 20000])
    e = max(4, min(n * random.randint(1, 30), n * n // 8))        # (sparse enough for neg_ratio * e negatives to exist)
    r = random.randint(2, 9)
    g = torch.Generator().manual_seed(100 + it)
    ei = torch.randint(0, max(2, int(n * random.choice([1.0, 0.6]))), (2, e), generator=g)
    if random.random() < 0.5:
        ei[1, : e // 3] = 3                                   # a hub target (long rows)
    if random.random() < 0.5:
        k = max(1, e // 10)
        ei[:, e - k:] = ei[:, :k]                              # duplicate edges
    ei[:, : min(20, e)] = torch.arange(min(20, e)).remainder(n).repeat(2, 1)                # self loops
    et = torch.randint(0, max(1, r - random.choice([0, 1])), (e,), generator=g)
    ei, et = ei.to(dev), et.to(dev)
    xg = torch.randn(n, 768, generator=g).to(dev)
    xk = torch.randn(n, 256, generator=g).to(dev)
    dec = random.choice(["dismult", "transe"])
    out = {}
    short = False
    for on in (False, True):
        G.KEPT_CSR_MIN_EDGES = 0 if on else 1 << 60
        G.PLANT_RELATION_ORDER = on
        ops.SORTED_POSITIVES = on
        ops._SORTED_MIN_TRIPLETS = 1
        ops.set_matmul_precision("fp32")
        ops.set_deterministic(True)
        torch.manual_seed(it)
        gcl = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(dev).train()
        gcl.model.p = 1.0
        kge = KGEModule("rgcn", dec, 256, 256, 256, 2, r, 2, "cosine", 1e-3, 0.2, "none", 2, "random").to(dev).train()
        og = torch.optim.Adam(gcl.model.parameters(), lr=1e-3)
        ok_ = torch.optim.Adam(kge.model.parameters(), lr=1e-3)
        ei_run, et_run = ei.clone(), et.clone()
        losses = []
        try:
            with R.use(R.DeviceRand(seed=it)):
                torch.manual_seed(1000 + it)
                for step in range(4):
                    og.zero_grad(set_to_none=True)
                    lg = gcl.calculate_loss(xg, ei_run)
                    lg.backward(); og.step()
                    ok_.zero_grad(set_to_none=True)
                    lk = kge.training_step(SimpleNamespace(x=xk, edge_index=ei_run, edge_type=et_run))
                    lk.backward(); ok_.step()
                    losses.append((float(lg.detach()), float(lk.detach())))
        except ValueError as ex:      # a graph so dense that the sampler finds fewer non-edges than neg_ratio * E: the
            if "do not match" not in str(ex):      # reference's decode fails on the same shapes (kge_module.py:86-102, 121-124)
                raise
            short = True
            break
        out[on] = (losses, [p.detach().clone() for p in list(gcl.model.parameters()) + list(kge.model.parameters())])
    ops.set_deterministic(False)
    if short:
        print(f"case {it}: n={n} e={e} r={r} {dec}: fewer negatives than asked for (the reference raises too) -- skipped", flush=True)
        continue
    (la, pa), (lb, pb) = out[False], out[True]
    rel = [max(abs(a - b) / max(abs(a), 1e-6) for a, b in zip(x, y)) for x, y in zip(la, lb)]     # per step
    dl, d1 = max(rel), rel[0]
    # per tensor: the relative difference -- or, for a tensor that is still small (a zero-initialised bias four steps in: its
    # norm is a few learning rates), no entry further apart than half a learning rate (case 24 of a 150-case sweep: a bias of
    # norm 0.03, relative difference 9e-3, largest entry difference 2.7e-4)
    dps = [float((a - b).norm() / a.norm().clamp_min(1e-12)) if float((a - b).abs().max()) > 0.5e-3 or float(a.norm()) > 0.1
           else min(float((a - b).norm() / a.norm().clamp_min(1e-12)), 5e-3) for a, b in zip(pa, pb)]
    dp = max(dps)
    if dp > 5e-3:
        k = dps.index(dp)
        a, b = pa[k], pb[k]
        print(f"   worst tensor #{k} shape {tuple(a.shape)} norm {float(a.norm()):.3e}: max |diff| {float((a - b).abs().max()):.2e}",
              flush=True)
    # the first step differs by summation order alone; later steps carry what Adam makes of it (it divides by sqrt(v): an
    # entry whose tiny gradient changes sign with the order moves by the learning rate either way)
    flag = "" if (d1 <= 2e-6 and dl <= 2e-4 and dp <= 5e-3) else "   <-- MISMATCH"
    bad += bool(flag)
    print(f"case {it}: n={n} e={e} r={r} {dec}: loss rel diff first step {d1:.2e}, worst of 4 {dl:.2e}, param rel diff {dp:.2e}{flag}",
          flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
