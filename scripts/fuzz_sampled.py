#!/usr/bin/env python3
"""Randomised sweep of the padded, replayed GraphSAINT KGE loop (biomedkg_amd/sampled.py): graph sizes / densities,
relation counts (incl. empty relations), walk counts, neg_ratio (incl. 0 = the positives' own relation ids), decoders,
sinks and isolated nodes.  Every step must give a finite loss, and the replayed loop must track the eager padded loop.
python scripts/fuzz_sampled.py  (MI355X)"""
import random
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd.kge_module import KGEModule            # noqa: E402
from biomedkg_amd.sampled import SaintGraphedKge         # noqa: E402

dev = "cuda"
random.seed(3)
bad = 0
for it in range(20):
    n = random.choice([300, 1000, 5000, 20000])
    e = int(n * random.choice([0.5, 2, 8, 30]))
    r = random.randint(1, 12)
    walks = random.choice([4, 16, 64])
    ratio = random.choice([0, 1, 5])
    dec = random.choice(["dismult", "transe"])
    d = random.choice([32, 64])
    g = torch.Generator().manual_seed(it)
    hi = max(2, int(n * random.choice([1.0, 0.6])))      # 0.6: a tail of isolated nodes / sinks
    ei = torch.randint(0, hi, (2, e), generator=g).to(dev)
    et = torch.randint(0, max(1, r - random.choice([0, 1])), (e,), generator=g).to(dev)
    x = (torch.randn(n, d, generator=g) * 0.3).to(dev)
    runs = {}
    only = [int(a) for a in sys.argv[1:] if a.isdigit()]
    if only and it not in only:
        continue
    print("case", dict(it=it, n=n, e=e, r=r, walks=walks, ratio=ratio, dec=dec, d=d, hi=hi), flush=True)
    try:
        for graphs in (True, False):
            torch.manual_seed(0)
            m = KGEModule("rgcn", dec, d, d, d, 1, r, 2, "cosine", 1e-3, 0.0, "none", ratio, "random").to(dev).train()
            sg = SaintGraphedKge(m, x, ei, et, walks, 10, None, clip=1.0, seed=it, graphs=graphs,
                                 buckets=(64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384))
            runs[graphs] = ([float(sg.step()) for _ in range(25)], dict(sg.stats))
        a, b = torch.tensor(runs[True][0]), torch.tensor(runs[False][0])
        # (a batch without a single edge gives NaN -- the mean over no triplets -- here as in the reference)
        both = torch.isfinite(a) & torch.isfinite(b)
        ok = bool((torch.isfinite(a) == torch.isfinite(b)).all()) and int(both.sum()) >= 20 and \
            float((a[both] - b[both]).abs().max()) < 5e-3
    except ValueError as why:                               # fewer negatives than neg_ratio * E on a tiny dense batch:
        ok, runs = "do not match" in str(why), {True: ([], repr(why)[:120])}    # refused, as the reference fails there
    except Exception as why:                                # noqa: BLE001
        ok, runs = False, {"error": repr(why)[:300]}
    import gc                       # a failed case leaves its loop in a reference cycle with the traceback: collect it now,
    sg = m = None                   # not in the middle of the next case's graph capture
    if "--nogc" not in sys.argv:
        gc.collect(); torch.cuda.synchronize()
    if not ok:
        bad += 1
        print("FAIL", dict(it=it, n=n, e=e, r=r, walks=walks, ratio=ratio, dec=dec, d=d), runs, flush=True)
    else:
        print("ok", it, n, e, r, walks, ratio, dec, runs[True][1], flush=True)
print("fuzz_sampled done, failures:", bad)
