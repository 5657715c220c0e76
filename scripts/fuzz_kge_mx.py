#!/usr/bin/env python3
"""Randomised shapes for the negatives' one-pass walk (csrc/kge_onepass_mx.hip) THROUGH the op: ``ops.kge_bce_loss(...,
lazy_scores=True)`` with the walk on against the same call with the negatives through scorer + list walk -- node counts
that are not multiples of four, 1..32 relations, hubs (lists of 10^5 entries), nodes without negatives, self pairs, negatives
several times the positives, a non-unit upstream gradient -- loss, dz, d rel_emb, the on-demand scores; sets with hubs (not
marked as the uniform sampler's) must stay on the two-pass form.
python scripts/fuzz_kge_mx.py [cases]  (MI355X)"""
import random
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops  # noqa: E402

dev = "cuda"
ops.set_matmul_precision("bf16")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
random.seed(0)
bad = 0
for it in range(cases):
    n = random.choice([1001, 4097, 20000, 65537, 130000, 200003])
    t_pos = random.choice([1 << 20, 1_200_000])
    t_neg = random.choice([1 << 20, 1_500_000, 3_000_000])
    r = random.choice([1, 2, 7, 30, 32])
    hub = random.random() < 0.5
    g = torch.Generator(device=dev).manual_seed(1000 + it)
    pe = torch.randint(0, n, (2, t_pos), device=dev, generator=g)
    pt = torch.randint(0, r, (t_pos,), device=dev, generator=g)
    hi = max(n - random.choice([0, 1, 5, n // 3]), 1)                       # (the last nodes carry no negative)
    ne = torch.randint(0, hi, (2, t_neg), device=dev, generator=g)
    if hub:
        ne[0][torch.rand(t_neg, device=dev, generator=g) < 0.05] = 3
        ne[1][torch.rand(t_neg, device=dev, generator=g) < 0.03] = 3
    nt = torch.randint(0, r, (t_neg,), device=dev, generator=g)
    if not hub:
        ne._bkg_exchangeable = True        # (what the library's uniform sampler marks its output with; a set with hubs -- not
                                           # marked -- must keep the two-pass form, whose row table cuts long lists into slices)
    z0, rel0 = torch.randn(n, 256, device=dev, generator=g) * 0.3, torch.randn(r, 256, device=dev, generator=g) * 0.3
    ops.onepass_stable(pe, pt)
    got = []
    for mx in (True, False):
        ops.KGE_ONEPASS_MX = mx
        z, rel = z0.clone().requires_grad_(), rel0.clone().requires_grad_()
        loss, pp, npred, inv = ops.kge_bce_loss(z, rel, pe, pt, ne, nt, ops.KGE_DISTMULT, reg=1e-2, lazy_scores=True)
        (loss * 0.7).backward()
        lazy = isinstance(npred, ops.LazyScores)
        npred = npred() if lazy else npred
        got.append((float(loss.detach()), npred, z.grad.clone(), rel.grad.clone(), lazy))
    ops.KGE_ONEPASS_MX = True
    (la, na, za, ra, lazy_a), (lb, nb, zb, rb, lazy_b) = got
    e_loss = abs(la - lb) / abs(lb)
    e_z = float((za - zb).norm() / zb.norm())
    e_r = float((ra - rb).norm() / rb.norm())
    e_s = float((na - nb).abs().max())
    ok = (lazy_a == (not hub)) and not lazy_b and e_loss < 2e-6 and e_z < 1e-5 and e_r < 3e-3 and e_s < 1e-3 and bool(torch.isfinite(za).all())
    bad += not ok
    print(("ok  " if ok else "BAD ") + f"{it}: n={n} t_pos={t_pos} t_neg={t_neg} r={r} hub={hub} loss {e_loss:.1e} dz {e_z:.1e} drel {e_r:.1e} "
          f"scores {e_s:.1e}", flush=True)
print(f"fuzz_kge_mx done, failures: {bad}")
sys.exit(1 if bad else 0)
