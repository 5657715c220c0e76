#!/usr/bin/env python3
"""Randomised shapes for DistMult's node-side backward: the stream form (bkg_kge_incidence_streams + bkg_kge_bwd_stream_f32)
against the one-pass form (bkg_kge_bwd_nodes_f32, the one the goldens pin) on the same inputs -- hub nodes (lists cut into
slices + fix-up), empty lists, every width the pieces take, few / many relations, both table types.
python scripts/fuzz_kge_bwd.py [cases]  (MI355X)"""
import random
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops

dev = "cuda"
random.seed(1)
torch.manual_seed(1)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
taken = [0]
_call = ops.L.call


def _counting(name, *a):
    taken[0] += name == "bkg_kge_bwd_stream_f32"
    return _call(name, *a)


ops.L.call = _counting
for it in range(cases):
    prec = random.choice(["bf16", "fp32"])
    ops.set_matmul_precision(prec)
    n = random.choice([4100, 5000, 9001, 20000, 66000])
    d = random.choice([32, 64, 128, 256])
    r = random.choice([1, 2, 7, 30, 60])
    t = random.choice([300_000, 400_001, 777_777])
    hubs = random.choice([0, 1, 3])
    def triplets(seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        ei = torch.randint(0, n, (2, t), device=dev, generator=g)
        if hubs:                                   # a few nodes on a third of the entries: rows of 10^5 entries
            m = torch.rand(t, device=dev, generator=g) < 0.33
            ei[random.randint(0, 1)][m] = torch.randint(0, hubs, (int(m.sum()),), device=dev, generator=g)
        ei[:, : t // 50] = ei[:, : t // 50] % (n // 2)         # and the upper half of the nodes a little emptier
        return ei, torch.randint(0, r, (t,), device=dev, generator=g)
    pe, pt = triplets(10 * it)
    ne, nt = triplets(10 * it + 1)
    z0 = torch.randn(n, d, device=dev) * 0.3
    rel0 = torch.randn(r, d, device=dev) * 0.3
    grads = []
    for streams in (False, True):
        ops.KGE_BWD_STREAMS = streams
        z, rel = z0.clone().requires_grad_(), rel0.clone().requires_grad_()
        loss = ops.kge_bce_loss(z, rel, pe.clone(), pt.clone(), ne.clone(), nt.clone(), ops.KGE_DISTMULT, reg=1e-2)[0]
        loss.backward()
        grads.append((float(loss.detach()), z.grad.clone(), rel.grad.clone()))
    (l0, gz0, gr0), (l1, gz1, gr1) = grads
    ez = float((gz0 - gz1).norm() / gz0.norm().clamp_min(1e-30))
    er = float((gr0 - gr1).norm() / gr0.norm().clamp_min(1e-30))
    ok = l0 == l1 and ez < 2e-6 and er == 0.0 and bool(torch.isfinite(gz1).all())
    if not ok:
        bad += 1
        print("MISMATCH", dict(prec=prec, n=n, d=d, r=r, t=t, hubs=hubs, ez=ez, er=er, l0=l0, l1=l1))
ops.KGE_BWD_STREAMS = True
print(f"{cases} cases, {bad} bad, {taken[0]} stream-form launches")
assert taken[0] == 2 * cases, "the stream form was not the path taken"
sys.exit(1 if bad else 0)
