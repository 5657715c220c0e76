#!/usr/bin/env python3
"""Randomised node partitions through the round-4 pieces of biomedkg_amd/partition.py, ranks simulated in one process (the
all-gather replaced by the table it delivers): the block-structured negative sampler (K distinct off-diagonal non-edges
over all ranks, both owners of a block agree) and both-owner scoring (counted losses add up to the whole-graph loss, every
rank's dz is the whole dz of its rows, d rel_emb adds up) -- over random graphs with hubs (tiny shards), isolated nodes,
empty relations, world sizes 1..9, DistMult / TransE, both precisions.  python scripts/fuzz_partition.py [cases]  (MI355X)"""
import random
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops, partition as P, sampling  # noqa: E402

dev = "cuda"
random.seed(3)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0


def slot_table(part, full):
    t = torch.zeros(part.world * part.n_max, full.size(1), dtype=full.dtype, device=full.device)
    t[part.node_slots()] = full
    return t


for it in range(cases):
    n = random.choice([40, 300, 1500, 6000, 20000])
    e = n * random.randint(1, 12)
    r = random.randint(1, 9)
    world = random.randint(1, 9)
    kind = random.choice([ops.KGE_DISTMULT, ops.KGE_TRANSE])
    mode = random.choice(["fp32", "bf16"])
    ratio = random.choice([1, 1, 2])
    d = random.choice([64, 128, 256])
    g = torch.Generator().manual_seed(it)
    hi = max(2, int(n * random.choice([1.0, 0.8])))                     # 0.8: isolated tail nodes
    ei = torch.randint(0, hi, (2, e), generator=g)
    if random.random() < 0.5:
        ei[1, : e // 3] = random.randrange(hi)                          # a hub: one rank ends up with very few rows
    ei = ei.to(dev)
    et = torch.randint(0, max(1, r - random.choice([0, 1])), (e,), generator=g).to(dev)   # maybe an empty relation
    case = dict(it=it, n=n, e=e, r=r, world=world, kind=kind, mode=mode, ratio=ratio, d=d)
    try:
        k = ratio * e
        pos_codes = set((ei[0] * n + ei[1]).tolist())
        z = (torch.randn(n, d, generator=g) * 0.3).to(dev)
        rel = (torch.randn(r, d, generator=g) * 0.3).to(dev)
        ops.set_matmul_precision(mode)
        counted, blocks = [], {}
        losses, drel, dz = [], torch.zeros_like(rel), torch.zeros_like(z)
        shares = []
        for rank in range(world):
            part = P.NodePartition(ei, n, rank, world)
            smp = sampling.BlockNegativeSampler(ei, et, part.bounds, rank, k, num_relations=r)
            idx = smp.draw(seed=1000 + it)
            assert smp.shortfall() == 0, "shortfall"
            for bid, o0, kq in smp.layout:
                tri = torch.stack([idx[0, o0:o0 + kq], idx[1, o0:o0 + kq], smp.types[o0:o0 + kq]]).cpu()
                assert bid not in blocks or torch.equal(blocks[bid], tri), "owners disagree"
                blocks[bid] = tri
            counted.append(torch.cat([idx[:, :smp.n_own], smp.types[None, :smp.n_own]]).cpu())
            shares.append((part, idx, smp))
        neg_all = torch.cat(counted, dim=1)
        codes = (neg_all[0] * n + neg_all[1]).tolist()
        assert neg_all.size(1) == k and len(set(codes)) == k and not (set(codes) & pos_codes), "not K distinct non-edges"
        assert bool((neg_all[0] != neg_all[1]).all()), "self pair"
        assert torch.equal(torch.sort(neg_all[2]).values, torch.sort(et.repeat(ratio)).values.cpu()), "id multiset"
        # whole-graph reference on the very same negatives
        nei, net = neg_all[:2].to(dev).contiguous(), neg_all[2].to(dev).contiguous()
        zw, rw = z.clone().requires_grad_(True), rel.clone().requires_grad_(True)
        whole, _, _ = ops.kge_bce_loss(zw, rw, ei, et, nei, net, kind, reg=1e-2)
        whole.backward()
        for part, idx, smp in shares:
            part.comm.all_gather = lambda shard, part=part: slot_table(part, z.to(shard.dtype))
            pos_s = P.TripletShare(part, ei, et)
            neg_s = P.TripletShare(part, idx, smp.types, n_own=smp.n_own, entries=smp.entries,
                                   types_own=smp.types_own, types_foreign=smp.types_foreign)
            zl, rl = part.shard(z).clone().requires_grad_(True), rel.clone().requires_grad_(True)
            loss = P.kge_bce_partitioned(zl, rl, pos_s, neg_s, kind, part, (e + k) / world, 1e-2, P.HipBackend)
            loss.backward()
            losses.append(float(loss))
            if zl.grad is not None and part.n_local:
                dz[part.lo:part.hi] = zl.grad / world
            drel += rl.grad / world
        tol = 2e-5 if mode == "fp32" else 5e-4
        assert abs(sum(losses) / world - float(whole)) <= tol * abs(float(whole)), ("loss", sum(losses) / world, float(whole))
        lim = (2e-5 if mode == "fp32" else 3e-3) if kind == ops.KGE_DISTMULT else 2e-2
        ez = float((dz - zw.grad).norm() / zw.grad.norm().clamp_min(1e-20))
        er = float((drel - rw.grad).norm() / rw.grad.norm().clamp_min(1e-20))
        assert ez < lim and er < lim, ("grads", ez, er)
        print("ok ", case)
    except Exception as why:      # noqa: BLE001
        bad += 1
        print("BAD", case, type(why).__name__, str(why)[:300])
    finally:
        ops.set_matmul_precision("fp32")
print("fuzz_partition done, failures:", bad)
