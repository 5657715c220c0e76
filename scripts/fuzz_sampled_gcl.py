#!/usr/bin/env python3
"""Randomised NeighborLoader / GGD regimes through biomedkg_amd/sampled_gcl.py: a replayed run against the same padded
steps run eagerly (bit-equal losses and weights), over random graph sizes, fan-outs, seed-batch sizes, widths, both
precisions and degenerate graphs (no edges, self loops only, a hub).  python scripts/fuzz_sampled_gcl.py  (MI355X)"""
import random
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops, synthetic                      # noqa: E402
from biomedkg_amd.gcl_module import GGDModule                # noqa: E402
from biomedkg_amd.sampled_gcl import NeighborGraphedGcl      # noqa: E402

dev = "cuda"
random.seed(1)
bad = 0


def graph(kind, n, e, seed):
    if kind == "random":      # (the generator draws DISTINCT pairs: ask for no more than a small graph has)
        return synthetic.gcl_small(n, max(min(e, n * (n - 1) // 4), 1), seed=seed).to(dev)
    if kind == "empty":
        return torch.zeros(2, 0, dtype=torch.int64, device=dev)
    if kind == "loops":
        i = torch.arange(min(n, max(e, 1)), device=dev)
        return torch.stack([i, i])
    g = torch.Generator().manual_seed(seed)                   # hub: every edge points at node 0
    src = torch.randint(0, n, (max(e, 1),), generator=g)
    return torch.stack([src, torch.zeros_like(src)]).to(dev)


def run(cfg, graphs):
    ops.set_matmul_precision(cfg["mode"])
    ops.set_deterministic(True)
    ei = graph(cfg["kind"], cfg["n"], cfg["e"], cfg["seed"])
    x = torch.randn(cfg["n"], 2, cfg["d"], device=dev, generator=torch.Generator(device=dev).manual_seed(cfg["seed"])) * 0.3
    torch.manual_seed(cfg["seed"])
    m = GGDModule(in_dim=cfg["d"], hidden_dim=cfg["h"], out_dim=cfg["h"], num_hidden_layers=cfg["layers"], learning_rate=1e-3,
                  fuse_method=cfg["fuse"]).to(dev).train()
    lp = NeighborGraphedGcl(m, x, ei, batch_size=cfg["bs"], num_neighbors=cfg["fan"], seed=cfg["seed"], graphs=graphs,
                            buckets=cfg["buckets"], max_graphs=cfg["cap"])
    losses = [lp.step().clone() for _ in range(cfg["steps"])]
    torch.cuda.synchronize()
    lp.close()
    return torch.stack(losses), [p.detach().clone() for p in m.model.parameters()], lp.stats


for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    n = random.choice([3, 17, 200, 1000, 3000])
    cfg = dict(kind=random.choice(["random", "random", "random", "empty", "loops", "hub"]), n=n,
               e=random.choice([1, n, 4 * n, 12 * n]), seed=it, d=random.choice([32, 64, 256]), h=random.choice([32, 64]),
               layers=random.choice([1, 2]), fuse=random.choice([None, "attention", "redaf"]),
               bs=random.choice([1, 4, 32, 128]), fan=random.choice([(3,), (4, 3), (5, 4, 3), (30, 30, 30)]),
               mode=random.choice(["fp32", "bf16"]), steps=8, cap=random.choice([2, 16]),
               buckets=random.choice([None, (64, 256, 1024, 4096, 1 << 20)]))
    if cfg["buckets"] is not None:
        pass
    try:
        la, pa, sa = run(cfg, True)
        lb, pb, sb = run(cfg, False)
        ok = bool(torch.isfinite(la).all()) and torch.equal(la, lb) and all(torch.equal(a, b) for a, b in zip(pa, pb))
    except Exception as why:      # noqa: BLE001
        ok = False
        print("raised", type(why).__name__, str(why).splitlines()[0][:200])
    print("ok " if ok else "BAD", it, {k: v for k, v in cfg.items() if k != "buckets"}, sa if ok else "", flush=True)
    bad += 0 if ok else 1
print("fuzz_sampled_gcl done, failures:", bad)
