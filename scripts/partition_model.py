#!/usr/bin/env python3
"""Wire / compute model of the node-partitioned step on the bench graph (no GPU needed): for P = 2, 4, 8 ranks, the rows
every rank receives per exchange from the halo patterns of biomedkg_amd/partition.py, the bytes of one bf16 step, and
the time they take over xGMI against the rank's share of the single-GPU kernel time.

    python scripts/partition_model.py [single_gpu_ms_ggd=4.5] [single_gpu_ms_kge=6.8]
"""
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import partition as P, synthetic  # noqa: E402

GGD_MS, KGE_MS = (float(a) for a in (sys.argv[1:3] + ["4.5", "6.8"])[:2])
LINK_GBS, LINKS = 153.0, 7          # xGMI: 7 point-to-point links per GPU, ~153 GB/s each (MI355X_MICROARCH.md)
N, E, R, H = 130_000, 4_000_000, 30, 256
g = synthetic.primekg_like(N, E, R, seed=42)
ei = g["edge_index"]
print(f"| P | rows/rank (min..max) | share of remote rows read in / out (max) | wire MB/step/rank (max; round 2 design) | "
      f"wire ms: one link .. seven links | compute ms/rank (max share) |")
print("|---|---|---|---|---|---|")
for world in (2, 4, 8):
    recv = {"in": [], "out": []}
    rows, load = [], []
    ld = P.NodePartition.load_of(ei, N, P.NodePartition.NODE_WEIGHT)
    for rank in range(world):
        part = P.NodePartition(ei, N, rank, world)
        rows.append(part.n_local)
        load.append(float(ld[part.lo:part.hi].sum() / ld.sum()))
        for name in ("in", "out"):
            pat = part.comm.patterns[name]
            recv[name].append(pat.n_recv if pat is not None else part.n_max * (world - 1))
    n_max = max(rows)
    full = n_max * (world - 1)                 # rows of a whole-table all-gather
    worst = 0.0
    for rank in range(world):
        i, o = recv["in"][rank], recv["out"][rank]
        b = 2                                   # bf16 on the wire
        ggd = full * H * b                      # layer 0: the shared first product, all-gathered whole (h columns)
        ggd += 4 * o * 2 * H * b                # 4 layers: gathered dOut, two views side by side
        ggd += full * H * b                     # layer 0 backward: the corrupted halves' table gradient (h columns)
        ggd += 3 * i * 2 * H * b                # layers 1-3 forward tables, two views
        kge = 4 * i * H * b + 3 * o * H * b     # R-GCN: 4 input tables, 3 gathered dOut (the first layer needs no dX)
        kge += full * H * b + full * H * 4      # scorer: z table (bf16) + dz reduce-scatter (fp32)
        worst = max(worst, (ggd + kge) / 1e6)
    # round 2: all-gather of every forward table + fp32 reduce-scatter of every table gradient
    old = (full * H * 2 + 3 * full * 2 * H * 2 + full * H * 4 + 3 * full * 2 * H * 4      # GGD
           + 4 * full * H * 2 + 3 * full * H * 4 + full * H * 2 + full * H * 4) / 1e6    # KGE
    t1, t7 = worst / LINK_GBS, worst / (LINK_GBS * LINKS)
    comp = (GGD_MS + KGE_MS) * max(load)
    fr_in = max(r_ / (N - n_) for r_, n_ in zip(recv["in"], rows))
    fr_out = max(r_ / (N - n_) for r_, n_ in zip(recv["out"], rows))
    print(f"| {world} | {min(rows)}..{max(rows)} | {fr_in:.2f} / {fr_out:.2f} | {worst:.0f} ({old:.0f}) | {t1:.2f} .. {t7:.2f} | "
          f"{comp:.2f} ({max(load):.3f}) |")
