#!/usr/bin/env python3
"""Wire / compute model of the node-partitioned step on the bench graph (no GPU needed): for P = 2, 4, 8 ranks, the rows
every rank receives per exchange from the patterns of biomedkg_amd/partition.py, the bytes of one bf16 step (round 4
design: gathers only, exact every-row exchanges for unequal shards, both-owner scoring -- no dz on the wire --, the first
R-GCN layer's constant input gathered once per graph), the time they take over xGMI against the rank's share of the
single-GPU kernel time, and an OVERLAP SCHEDULE: which exchange can hide under which kernels.

    python scripts/partition_model.py [single_gpu_ms_ggd=4.2] [single_gpu_ms_kge=6.3] [--json out.json]
                                      [--measured profiles/r05_simulated_ranks.json]

Compute per rank: with ``--measured``, the MEASURED ms per step of the slowest simulated rank (scripts/simulate_ranks.py on
one MI355X: the rank's real structures and graphed sub-steps, every exchange a local copy of the same byte count) -- round
4's linear term (single-GPU sub-steps x the rank's load share) had no per-kernel floor and no replicated work and was
2.7x too low at 8 ranks.  Without it: that linear term, plus what both-owner scoring adds (a
cross-rank triplet is scored by both owners: scorer forward x (2 - 1/P) of its 1/P share).  Wire time = bytes the busiest
rank RECEIVES / (links x 153 GB/s); the full mesh has 7 links per GPU and every exchange here is a set of direct
point-to-point transfers (all_to_all_single with per-peer counts, or the all-gather of equal shards), so the seven-link
figure is what a transfer that spreads over all peers approaches; the one-link figure is the other end.
"""
import json
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import partition as P, synthetic  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--") and not a.endswith(".json")]
GGD_MS, KGE_MS = (float(a) for a in (args[:2] + ["4.2", "6.3"][len(args[:2]):]))
JSON_OUT = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
MEASURED = None
if "--measured" in sys.argv:
    _m = json.load(open(sys.argv[sys.argv.index("--measured") + 1]))
    MEASURED = {w["world"]: w for w in _m["worlds"] if "compute_ms" in w and w["compute_ms"]}
LINK_GBS, LINKS = 153.0, 7          # xGMI: 7 point-to-point links per GPU, ~153 GB/s each (MI355X_MICROARCH.md)
N, E, R, H = 130_000, 4_000_000, 30, 256
# measured single-GPU pieces of the sub-steps (docs/experiments.md 9-7, scripts/attic/dev_time_calls.py, v28): what an exchange can hide under
SCORER_FWD_MS = 1.14                 # kge_fused_bce over positives + negatives
RGCN_PRODUCTS_MS = 1.35              # grouped / root / weight-gradient products of the 4 R-GCN layers
GCN_PRODUCTS_MS = 0.8                # x W products + weight gradients of the 4 GCN layers (two views)
NEG_STRUCT_MS = 0.62                 # the fresh negatives' sampler + incidence structure

# the one message that does not shrink with P: the flat gradient buckets (fp32), all-reduced every step -- the R-GCN + DistMult
# parameters (4 x (30 + 1) x 256 x 256 + biases + 30 x 256) and the GGD stack's (768 x 256 + 3 x 256 x 256 + 2 MLP layers)
KGE_GRAD_MB = (4 * (R + 1) * H * H + 4 * H + R * H) * 4 / 1e6
GGD_GRAD_MB = (768 * H + 3 * H * H + 2 * H * H + 6 * H) * 4 / 1e6

g = synthetic.primekg_like(N, E, R, seed=42)
ei = g["edge_index"]
rows_out = []
print("| P | rows / rank (min..max) | padded / real rows of a whole-table exchange | share of remote rows read in / out (max) | "
      "wire MB / step / rank (round 3; round 2) | wire ms: one link .. seven links | compute ms / rank | exposed wire ms (as "
      "coded; seven links) | step bound: as coded .. two sub-steps interleaved | speed-up over one GPU |")
print("|---|---|---|---|---|---|---|---|---|---|")
for world in (2, 4, 8):
    recv = {"in": [], "out": [], "all": []}
    rows, load, waste = [], [], 1.0
    ld = P.NodePartition.load_of(ei, N, P.NodePartition.NODE_WEIGHT)
    for rank in range(world):
        part = P.NodePartition(ei, N, rank, world)
        rows.append(part.n_local)
        waste = part.padding_waste
        load.append(float(ld[part.lo:part.hi].sum() / ld.sum()))
        for name in ("in", "out", "all"):
            pat = part.comm.patterns.get(name)
            recv[name].append(pat.n_recv if pat is not None else part.n_max * (world - 1))
    n_max = max(rows)
    padded = n_max * (world - 1)
    b = 2                                       # bf16 on the wire
    worst, worst_old3, parts_of_worst = 0.0, 0.0, None
    for rank in range(world):
        i, o, a = recv["in"][rank], recv["out"][rank], recv["all"][rank]
        ggd = {"layer-0 product, every row (h columns, both views share it)": a * H * b,
               "forward tables of layers 1-3 (two views)": 3 * i * 2 * H * b,
               "gathered dOut of 4 layers (two views)": 4 * o * 2 * H * b,
               "layer-0 backward: the corrupted halves' table gradient, every row": a * H * b}
        kge = {"R-GCN input tables of layers 1-3 (layer 0: constant, gathered once per graph)": 3 * i * H * b,
               "gathered dOut of layers 1-3": 3 * o * H * b,
               "scorer: the embedding table, every row (its gradient: nothing, both-owner scoring)": a * H * b}
        total = (sum(ggd.values()) + sum(kge.values())) / 1e6
        if total > worst:
            worst, parts_of_worst = total, (ggd, kge)
        # round 3 design on the same patterns: padded whole-table all-gathers, 4 R-GCN input tables, fp32 dz reduce-scatter
        i3 = i if part.comm.patterns.get("in") is not part.comm.patterns.get("all") else padded
        o3 = o if part.comm.patterns.get("out") is not part.comm.patterns.get("all") else padded
        old3 = (padded * H * b + 4 * o3 * 2 * H * b + padded * H * b + 3 * i3 * 2 * H * b
                + 4 * i3 * H * b + 3 * o3 * H * b + padded * H * b + padded * H * 4) / 1e6
        worst_old3 = max(worst_old3, old3)
    # round 2: all-gather of every forward table + fp32 reduce-scatter of every table gradient
    old2 = (padded * H * 2 + 3 * padded * 2 * H * 2 + padded * H * 4 + 3 * padded * 2 * H * 4
            + 4 * padded * H * 2 + 3 * padded * H * 4 + padded * H * 2 + padded * H * 4) / 1e6
    t1, t7 = worst / LINK_GBS, worst / (LINK_GBS * LINKS)
    share = max(load)
    extra = SCORER_FWD_MS * share * (1.0 - 1.0 / world)       # cross-rank triplets scored twice
    comp_linear = (GGD_MS + KGE_MS) * share + extra
    comp = comp_linear
    measured = MEASURED.get(world) if MEASURED else None
    if measured:
        # the slowest simulated rank, sub-steps replayed as hipGraphs; its local stand-in copies of the exchanged rows are in
        # it (a lower bound of what the rank's stream spends landing them)
        comp = measured["compute_ms"]
    # ---- overlap schedule as coded (per sub-step the chain layer -> exchange -> layer): what runs under each exchange
    ggd, kge = parts_of_worst
    ms = lambda by: by / 1e6 / (LINK_GBS * LINKS)             # noqa: E731  (seven links)
    hide = {
        # GCN forward: the product is cut into 4 row chunks, 3/4 of the exchange overlaps the remaining chunks' products
        "forward tables of layers 1-3 (two views)": 0.75 * GCN_PRODUCTS_MS * 0.3 * share,
        # R-GCN forward: the root product runs under the input gather
        "R-GCN input tables of layers 1-3 (layer 0: constant, gathered once per graph)": RGCN_PRODUCTS_MS * 0.15 * share,
        # R-GCN backward: weight / root / bias gradients (all local) run under the gather of dOut
        "gathered dOut of layers 1-3": RGCN_PRODUCTS_MS * 0.45 * share,
        # the scorer's table: the negatives' draw + their incidence structure do not depend on z
        "scorer: the embedding table, every row (its gradient: nothing, both-owner scoring)": NEG_STRUCT_MS * share * (2 - 1 / world),
    }
    exposed = sum(max(ms(by) - hide.get(name, 0.0), 0.0) for grp in (ggd, kge) for name, by in grp.items())
    # gradient all-reduce (ring: 2 (P - 1) / P of the bucket per rank; one ring = one link, RCCL's multi-ring forms approach
    # seven): the GGD bucket is reduced inside its sub-step's graph; the KGE bucket on a stream and communicator of its own
    # UNDER the next GGD sub-step (bench.py make(defer_update=True)), exposed only where it outlasts that sub-step
    ring = 2.0 * (world - 1) / world
    ar_kge_1, ar_kge_7 = ring * KGE_GRAD_MB / LINK_GBS, ring * KGE_GRAD_MB / (LINK_GBS * LINKS)
    ar_ggd_7 = ring * GGD_GRAD_MB / (LINK_GBS * LINKS)
    ggd_compute = (measured["compute_ms_ggd"] if measured else GGD_MS * share)
    ar_exposed_7 = ar_ggd_7 + max(ar_kge_7 - ggd_compute, 0.0)
    ar_exposed_1 = ring * GGD_GRAD_MB / LINK_GBS + max(ar_kge_1 - ggd_compute, 0.0)
    ar_inline_7 = ar_ggd_7 + ar_kge_7                              # (both buckets inside their sub-steps' graphs: round 4 before)
    exposed += ar_exposed_7
    as_coded = comp + exposed
    interleaved = max(comp, t7) + 0.05                        # the two sub-steps on two streams: each hides the other's waits
    single = GGD_MS + KGE_MS
    fr_in = max(r_ / (N - n_) for r_, n_ in zip(recv["in"], rows))
    fr_out = max(r_ / (N - n_) for r_, n_ in zip(recv["out"], rows))
    print(f"| {world} | {min(rows)}..{max(rows)} | {waste:.2f} | {fr_in:.2f} / {fr_out:.2f} | {worst:.0f} ({worst_old3:.0f}; {old2:.0f}) | "
          f"{t1:.2f} .. {t7:.2f} | {comp:.2f} | {exposed:.2f} | {as_coded:.2f} .. {interleaved:.2f} | "
          f"{single / as_coded:.1f}x .. {single / interleaved:.1f}x |   grad all-reduce {KGE_GRAD_MB:.1f} + {GGD_GRAD_MB:.1f} MB: KGE bucket "
          f"{ar_kge_1:.2f} .. {ar_kge_7:.3f} ms (one ring .. seven) under a GGD sub-step of {ggd_compute:.2f} ms: exposed "
          f"{ar_exposed_1:.2f} .. {ar_exposed_7:.3f} ms (inline: {ar_inline_7:.3f} at seven)")
    rows_out.append({"world": world, "rows_min": min(rows), "rows_max": max(rows), "padded_over_real": waste,
                     "compute_ms_linear_model_r04": comp_linear,
                     "compute_ms_measured": (measured["compute_ms"] if measured else None),
                     "compute_ms_measured_per_rank": ([round(r["ms_per_step"], 3) for r in measured["ranks"]] if measured else None),
                     "compute_only_speedup_bound": (single / comp),
                     "halo_in": fr_in, "halo_out": fr_out, "wire_MB": worst, "wire_MB_round3": worst_old3,
                     "wire_MB_round2": old2, "wire_ms_one_link": t1, "wire_ms_seven_links": t7, "compute_ms": comp,
                     "exposed_ms_as_coded": exposed, "step_ms_as_coded": as_coded, "step_ms_interleaved": interleaved,
                     "speedup_as_coded": single / as_coded, "speedup_interleaved": single / interleaved,
                     "grad_allreduce": {"kge_MB": KGE_GRAD_MB, "ggd_MB": GGD_GRAD_MB, "kge_ms_one_ring": ar_kge_1,
                                        "kge_ms_seven_rings": ar_kge_7, "exposed_ms_seven_rings": ar_exposed_7,
                                        "exposed_ms_one_ring": ar_exposed_1, "inline_ms_seven_rings": ar_inline_7,
                                        "ggd_substep_compute_ms": ggd_compute},
                     "bytes": {"ggd": {k: v / 1e6 for k, v in ggd.items()}, "kge": {k: v / 1e6 for k, v in kge.items()}}})
if JSON_OUT:
    json.dump({"single_gpu_ms": {"ggd": GGD_MS, "kge": KGE_MS}, "link_GBs": LINK_GBS, "links": LINKS,
               "compute_term": ("measured: slowest simulated rank per world size (scripts/simulate_ranks.py, one MI355X, graphed "
                                "sub-steps, exchanges = local copies of the same byte count)" if MEASURED else
                                "linear: single-GPU sub-steps x the rank's load share"),
               "measured_source": (sys.argv[sys.argv.index("--measured") + 1] if MEASURED else None),
               "what_bounds_it": ("compute does not shrink with 1/P: ~350 launches per rank and step whose kernels fall to 5-20 us "
                                  "each (per-kernel floors: launch ramp, tails, split-K reductions), work every rank repeats (the "
                                  "dropout_edge view's weights over all E edges, the paired first-layer table over all N rows, "
                                  "both-owner scoring x (2 - 1/P)), pack / unpack passes of the halo exchanges; the wire is the "
                                  "smaller term" if MEASURED else None),
               "ranks": rows_out},
              open(JSON_OUT, "w"), indent=1)
