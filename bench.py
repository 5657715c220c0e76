#!/usr/bin/env python3
"""Benchmark of the BioMedKG GCL + KGE training hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One *step* = one GCL training step (GGD, the reference's default ``model_name``; 4-layer GCN
768->256) + one KGE training step (4-layer R-GCN d=256 + DistMult, ``neg_ratio`` negatives per
positive, BCE + L2 reg) over the same synthetic PrimeKG++-shaped graph (BASELINE.json configs[2]:
~130k nodes / ~4M edges / 30 relations), each step = forward + loss + backward + grad-clip 1.0 +
Adam, inputs resident in HBM.  ``value`` = input edges / step time (every input edge counted once
although both sub-steps consume it), whole job.  N > 1: one process per GPU, each rank owns its
own graph of the same shape (weak scaling, the reference's Lightning-DDP regime) and gradients are
all-reduced over RCCL every sub-step (the KGE bucket asynchronously, under the next GGD sub-step).
Default ``--precision bf16`` = BASELINE's "bf16 storage / fp32 accumulate" configuration; ``--precision fp32``
is the exact parity mode.

Also reported on the same JSON line: ``roofline`` for the dominant HBM kernel (the R-GCN
segment-mean gather/aggregate launch, timed with HIP events on the launch stream inside the timed
region) and ``cpu_baseline`` (the CPU oracle -- a torch restatement of the reference's PyG path --
timed on this box's host cores on a bounded down-scaled sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


class Batch:
    pass


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=130_000)
    ap.add_argument("--edges", type=int, default=4_000_000)
    ap.add_argument("--relations", type=int, default=30)
    ap.add_argument("--neg-ratio", type=int, default=1)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16",
                    help="matrix-core operand precision (BASELINE configs[2] is the bf16 config; fp32 = exact mode)")
    ap.add_argument("--mode", choices=["replicas", "partition"], default="replicas",
                    help="replicas (default, what the contract measures): one graph per rank, gradient all-reduce only; "
                         "partition: ONE graph with its nodes partitioned over the ranks (all-gather per conv layer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scale", type=int, default=16, help="CPU baseline runs on a 1/scale graph")
    return ap.parse_args()


def make_step(module, batch, world, overlap=False):
    """One sub-step: zero_grad, training_step, backward, gradient all-reduce, clip, Adam.

    ``overlap`` (multi-rank only): the all-reduce is started right after backward and finished -- wait, clip,
    Adam -- at the top of this sub-step's NEXT call (or by ``step.finish()``).  The two sub-steps of the bench own
    disjoint parameters, so the KGE bucket (32.5 MB) travels over xGMI while the next GGD sub-step computes; the
    update still lands before the KGE parameters are read again, i.e. the arithmetic of the loop is unchanged."""
    from biomedkg_amd.distributed import GradBucket
    params = [p for p in module.model.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3, fused=True)   # torch's single-kernel Adam (same update; ~0.4 ms per step)
    bucket = GradBucket(params, world)   # one flat RCCL all-reduce per sub-step (<= 49 MB)
    state = {"pending": False}

    def finish():
        if state["pending"]:
            bucket.finish()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            state["pending"] = False

    def step():
        finish()
        opt.zero_grad(set_to_none=True)
        loss = module.training_step(batch)
        loss.backward()
        if overlap and world > 1:
            bucket.begin()
            state["pending"] = True
        else:
            bucket.all_reduce()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
        return loss

    step.finish = finish
    return step


def run_partitioned(args, rank, world, dev):
    """``--mode partition`` (SURVEY 8e / BASELINE configs[3]): ONE graph, nodes partitioned over the ranks, an
    all-gather of the gather table per conv layer (reduce-scatter in backward), the scorer over each rank's share of
    the triplets, weight gradients through the flat bucket.  Strong scaling: the total work is fixed.  The step is
    the same pair of sub-steps as the default mode, minus GGD's optional augmentation (the edge mask would have to be
    re-partitioned every step)."""
    from biomedkg_amd import ops, partition as P, sampling, synthetic
    from biomedkg_amd.distributed import GradBucket
    from biomedkg_amd.gcl_module import GGDModule
    from biomedkg_amd.kge_module import KGEModule
    ops.set_matmul_precision(args.precision)
    g = synthetic.primekg_like(args.nodes, args.edges, args.relations, seed=42)     # the same graph on every rank
    n, e, r, h = g["num_nodes"], g["edge_index"].size(1), g["num_relations"], args.hidden
    ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
    torch.manual_seed(42)
    xk, xg = torch.randn(n, h, device=dev), torch.randn(n, 768, device=dev)
    torch.manual_seed(1234)
    kge = KGEModule("rgcn", "dismult", h, h, h, 2, r, 2, "cosine", 1e-3, 0.2, "none", args.neg_ratio, "random").to(dev)
    gcl = GGDModule(in_dim=768, hidden_dim=h, out_dim=h, num_hidden_layers=2, fuse_method="none").to(dev)
    kge.train(); gcl.train()
    part = P.NodePartition(ei, n, rank, world)
    pg = P.PartitionedGraph(part, ei, P.HipBackend)
    prg = P.PartitionedRelGraph(part, ei, et, r, P.HipBackend)
    share = slice(rank, None, world)                          # this rank's positives (strided: every relation on every rank)
    pos_ei, pos_et = ei[:, share].contiguous(), et[share].contiguous()
    seeds = torch.Generator().manual_seed(7)

    def dropout_of(rows, cols, device):
        return ops.Dropout(0.2, keep=None, seed=int(torch.randint(0, 1 << 31, (1,), generator=seeds)), stream_id=rank)

    def make(params, loss_fn):
        opt = torch.optim.Adam(params, lr=1e-3)
        bucket = GradBucket(params, world)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = loss_fn()
            loss.backward()
            bucket.all_reduce()
            torch.nn.utils.clip_grad_norm_(params, 1.0)
            opt.step()
            return loss
        return step

    def ggd_loss():
        perm = torch.randperm(n, generator=seeds).to(dev)     # CPU generator with a common seed: identical on all ranks
        return P.ggd_loss_partitioned(gcl.model, xg, pg, P.HipBackend, perm, dropout_of)

    def kge_loss():
        k = args.neg_ratio * pos_ei.size(1)
        neg = sampling.negative_sampling(ei, n, k, seed=int(torch.randint(0, 1 << 31, (1,), generator=seeds)) * world + rank)
        nt = pos_et.repeat(args.neg_ratio)[torch.randperm(k, device=dev)]
        return P.kge_loss_partitioned(kge.model, xk, prg, P.HipBackend, pos_ei, pos_et, neg, nt, kge.model.decoder.kind,
                                      dropout_of)

    step_g = make(list(gcl.model.parameters()), ggd_loss)
    step_k = make(list(kge.model.parameters()), kge_loss)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step_g(); step_k()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lg = step_g(); lk = step_k()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    if rank == 0:
        print(json.dumps({
            "metric": "edges/sec, GCL (GGD) step + KGE (R-GCN+DistMult) step, PrimeKG++ scale", "value": e / (ms * 1e-3),
            "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"node-partitioned full graph N={n} E={e} R={r} over {world} ranks: all-gather of the "
                                   "gather table per conv layer, reduce-scatter in backward, scorer over each rank's "
                                   "share of the triplets; GGD without the optional augmentation",
                       "parallelism": f"np{world}"},
            "loss_gcl_rank0": float(lg.detach()), "loss_kge_rank0": float(lk.detach()),
            "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def pmc_traffic(kge_module, precision):
    """HBM-side bytes per launch of the roofline kernel, from the committed rocprofv3 --pmc passes
    (profiles/*pmc_traffic.json, produced by scripts/pmc_summary.py with the gfx950 FETCH_SIZE
    correction); matched by kernel name and grid size, None if this shape was not profiled."""
    try:
        import glob
        rg = kge_module  # the cached RelGraph of the bench graph (holds the work-item table)
        if rg is None or rg.seg_split is None:
            return None
        grid = ((rg.seg_split.n_items_max + 3) // 4) * 256
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
            for k in json.load(open(path))["kernels"]:
                if "spmm_items" in k["kernel"] and k["grid_size"] == grid and k.get("precision") == precision:
                    return k["traffic_bytes_per_launch"]
    except Exception:  # noqa: BLE001 - reporting only
        return None
    return None


def cpu_baseline(args):
    """CPU oracle (kind "port": the torch restatement of the reference's PyG/PyGCL path) on a
    1/cpu_scale graph with the same N:E:R ratios; one KGE + one GCL step like the GPU step."""
    from biomedkg_amd import synthetic
    from oracle import ref_modules as R
    # the cores this process may actually run on (a cgroup-limited box reports more in cpu_count())
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64)
    torch.set_num_threads(cores)
    n, e = args.nodes // args.cpu_scale, args.edges // args.cpu_scale
    g = synthetic.primekg_like(n, e, args.relations, seed=7)
    torch.manual_seed(0)
    xk, xg = torch.randn(n, args.hidden), torch.randn(n, 768)
    kge = R.make_kge_model("rgcn", "dismult", args.hidden, args.hidden, args.hidden, 2, g["num_relations"])
    ggd = R.GGD(R.GCNEncoder(768, args.hidden, args.hidden, 2), args.hidden, 1, 0.5)
    ok, og = torch.optim.Adam(kge.parameters(), lr=1e-3), torch.optim.Adam(ggd.parameters(), lr=1e-3)

    def step():
        og.zero_grad(); lg = R.ggd_loss(ggd, xg, g["edge_index"]); lg.backward()
        torch.nn.utils.clip_grad_norm_(ggd.parameters(), 1.0); og.step()
        ok.zero_grad(); lk = R.kge_step(kge, xk, g["edge_index"], g["edge_type"], args.neg_ratio)[0]; lk.backward()
        torch.nn.utils.clip_grad_norm_(kge.parameters(), 1.0); ok.step()

    t0 = time.perf_counter()
    step()  # warm-up (also bounds the cost: a slow host gets a single timed step)
    warm = time.perf_counter() - t0
    t0, reps = time.perf_counter(), 0
    while reps < 1 or (time.perf_counter() - t0 + warm < 20.0 and reps < 10):
        step(); reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": e / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle (torch restatement of the reference's PyG path), GGD + R-GCN/DistMult step on a "
                      f"1/{args.cpu_scale}-scale graph (N={n}, E={e}, R={g['num_relations']}), {reps} steps, "
                      f"{dt:.2f} s/step, {cores} threads"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1 (see module docstring)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # BKG_BENCH_ONE_DEVICE=1 / BKG_BENCH_BACKEND=gloo: functional check of the multi-rank code path on a 1-GPU box
    # (all ranks share cuda:0, gloo instead of RCCL); never used for reported numbers.
    if os.environ.get("BKG_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BKG_BENCH_BACKEND", "nccl")
        kw = {"device_id": dev} if backend == "nccl" else {}
        torch.distributed.init_process_group(backend, **kw)

    if args.mode == "partition":
        return run_partitioned(args, rank, world, dev)

    from biomedkg_amd import ops, synthetic
    from biomedkg_amd.graph import rel_graph_of
    from biomedkg_amd.gcl_module import GGDModule
    from biomedkg_amd.kge_module import KGEModule

    ops.set_matmul_precision(args.precision)
    g = synthetic.primekg_like(args.nodes, args.edges, args.relations, seed=42 + rank)
    torch.manual_seed(42 + rank)
    n, e, r, h = g["num_nodes"], g["edge_index"].size(1), g["num_relations"], args.hidden
    ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
    bk, bg = Batch(), Batch()
    bk.x, bk.edge_index, bk.edge_type = torch.randn(n, h, device=dev), ei, et
    bg.x, bg.edge_index = torch.randn(n, 768, device=dev), ei
    torch.manual_seed(1234)  # identical initial weights on every rank
    kge = KGEModule("rgcn", "dismult", h, h, h, 2, r, 2, "cosine", 1e-3, 0.2, "none", args.neg_ratio, "random").to(dev)
    gcl = GGDModule(in_dim=768, hidden_dim=h, out_dim=h, num_hidden_layers=2, fuse_method="none").to(dev)
    kge.train(); gcl.train()
    step_k, step_g = make_step(kge, bk, world, overlap=True), make_step(gcl, bg, world)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step_g(); step_k()
    step_k.finish()
    sync()
    ops.KERNEL_TIMERS = {}
    t_g = t_k = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a = time.perf_counter(); step_g()
        b = time.perf_counter(); step_k()
        t_g += b - a; t_k += time.perf_counter() - b   # host-side split only (asynchronous launches)
    step_k.finish()                                     # the last KGE update belongs to the timed region
    sync()
    dt = time.perf_counter() - t0
    timers, ops.KERNEL_TIMERS = ops.KERNEL_TIMERS, None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3

    # separately timed sub-steps (not part of the headline; one extra pass each)
    def timed(fn, reps=3):
        sync(); t = time.perf_counter()
        for _ in range(reps):
            fn()
        getattr(fn, "finish", lambda: None)()
        sync()
        return (time.perf_counter() - t) / reps

    ms_g, ms_k = timed(step_g) * 1e3, timed(step_k) * 1e3
    n_neg = args.neg_ratio * e

    roof = None
    rec = timers.get("rgcn_segment_mean")
    if rec and rec["events"]:
        durs = [a.elapsed_time(b) for a, b in rec["events"]]
        avg_ms = sum(durs) / len(durs)
        gbs = rec["bytes"] / (avg_ms * 1e-3) / 1e9
        table = "bf16 table: spmm_items_kernel<__bf16,32,1>" if args.precision == "bf16" else \
            "fp32 table: spmm_items_kernel<float,64,1>"
        roof = {"bound": "hbm", "kernel": f"{table} + fixup (R-GCN segment-mean aggregate, d={h}, fp32 accumulate, "
                                            f"{'bf16' if args.precision == 'bf16' else 'fp32'} result)",
                "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "traffic": pmc_traffic(rel_graph_of(ei, et, n, r), args.precision), "avg_launch_ms": avg_ms,
                "launches": len(durs),
                "bytes_per_launch": rec["bytes"],
                "note": "algorithmic bytes / HIP-event time; the 133 MB feature table is Infinity-Cache/L2 resident, "
                        "so the algorithmic rate can exceed the HBM peak (traffic = PMC bytes that left L2)"}
        rec2 = timers.get("gcn_propagate") or timers.get("gcn_propagate_views")
        if rec2 and rec2["events"]:
            d2 = [a.elapsed_time(b) for a, b in rec2["events"]]
            roof["gcn_propagate_GBs"] = rec2["bytes"] / (sum(d2) / len(d2) * 1e-3) / 1e9

    if rank == 0:
        out = {
            "metric": "edges/sec, GCL (GGD) step + KGE (R-GCN+DistMult) step, PrimeKG++ scale",
            "value": world * e / (ms * 1e-3), "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32",
            "dtype_note": ("bf16 storage / fp32 accumulate: matrix-core operands, gather tables, hidden activations "
                           "between conv layers (and their gradients) and intermediates that only feed products are "
                           "bf16; every accumulation, encoder outputs, scorer, losses, weights, weight gradients and "
                           "optimizer state are fp32" if args.precision == "bf16"
                           else "exact fp32 everywhere (fp32-input MFMA)"),
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] shape: synthetic PrimeKG++-like graph per GPU, "
                                   f"N={n} E={e} R={r}; GGD/GCN 768->{h} (4 conv layers) step + R-GCN d={h} "
                                   f"(4 conv layers) + DistMult, neg_ratio={args.neg_ratio}, step; "
                                   f"{args.precision} matmuls; fwd+bwd+clip+Adam", "parallelism": f"dp{world}"},
            "triplets_per_sec_kge_step": world * (e + n_neg) / (ms_k * 1e-3),
            "edges_per_sec_gcl_step": world * e / (ms_g * 1e-3),
            "ms_gcl_step": ms_g, "ms_kge_step": ms_k,
            "roofline": roof,
            "cpu_baseline": None,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
