#!/usr/bin/env python3
"""Measured compute term of the node-partitioned step, taken on ONE GPU.

For P = 2, 4, 8 and every rank r of P, this process stands in for rank r: it builds that rank's real structures on the
bench graph (``NodePartition`` and its halo patterns, the local CSRs and (relation, node) segments, ``TripletShare``,
``BlockNegativeSampler``) through ``bench.build_partitioned_steps(simulate=True)`` -- the code the RCCL ranks run -- with
every exchange replaced by a local copy of the same byte count (``partition.Comm(simulate=True)``: pack, land and buffer
sizes as live, only the wire is missing), captures the two sub-steps as hipGraphs exactly as ``bench.py --gpus P`` does and
times the replays.  What comes out per rank: ms per step / per sub-step of GPU work, the host's time to issue a step, the
launches per step, rows and edges of the shard.  ``scripts/partition_model.py --measured`` puts the wire term next to it.

    python scripts/simulate_ranks.py --json profiles/r05_simulated_ranks.json      # P = 2, 4, 8, one child process per P
    python scripts/simulate_ranks.py --world 8 --ranks 0,7                         # in this process

Never a reported throughput: no bytes cross a wire here.
"""
import argparse
import gc
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=0, help="0: P = 2, 4, 8, each in a child process of its own")
    ap.add_argument("--ranks", default="all")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--nodes", type=int, default=130_000)
    ap.add_argument("--edges", type=int, default=4_000_000)
    ap.add_argument("--relations", type=int, default=30)
    ap.add_argument("--neg-ratio", type=int, default=1)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--eager", action="store_true", help="time the eager step instead of the graphed sub-steps")
    ap.add_argument("--json", default=None)
    return ap.parse_args()


def one_rank(args, rank, world, dev):
    import torch
    import bench
    from biomedkg_amd import _lib as L, ops
    step_g, step_k, n, e, r = bench.build_partitioned_steps(args, rank, world, dev, simulate=True)
    part = step_g.part

    def sync():
        torch.cuda.synchronize()

    for _ in range(args.warmup):            # both coin outcomes of GGD and the KGE sub-step: eager once, captured, replayed
        step_g(); step_k()
    step_k.finish(); sync()

    def timed(fns, reps):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        sync(); t0 = time.perf_counter(); marks[0].record()
        for i in range(reps):
            for fn in fns:
                fn()
            marks[i + 1].record()
        host = (time.perf_counter() - t0) / reps * 1e3
        for fn in fns:
            getattr(fn, "finish", lambda: None)()
        sync()
        wall = (time.perf_counter() - t0) / reps * 1e3
        per = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:]))
        return wall, host, per[len(per) // 2]

    wall, host, med = timed([step_g, step_k], args.steps)
    wall_g, _, med_g = timed([step_g], args.steps)
    wall_k, _, med_k = timed([step_k], args.steps)
    stats = dict(step_g.graph_stats) if step_g.graph_stats else None
    # launches of one eager step (library calls only: torch's own kernels are not counted)
    calls = [0]
    real = L.call

    def counting(name, *a):
        calls[0] += 1
        return real(name, *a)
    step_g.force_eager = step_k.force_eager = True
    L.call = ops.L.call = counting
    try:
        step_g(); step_k()
    finally:
        L.call = ops.L.call = real
    sync()
    wall_e, host_e, _ = timed([step_g, step_k], 4)
    step_g.force_eager = step_k.force_eager = False
    ei = step_g.part_edges if hasattr(step_g, "part_edges") else None
    out = {"rank": rank, "world": world, "rows": part.n_local, "n_max": part.n_max,
           "ms_per_step": wall, "ms_per_step_median_gpu": med, "host_ms_per_step": host,
           "ms_ggd": wall_g, "ms_kge": wall_k, "ms_ggd_median_gpu": med_g, "ms_kge_median_gpu": med_k,
           "ms_per_step_eager": wall_e, "host_ms_per_step_eager": host_e, "library_calls_per_eager_step": calls[0],
           "graphed_sub_steps": stats,
           "halo": {k: (None if part.comm.patterns.get(k) is None else
                        {"rows_received": part.comm.patterns[k].n_recv, "rows_sent": part.comm.patterns[k].n_send})
                    for k in ("in", "out", "all")}}
    del step_g, step_k
    gc.collect()
    torch.cuda.empty_cache()
    return out


def in_process(args):
    import torch
    from biomedkg_amd import ops
    if not torch.cuda.is_available():
        raise SystemExit("simulate_ranks.py needs an MI355X: the HIP path has no CPU fallback")
    if not args.eager:
        os.environ["BKG_BENCH_PARTITION_GRAPH"] = "1"
    # the form bench.py's first attempts ask for (in-place list-form all-to-all of whole-table exchanges; the library's
    # default is the packed all_to_all_single): BKG_DIRECT_ALL_TO_ALL=0 measures the packed form
    os.environ.setdefault("BKG_DIRECT_ALL_TO_ALL", "1")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ops.set_matmul_precision(args.precision)
    world = args.world
    ranks = list(range(world)) if args.ranks == "all" else [int(r) for r in args.ranks.split(",")]
    rows = []
    for r in ranks:
        t0 = time.time()
        rec = one_rank(args, r, world, dev)
        rec["seconds_to_measure"] = time.time() - t0
        rows.append(rec)
        sys.stderr.write(f"P={world} rank {r}: {rec['ms_per_step']:.3f} ms/step (ggd {rec['ms_ggd']:.3f}, kge {rec['ms_kge']:.3f}), "
                         f"host {rec['host_ms_per_step']:.3f}, eager {rec['ms_per_step_eager']:.3f} ms, rows {rec['rows']}\n")
    slow = max(rows, key=lambda x: x["ms_per_step"])
    return {"world": world, "ranks": rows, "slowest_rank": slow["rank"], "compute_ms": slow["ms_per_step"],
            "whole_table_exchange": ("in-place list form" if os.environ.get("BKG_DIRECT_ALL_TO_ALL") == "1"
                                     else "packed all_to_all_single"),
            "compute_ms_ggd": max(x["ms_ggd"] for x in rows), "compute_ms_kge": max(x["ms_kge"] for x in rows),
            "compute_ms_eager": max(x["ms_per_step_eager"] for x in rows),
            "mean_ms": sum(x["ms_per_step"] for x in rows) / len(rows)}


def main():
    args = parse()
    if args.world:
        out = in_process(args)
        print(json.dumps(out))
        return
    # one fresh child per P (this process never touches the GPU)
    results = []
    for world in (1, 2, 4, 8):
        cmd = [sys.executable, os.path.abspath(__file__), "--world", str(world), "--steps", str(args.steps), "--warmup",
               str(args.warmup), "--nodes", str(args.nodes), "--edges", str(args.edges), "--relations", str(args.relations),
               "--neg-ratio", str(args.neg_ratio), "--hidden", str(args.hidden), "--precision", args.precision]
        if world == 1:
            cmd += ["--ranks", "0"]
        got = subprocess.run(cmd, capture_output=True, text=True)
        sys.stderr.write(got.stderr[-3000:])
        lines = [l for l in got.stdout.splitlines() if l.startswith("{")]
        if got.returncode != 0 or not lines:
            results.append({"world": world, "error": got.stderr[-500:]})
            continue
        results.append(json.loads(lines[-1]))
    one = next((r for r in results if r.get("world") == 1 and "compute_ms" in r), None)
    for r in results:
        if one and "compute_ms" in r:
            r["speedup_bound_compute_only"] = one["compute_ms"] / r["compute_ms"]
    out = {"what": "simulated ranks on ONE MI355X: each rank's real structures and graphed sub-steps, every exchange a local copy "
                   "of the same byte count; compute_ms = the slowest rank's ms per step (no wire).  world 1 = the same code at "
                   "one rank (no exchange at all).",
           "graph": {"nodes": args.nodes, "edges": args.edges, "relations": args.relations, "neg_ratio": args.neg_ratio,
                     "hidden": args.hidden, "precision": args.precision},
           "worlds": results}
    text = json.dumps(out, indent=1)
    if args.json:
        os.makedirs(os.path.dirname(os.path.abspath(args.json)), exist_ok=True)
        open(args.json, "w").write(text)
    print(json.dumps({"worlds": [{k: r.get(k) for k in ("world", "compute_ms", "compute_ms_eager", "mean_ms",
                                                       "speedup_bound_compute_only", "error")} for r in results]}))


if __name__ == "__main__":
    main()
