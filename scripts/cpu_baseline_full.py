#!/usr/bin/env python3
"""The CPU oracle's full-size training step (bench.py ``cpu_full_step``: oracle/lean.py at N = 130,000 / E = 4,000,000 /
R = 30) with ALL cores and with ONE thread on the same graph -- VERDICT round 3 #9.  Minutes per step on one thread, so it
runs once per round through gpurun and its result is committed:

    python3 scripts/cpu_baseline_full.py gpurun_out/r04_cpu_baseline_full.json     (copy to profiles/)
"""
import json
import os
import sys
import time

sys.path.insert(0, ".")
import bench  # noqa: E402


def main(out_path):
    cores = min(len(os.sched_getaffinity(0)), 64)
    n, e, r, h = 130_000, 4_000_000, 30, 256
    res = {"nodes": n, "edges": e, "relations": r, "hidden": h, "neg_ratio": 1, "cores": cores, "cpu_model": bench.cpu_model()}
    t0 = time.time()
    sec_all, ggd_all = bench.cpu_full_step(n, e, r, h, 1, cores, steps=2)
    res.update(seconds_all_cores=sec_all, seconds_all_cores_ggd=ggd_all, edges_per_s_all_cores=e / sec_all)
    print(json.dumps(res), flush=True)
    sec_1, ggd_1 = bench.cpu_full_step(n, e, r, h, 1, 1, steps=1)
    res.update(seconds_1_thread=sec_1, seconds_1_thread_ggd=ggd_1, edges_per_s_1_thread=e / sec_1,
               thread_scaling=sec_1 / sec_all, wall_seconds=time.time() - t0,
               note="one GGD + one R-GCN/DistMult training step (fwd + bwd + clip + Adam) of the CPU oracle, memory-lean form "
                    "(oracle/lean.py), same graph and seeds for both thread counts; all-core figure = the faster of two steps")
    print(json.dumps(res), flush=True)
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cpu_baseline_full.json")
