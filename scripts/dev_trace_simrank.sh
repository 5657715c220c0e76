#!/bin/bash
# Kernel trace of ONE simulated rank of the node-partitioned step (scripts/simulate_ranks.py, eager form): where a rank's
# time goes when the exchanges are local copies.  Usage: dev_trace_simrank.sh WORLD RANK  -> gpurun_out/simrank_W_R.json
cd /tmp && export TMPDIR=/tmp
W=${1:-8}; R=${2:-1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=/tmp/simrank_${W}_${R}
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o k -- python3 $ROOT/scripts/simulate_ranks.py --world $W --ranks $R --eager --steps 6 --warmup 3 > $OUT/run.json 2> $OUT/run.err
tail -2 $OUT/run.err
python3 - "$ROOT" "$OUT" "$W" "$R" <<'PY'
import glob, json, sys
root, out, w, r = sys.argv[1:5]
sys.path.insert(0, root + "/scripts")
import pmc_summary as P
path = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
groups, step = P.load_trace(path, 3, 6)
top = sorted(((sum(d) / 1e6 / 6, len(d) / 6, k[0][:110], k[1]) for k, d in groups.items()), reverse=True)[:45]
rec = {"world": int(w), "rank": int(r), "step": {k: step.get(k) for k in ("ms_per_step_traced", "gpu_busy_ms_per_step", "kernel_ms_per_step", "launches_per_step", "by_family_ms_per_step", "not_library_kernels", "error")},
       "top_kernels_ms_per_step": [{"ms": round(a, 4), "launches": round(b, 2), "kernel": c, "grid": d} for a, b, c, d in top]}
json.dump(rec, open(f"{root}/gpurun_out/simrank_{w}_{r}.json", "w"), indent=1)
print(json.dumps(rec["step"]["by_family_ms_per_step"], indent=0))
for a, b, c, d in top[:40]:
    print(f"{a:8.4f} ms x{b:5.1f} grid {d:>8} {c}")
PY
