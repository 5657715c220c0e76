#!/bin/bash
# Per-kernel listing of the timed steps of bench.py (single GPU, bf16): every (kernel, grid) with launches and ms per step,
# grouped by the families of scripts/pmc_summary.py.  -> gpurun_out/step_kernels.txt
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && rm -rf /tmp/ks
rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -o k -- python3 $ROOT/bench.py --precision bf16 --steps 6 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - "$ROOT" <<'PY' | tee $ROOT/gpurun_out/step_kernels.txt
import glob, json, sys
root = sys.argv[1]
sys.path.insert(0, root + "/scripts")
import pmc_summary as P
g, step = P.load_trace(glob.glob("/tmp/ks/**/*kernel_trace.csv", recursive=True)[0], 3, 6)
print(json.dumps({k: step[k] for k in ("ms_per_step_traced", "kernel_ms_per_step", "launches_per_step", "by_family_ms_per_step")}, indent=0))
fam = {}
for (name, grid), d in g.items():
    fam.setdefault(P.family_of(name), []).append((sum(d) / 1e6 / 6, len(d) / 6, name[:100], grid))
for f, rows in sorted(fam.items(), key=lambda kv: -sum(r[0] for r in kv[1])):
    print(f"== {f}: {sum(r[0] for r in rows):.3f} ms")
    for ms, n, name, grid in sorted(rows, reverse=True)[:14]:
        print(f"   {ms:7.4f} ms x{n:5.1f} grid {grid:>9} {name}")
PY
