#!/bin/bash
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && rm -rf /tmp/k10
rocprofv3 --kernel-trace --output-format csv -d /tmp/k10 -o k -- python3 $ROOT/scripts/dev_kge_neg10.py > /tmp/k10.log 2>&1
python3 - "$ROOT" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
sys.path.insert(0, root + "/scripts")
import pmc_summary as P
path = glob.glob("/tmp/k10/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
ends = [b for a, b, n in rows if "clip_adam_kernel" in n]
t0, t1 = ends[2], ends[7]                       # 5 steps after 3 warm-up steps
win = [r for r in rows if r[0] >= t0 and r[1] <= t1]
fam, top = collections.defaultdict(float), collections.defaultdict(lambda: [0, 0.0])
for a, b, n in win:
    fam[P.family_of(n)] += (b - a) / 1e6 / 5
    k = n.split("(")[0][-80:]
    top[k][0] += 1; top[k][1] += (b - a) / 1e6 / 5
print("ms per step", (t1 - t0) / 1e6 / 5, "kernels", sum(fam.values()))
for f, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print(f"  {v:7.3f} ms  {f}")
for k, (c, v) in sorted(top.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"  {v:7.3f} ms x{c / 5:5.1f}  {k}")
PY
