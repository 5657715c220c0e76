#!/bin/bash
# Kernel timeline of ONE first call of AttentionFusion (bf16 mode, 130k nodes): which launches make up its 1.2-1.3 ms.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && rm -rf /tmp/af
rocprofv3 --kernel-trace --output-format csv -d /tmp/af -o k -- python3 $ROOT/scripts/dev_attn_first_call.py > /tmp/af.log 2>&1
python3 - <<'PY'
import csv, glob
path = glob.glob("/tmp/af/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))
last = max(i for i, r in enumerate(rows) if "attn_scores_combine" in r[2])
lo = max(0, last - 8)
t0 = rows[lo][0]
for a, b, n in rows[lo:last + 6]:
    print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:8.1f} us  {n[:100]}")
PY
