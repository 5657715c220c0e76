#!/bin/bash
# L2-miss traffic of the K = 768 products (FETCH_SIZE x 2, WRITE_SIZE; separate --pmc passes): does A leave the fabric once
# per call (the column groups of a row tile sharing it through their XCD's L2) or once per column group?
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/xwf_$n
  rocprofv3 --pmc $c --output-format csv -d /tmp/xwf_$n -o p -- python3 $ROOT/scripts/dev_xw_slab.py > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
def load(n, col):
    out = collections.defaultdict(list)
    for path in glob.glob(f"/tmp/xwf_{n}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == col and ("gemm_xw" in r["Kernel_Name"]):
                out[(r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return out
f, w = load("FETCH_SIZE", "FETCH_SIZE"), load("WRITE_SIZE", "WRITE_SIZE")
h, m = load("TCC_HIT_sum", "TCC_HIT_sum"), load("TCC_HIT_sum", "TCC_MISS_sum")
for k in sorted(f, key=lambda k: -sum(f[k])):
    fe = 2 * 1024 * sum(f[k]) / len(f[k]); wr = 1024 * sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1)
    hh, mm = sum(h.get(k, [0])), sum(m.get(k, [0]))
    print(f"{k[0]:62s} grid {k[1]:>8} n={len(f[k]):3d} fetch {fe/1e6:8.1f} MB  write {wr/1e6:8.1f} MB  L2 hit {hh/max(hh+mm,1):.2f}")
PY
