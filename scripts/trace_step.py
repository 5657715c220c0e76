#!/usr/bin/env python3
"""Summarise ONE step of a rocprofv3 kernel trace (csv) of scripts/dev_sampled.py: steps are delimited by the random-walk
kernel; prints the launch count, the span, kernels longer than a threshold and the rest grouped by name."""
import csv, re, sys, collections
path, thr = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "random_walk_kernel" in r["Kernel_Name"]]
a, b = idx[-4], idx[-3]
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    n = re.sub(r"at::native::", "", n); n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rp::", n)
    return n[:100]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
print(f"launches {b - a}  span {span:.1f} us  kernel-sum {sum(dur(r) for r in rows[a:b]):.1f} us")
small = collections.Counter(); small_t = collections.Counter()
for i in range(a, b):
    r = rows[i]; d = dur(r)
    if d >= thr:
        print(f"{i - a:4d} {d:7.1f}  {short(r['Kernel_Name'])}  grid={r.get('Grid_Size_X', '')}")
    else:
        small[short(r["Kernel_Name"])[:70]] += 1; small_t[short(r["Kernel_Name"])[:70]] += d
print("--- below threshold:", sum(small.values()), "launches,", round(sum(small_t.values()), 1), "us")
for k, c in small.most_common(40):
    print(f"{c:4d} {small_t[k]:7.1f}  {k}")
