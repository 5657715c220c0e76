#!/usr/bin/env python3
"""Median duration per (kernel, grid) of a rocprofv3 *_kernel_trace.csv: python scripts/ktrace.py <csv> [substr]"""
import csv
import statistics
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
groups = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if sub and sub not in name:
        continue
    grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    groups[(name[:110], grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (name, grid), d in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    print(f"{statistics.median(d) / 1e3:9.1f} us  x{len(d):<5} grid {grid:>9}  {name}")
