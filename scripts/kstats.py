#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 *_kernel_stats.csv: python scripts/kstats.py <csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    print(f"{int(r['TotalDurationNs']) / 1e6:9.2f} ms {100 * int(r['TotalDurationNs']) / tot:5.1f}% {r['Calls']:>6} "
          f"{float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:96]}")
