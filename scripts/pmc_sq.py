#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ pass per kernel: python scripts/pmc_sq.py <counter_collection.csv> [name-substring]
Prints per-kernel averages of the collected counters and the derived MFMA-busy share
(SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 4 SIMDs x CUs; SQ_BUSY_CU_CYCLES likewise over CUs)."""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, ctrs in rows.items():
    avg = {k: sum(v) / len(v) for k, v in ctrs.items()}
    print(name[:90], f"(launches {len(next(iter(ctrs.values())))})")
    for k in sorted(avg):
        print(f"   {k:32s} {avg[k]:16.0f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CU_CYCLES" in avg and avg["SQ_BUSY_CU_CYCLES"]:
        print(f"   MFMA busy / (4 SIMDs x busy CU cycles) = {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * avg['SQ_BUSY_CU_CYCLES']):.3f}")
