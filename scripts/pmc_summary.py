#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md
"rocprofv3 PMC slots" requires) into per-kernel HBM-side traffic per launch.

    python scripts/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/*/*_counter_collection.csv \
                                  gpurun_out/pmc_WRITE_SIZE/*/*_counter_collection.csv  profiles/rNN_pmc_traffic.json [fp32|bf16] \
                                  [bench.json]

``bench.json`` (a bench.py output line of the same build): its roofline entries carry (tag, grid_size); the matching
spmm_items launches get that ``tag`` so that bench.py can look its traffic up.

Units/corrections (MI355X_MICROARCH.md "HBM"): the counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of wide (16 B/lane) coalesced reads, which is what every gather in these
kernels is, so fetched bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16 B/lane stores.
Launches of one kernel are grouped by Grid_Size (= one call site / shape).
"""
import collections
import csv
import json
import sys


def load(path, counter):
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "bkg" in r["Kernel_Name"]:   # demangled "bkg::..." or "_ZN3bkg..."
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            groups[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return groups


def main(fetch_csv, write_csv, out_json, precision="fp32", bench_json=None):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    tags = {}
    if bench_json:
        b = json.loads([ln for ln in open(bench_json) if ln.startswith("{")][-1])
        for r in [b.get("roofline")] + list(b.get("kernels") or []):
            if r and r.get("grid_size"):
                tags[(int(r["grid_size"]), int(r["pieces"]))] = r["tag"]
    out = []
    for key in sorted(f, key=lambda k: -sum(f[k])):
        fv, wv = f[key], w.get(key, [0.0])
        fetch = 2.0 * 1024 * sum(fv) / len(fv)
        write = 1024 * sum(wv) / len(wv)
        tag = None
        if "spmm_items" in key[0]:      # "spmm_items_kernel<__bf16, 64, 1>" or its mangled form "...IDF16bLi64ELi1EE..."
            for (grid, pieces), t in tags.items():
                if grid == key[1] and (f", {pieces}, " in key[0] or f"Li{pieces}ELi" in key[0]):
                    tag = t
        out.append({"kernel": key[0], "grid_size": key[1], "precision": precision, "launches": len(fv), "tag": tag,
                    "fetch_bytes_per_launch": fetch,
                    "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write})
    prev = []
    try:
        prev = [k for k in json.load(open(out_json))["kernels"] if k.get("precision") != precision]
    except (OSError, ValueError, KeyError):
        pass
    json.dump({"note": "FETCH_SIZE doubled (gfx950 wide-read correction), KiB -> bytes; separate --pmc passes; "
                       "bench.py --precision given per entry", "kernels": prev + out}, open(out_json, "w"), indent=1)
    for o in out[:8]:
        print(f"{o['kernel'][:48]:48s} grid={o['grid_size']:>10d} n={o['launches']:3d} "
              f"traffic={o['traffic_bytes_per_launch'] / 1e9:6.2f} GB")


if __name__ == "__main__":
    main(*sys.argv[1:6])
