#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE and optionally TCC_HIT_sum + TCC_MISS_sum; separate runs as
MI355X_MICROARCH.md "rocprofv3 PMC slots" requires) into per-kernel L2-miss traffic per launch.

    python scripts/pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv profiles/rNN_pmc_traffic.json \
                                  [fp32|bf16] [bench.json] [TCC_counter_collection.csv] [kernel_trace.csv warmup steps]

``kernel_trace.csv`` (a ``rocprofv3 --kernel-trace`` run of ``bench.py --steps <steps> --warmup <warmup> --no-extras``): every
(kernel, grid) group also gets ``avg_duration_us`` -- the mean launch duration over the TIMED steps of that run, i.e. between
the end of the 2*warmup-th and of the 2*(warmup+steps)-th ``clip_adam`` launch (each of a step's two sub-steps ends with
one) -- so that every ``kernels[].frac`` of the bench line can be recomputed from profiles/ alone (bytes / duration), and a
``step`` object: traced ms per step, GPU-busy ms per step, the tagged calls' ms per step and the rest BY KERNEL FAMILY.

``bench.json`` (a bench.py output line of the same build): its roofline entries carry (tag, grid_size,
grid_size_second_launch, pieces); the matching spmm_items / spmm_fixup launches get that ``tag`` so that bench.py can look
the call's traffic up (it sums the launches of a call).  The two KGE kernels of the step are tagged by name.

Units/corrections (MI355X_MICROARCH.md "HBM"): the counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of wide (16 B/lane) coalesced reads, which is what every gather in these
kernels is, so fetched bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact for 16 B/lane stores.  Both derive from the
L2's fabric-side request counters: Infinity-Cache hits are INCLUDED (this is L2-miss traffic, not HBM traffic).
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


FAMILIES = (("spmm", "gather / aggregate (spmm_items, spmm_fixup)"), ("kge_", "KGE scorer + node-side backward"),
            ("gemm_", "products (gemm_*)"), ("weight_to_bf16", "products (gemm_*)"), ("rocprim", "rocPRIM sorts / scans"),
            ("clip_adam", "optimizer (clip + Adam)"), ("grad_sqnorm", "optimizer (clip + Adam)"),
            ("cast", "casts / gates / column sums / row dots"), ("colsum", "casts / gates / column sums / row dots"),
            ("gate", "casts / gates / column sums / row dots"), ("rowdot", "casts / gates / column sums / row dots"),
            ("scaled_rows", "casts / gates / column sums / row dots"), ("pair", "casts / gates / column sums / row dots"),
            ("negative_", "negative sampler + structures"), ("hashset", "negative sampler + structures"),
            ("rowsplit", "negative sampler + structures"), ("sort", "negative sampler + structures"),
            ("role_incidence", "negative sampler + structures"),
            ("ggd_bce", "losses"), ("bkg", "other library kernels"), ("at::", "ATen (fills, norms, glue)"),
            ("6native", "ATen (fills, norms, glue)"), ("fillBuffer", "ATen (fills, norms, glue)"))


def family_of(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return "ATen (fills, norms, glue)"


def load_trace(path, warmup, steps):
    """-> ({(kernel, grid): [durations ns]}, step object) over the timed steps of a traced bench run."""
    rows = []
    for r in csv.DictReader(open(path)):
        grid = int(r.get("Grid_Size", 0) or 0)
        if not grid:
            grid = int(r.get("Grid_Size_X", 1) or 1) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], grid))
    rows.sort()
    ends = [b for a, b, name, _ in rows if "clip_adam_kernel" in name or "clip_adam_images_kernel" in name]
    lo_i, hi_i = 2 * warmup, 2 * (warmup + steps)
    if len(ends) < hi_i or hi_i <= lo_i:
        return {}, {"error": f"{len(ends)} clip_adam launches in the trace, need {hi_i}"}
    t0, t1 = (ends[lo_i - 1] if lo_i else rows[0][0]), ends[hi_i - 1]
    window = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    groups = collections.defaultdict(list)
    fam = collections.defaultdict(float)
    foreign = collections.defaultdict(lambda: [0, 0.0])      # kernels that are not the library's: launches, ms per step
    busy, cur = 0, t0
    for a, b, name, grid in window:
        short = name.split("(")[0].replace("void ", "")
        groups[(short, grid)].append(b - a)
        fam[family_of(name)] += (b - a) / 1e6 / steps
        if "bkg" not in name:
            rec = foreign[short[:96]]
            rec[0] += 1
            rec[1] += (b - a) / 1e6 / steps
        a2 = max(a, cur)
        if b > a2:
            busy += b - a2
            cur = b
    step = {"steps": steps, "warmup": warmup, "ms_per_step_traced": (t1 - t0) / 1e6 / steps,
            "gpu_busy_ms_per_step": busy / 1e6 / steps, "kernel_ms_per_step": sum(fam.values()),
            "launches_per_step": len(window) / steps,
            "by_family_ms_per_step": dict(sorted(fam.items(), key=lambda kv: -kv[1])),
            "library_share_of_kernel_time": 1.0 - sum(v[1] for v in foreign.values()) / max(sum(fam.values()), 1e-30),
            "not_library_kernels": [{"kernel": k, "launches_per_step": v[0] / steps, "ms_per_step": v[1]}
                                    for k, v in sorted(foreign.items(), key=lambda kv: -kv[1][1])[:25]],
            "note": "window = the timed steps of a rocprofv3 --kernel-trace run of bench.py (between clip_adam launches); "
                    "kernel_ms_per_step sums durations (kernels on two queues may overlap: gpu_busy is their union)"}
    return groups, step


def main(fetch_csv, write_csv, out_json, precision="fp32", bench_json=None, tcc_csv=None, trace_csv=None, warmup=2, steps=5):
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    durations, step = load_trace(trace_csv, int(warmup), int(steps)) if trace_csv else ({}, None)
    hit = load(tcc_csv, "TCC_HIT_sum") if tcc_csv else {}
    miss = load(tcc_csv, "TCC_MISS_sum") if tcc_csv else {}
    tags = {}
    if bench_json:
        try:                                  # the full record (bench_detail.json, round 5) ...
            b = json.load(open(bench_json))
        except ValueError:                    # ... or a file holding bench.py's stdout (the line comes last)
            b = json.loads([ln for ln in open(bench_json) if ln.startswith("{")][-1])
        for r in [b.get("roofline")] + list(b.get("kernels") or []):
            if r and r.get("grid_size"):
                tags.setdefault(("spmm_items", int(r["grid_size"]), int(r["pieces"])), []).append(r["tag"])
                if r.get("grid_size_second_launch"):
                    tags.setdefault(("spmm_fixup", int(r["grid_size_second_launch"]), int(r["pieces"])), []).append(r["tag"])
    out = []
    for key in sorted(f, key=lambda k: -sum(f[k])):
        fv, wv = f[key], w.get(key, [0.0])
        fetch = 2.0 * 1024 * sum(fv) / len(fv)
        write = 1024 * sum(wv) / len(wv)
        tag, also = None, []
        for kind in ("spmm_items", "spmm_fixup"):      # "spmm_items_kernel<__bf16, 64, 1>" or mangled "...IDF16bLi64ELi1E..."
            if kind in key[0]:
                for (k2, grid, pieces), ts in tags.items():
                    lpr = min(pieces, 64)
                    if k2 == kind and grid == key[1] and (f", {lpr}, " in key[0] or f"Li{lpr}ELi" in key[0]):
                        tag, also = ts[0], ts      # calls that launch the same kernel over the same grid share the group
        if any(k in key[0] for k in ("kge_bwd_nodes_kernel", "kge_bwd_nodes_fixup", "kge_bwd_stream_kernel",
                                     "kge_incidence_streams_kernel")):
            tag = "kge_bwd_nodes"
        if "kge_fused_bce_kernel" in key[0]:
            tag = "kge_fused_bce"
        if "kge_onepass_kernel" in key[0]:
            tag = "kge_onepass"
        if "kge_onepass_mx_kernel" in key[0]:
            tag = "kge_onepass_mx"
        rec = {"kernel": key[0], "grid_size": key[1], "precision": precision, "launches": len(fv), "tag": tag, "tags": also,
               "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write}
        if key in durations:
            d = durations[key]
            rec["avg_duration_us"] = sum(d) / len(d) / 1e3
            rec["launches_traced"] = len(d)
        if key in hit or key in miss:
            rec["tcc_hit"] = sum(hit.get(key, [0.0])) / max(len(hit.get(key, [0.0])), 1)
            rec["tcc_miss"] = sum(miss.get(key, [0.0])) / max(len(miss.get(key, [0.0])), 1)
            rec["l2_hit_rate"] = rec["tcc_hit"] / max(rec["tcc_hit"] + rec["tcc_miss"], 1.0)
        out.append(rec)
    prev = []
    try:
        prev = [k for k in json.load(open(out_json))["kernels"] if k.get("precision") != precision]
    except (OSError, ValueError, KeyError):
        pass
    if step is not None and "error" not in step:
        tagged = collections.defaultdict(float)
        for o in out:
            if o.get("tag") and "avg_duration_us" in o:
                tagged[o["tag"]] += o["avg_duration_us"] * o["launches_traced"] / 1e3 / step["steps"]
        step["tagged_ms_per_step"] = dict(tagged)
        step["tagged_ms_per_step_total"] = sum(tagged.values())
        step["untagged_ms_per_step"] = step["kernel_ms_per_step"] - sum(tagged.values())
    doc = {"note": "FETCH_SIZE doubled (gfx950 wide-read correction), KiB -> bytes; separate --pmc passes; L2 fabric-side "
                   "requests: Infinity-Cache hits included (guide l. 297); tcc_* = TCC_HIT_sum / TCC_MISS_sum per launch; "
                   "avg_duration_us: mean launch duration of the group over the timed steps of a --kernel-trace run of the "
                   "same command; bench.py --precision given per entry", "kernels": prev + out}
    if step is not None:
        doc["step"] = {precision: step}
    json.dump(doc, open(out_json, "w"), indent=1)
    for o in out[:10]:
        print(f"{o['kernel'][:48]:48s} grid={o['grid_size']:>10d} n={o['launches']:3d} tag={str(o['tag']):20s} "
              f"traffic={o['traffic_bytes_per_launch'] / 1e9:6.2f} GB  l2 hit {o.get('l2_hit_rate', float('nan')):.2f}  "
              f"avg {o.get('avg_duration_us', float('nan')):8.1f} us")
    if step is not None:
        print(json.dumps(step, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:10])
