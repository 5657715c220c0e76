#!/usr/bin/env python3
"""Benchmark of the BioMedKG GCL + KGE training hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # starts its own N ranks (launch_ranks), one per GPU, RCCL
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W   # ... under an outer launcher: its rank 0 does the same
                                                                  # (take_over_from_outer_launcher), the other workers leave

One *step* = one GCL training step (GGD, the reference's default ``model_name``; 4-layer GCN
768->256) + one KGE training step (4-layer R-GCN d=256 + DistMult, ``neg_ratio`` negatives per
positive, BCE + L2 reg) over the same synthetic PrimeKG++-shaped graph (BASELINE.json configs[2]:
~130k nodes / ~4M edges / 30 relations), each step = forward + loss + backward + grad-clip 1.0 +
Adam (clip + update fused: biomedkg_amd/optim.py), inputs resident in HBM.  ``value`` = input edges / step time (every input edge counted once
although both sub-steps consume it), whole job.

N = 1 runs the whole graph on the GPU.  N > 1, ``--mode auto`` runs BOTH multi-GPU regimes one after the other, each on N
fresh ranks, and prints ONE line (``launch_ranks``):
* the headline -- ``value``, ``ms_per_step``, ``"scaling": "strong"``, ``"mode": "partition"`` -- is BASELINE configs[3] /
  north_star's node sharding: ONE graph, its nodes partitioned over the ranks (biomedkg_amd/partition.py): per conv layer a
  gather of the table rows a rank's in-edges read (forward) and of the output-gradient rows its out-edges point at
  (backward), every triplet scored by the owners of both its endpoints (no gradient rows cross the wire), flat-bucket
  gradient all-reduce.  Rounds 5 and 6 measured each rank's compute on one GPU (scripts/simulate_ranks.py): at most
  1.5-1.7x at 8 GPUs whatever the wire does -- a rank's share of a step that takes one GPU 8-9 ms is below the floor of
  the ~350 kernels that carry it;
* ``replicas`` on the same line -- its own ``value`` / ``ms_per_step`` / ``"scaling": "weak"`` -- is the reference's own
  multi-GPU regime (Lightning DDP: train_gcl.py:63-68, configs/gcl.yaml:11): one PrimeKG++-shaped graph per rank, the ranks'
  gradients averaged by one flat RCCL all-reduce per sub-step (the KGE bucket under the next GGD sub-step), per-GPU work
  fixed, value = the edges of all ranks' graphs / step time.
``--mode partition`` / ``--mode replicas`` run one of them.  (Rounds 1-4 reported the partition as the N-GPU figure, round 5
the replicas: the two are different workloads, hence both, each labelled.)
Default ``--precision bf16`` = BASELINE's "bf16 storage / fp32 accumulate" configuration; ``--precision fp32``
is the exact parity mode (its step time is reported on the same line as ``ms_per_step_fp32``).

In the full record (rank 0, N = 1): ``roofline`` for the kernel with the most GPU time among the gather /
aggregate launches (HIP events on the launch stream inside the timed region; see ``roofline_of``), ``kernels`` for the
others, ``grace_step`` (BASELINE configs[1]: GRACE / GCN d=128 on the 5k/50k graph, fp32) and ``ntxent`` (the NT-Xent
matrix-core kernels against the fp32 / bf16 MFMA peaks), ``fusion`` (the modality projection), ``sampled_regime`` (the
reference's own GraphSAINT mini-batch KGE loop: eager steps/s vs one hipGraph replay per step), and ``cpu_baseline``
(the CPU oracle -- a torch restatement of the reference's PyG path -- timed live on this box's host cores: full training
steps on a 1/8-scale graph for ~15 s; the full-size step, 100-110 s on 64 cores, is measured once per round and read from
profiles/).

stdout's LAST line is a compact (< 8 KB) strict-JSON object -- the contract's keys, ``roofline``, ``cpu_baseline``,
``exchange`` for N > 1 (``compact_line``); the full record (the other kernels, step accounting, every side figure) is written
to ``bench_detail.json`` next to this file.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def clip_adam_step(opt, max_norm):
    """torch.nn.utils.clip_grad_norm_(params, max_norm) + opt.step() as two launches (biomedkg_amd/optim.py: the same
    update on the torch.optim.Adam object's own state)."""
    from biomedkg_amd.optim import clip_adam_step as fused
    return fused(opt, max_norm)

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0        # HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
L2_GATHER_GBS = 16800.0      # "Indexed rows: gather": rows served by the XCD's L2, 16.8-18.8 TB/s chip-wide
MALL_GATHER_GBS = 8600.0     # ... rows served by the Infinity Cache (38 MB table), 8.6 TB/s
MFMA_F32_TFLOPS = 155.0      # v_mfma_f32_32x32x2_f32, measured
MFMA_BF16_TFLOPS = 2500.0    # dense bf16


# what ``frac`` means, once (bench_detail.json ``roofline_model``), not per kernel
ROOFLINE_MODEL = {
    "guide_GBs": {"hbm_spec": HBM_PEAK_GBS, "hbm_achievable": 6300.0, "infinity_cache_gather": MALL_GATHER_GBS,
                  "l2_gather": L2_GATHER_GBS},
    "frac_definition": "frac = t_min / t with t_min = max(algorithmic bytes / 16.8 TB/s [every gathered byte through "
                       "the XCD L2s: the guide's L2 gather rate], L2-miss bytes / 8.6 TB/s [PMC FETCH_SIZE x 2 + "
                       "WRITE_SIZE per call out of the Infinity Cache: the guide's gather rate; those bytes include "
                       "Infinity-Cache hits, guide l. 297]) and t = the call's HIP-event time (avg_launch_ms; "
                       "avg_launch_ms_traced = the same call in the committed --kernel-trace pass).  Components: "
                       "frac_l2_gather = t_min_l2 / t, frac_mall_gather = t_min_mall / t, frac_hbm_spec = L2-miss "
                       "rate / 8 TB/s.  achieved / peak = algorithmic_over_hbm is NOT a fraction (cache reuse); frac "
                       "is null when no PMC pass of this call is committed under profiles/"}


class Batch:
    pass


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=130_000)
    ap.add_argument("--edges", type=int, default=4_000_000)
    ap.add_argument("--relations", type=int, default=30)
    ap.add_argument("--neg-ratio", type=int, default=1)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16",
                    help="matrix-core operand precision (BASELINE configs[2] is the bf16 config; fp32 = exact mode)")
    ap.add_argument("--mode", choices=["auto", "replicas", "partition"], default="auto",
                    help="auto: the whole graph on one GPU; over N > 1 GPUs BOTH regimes on one line -- headline: partition "
                         "(ONE graph, its nodes partitioned over the N GPUs: BASELINE configs[3], strong scaling), beside it: "
                         "replicas (one graph per rank with a flat gradient all-reduce: the reference's DDP regime, weak "
                         "scaling).  partition / replicas: that regime alone")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the fp32 step / GRACE / NT-Xent / fusion figures")
    ap.add_argument("--cpu-scale", type=int, default=8, help="the live CPU baseline probe runs on a 1/scale graph")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (kernels, step accounting, NT-Xent / fusion / sampled-regime figures); "
                         "stdout's last line is the compact (< 8 KB) strict-JSON line")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------- roofline
def pmc_traffic(tag, precision):
    """Bytes per call that left the XCD L2s for a tagged library call, from the committed rocprofv3 --pmc passes
    (profiles/*pmc_traffic.json, scripts/pmc_summary.py: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 FETCH_SIZE x2
    correction; a call's launches -- e.g. the item kernel and the fix-up / short-row kernel of one SpMM -- are summed);
    plus the L2 hit rate of those launches when a TCC_HIT / TCC_MISS pass is there.  The newest file that has the
    (tag, precision) wins; (None, None) if this call was not profiled."""
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
            hits = [k for k in json.load(open(path))["kernels"]
                    if (k.get("tag") == tag or tag in (k.get("tags") or [])) and k.get("precision") == precision]
            if hits:
                # a call = one launch of its main kernel (+ its second kernel); a tag may cover several shapes of the same
                # call (the positive and the negative triplet set): bytes over all launches / number of calls
                # (a fix-up kernel may also run behind ANOTHER call -- the one-pass walk's long rows end in the node-side
                # backward's fix-up kernel --: calls are counted on the main kernels)
                main = [k for k in hits if "fixup" not in k["kernel"]] or hits
                per_kernel = {}
                for k in main:
                    per_kernel[k["kernel"]] = per_kernel.get(k["kernel"], 0) + k["launches"]
                calls = max(per_kernel.values())      # (a group shared by several tags -- "tags" -- averages over their calls)
                total = sum(k["traffic_bytes_per_launch"] * k["launches"] for k in hits) / calls
                h = sum(k.get("tcc_hit", 0) * k["launches"] for k in hits)
                m = sum(k.get("tcc_miss", 0) * k["launches"] for k in hits)
                # the call's duration from the --kernel-trace pass of the same command (scripts/pmc_summary.py): its
                # launches' mean durations, summed -- the figure the live HIP-event time must agree with
                traced = None
                if all("avg_duration_us" in k for k in hits):
                    per = {}
                    for k in main:
                        per[k["kernel"]] = per.get(k["kernel"], 0) + k["launches_traced"]
                    traced = sum(k["avg_duration_us"] * k["launches_traced"] for k in hits) / max(per.values()) / 1e3
                return total, (h / (h + m) if (h + m) else None), os.path.basename(path), traced
    except Exception:  # noqa: BLE001 - reporting only
        return None, None, None, None
    return None, None, None, None


def step_accounting(precision):
    """The step-level object of the newest committed PMC / trace summary (profiles/*pmc_traffic.json ``step``): traced ms
    per step, the tagged calls' share and the untagged remainder by kernel family."""
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")), reverse=True):
            step = (json.load(open(path)).get("step") or {}).get(precision)
            if step and "error" not in step:
                return dict(step, source=os.path.basename(path))
    except Exception:  # noqa: BLE001 - reporting only
        pass
    return None


KERNEL_NAMES = {
    "rgcn_segment_mean": "spmm_items_kernel<{t},{p},1> + spmm_fixup_kernel (long-row fix-up, short rows): R-GCN (relation, "
                         "target) segment-mean gather/aggregate, d={d}",
    "gcn_propagate": "spmm_items_kernel<{t},{p},1> + spmm_fixup_kernel: GCN A_hat (X W^T) gather/aggregate, d={d}",
    "gcn_propagate_views": "spmm_items_kernel<{t},{p},1> + spmm_fixup_kernel: GCN A_hat (X W^T) gather/aggregate, two GGD "
                           "views side by side, d={d} (one launch pair; each view's 256 columns on four of the eight XCDs)",
    "gcn_propagate_views_bwd": "spmm_items_kernel<{t},{p},1> + spmm_fixup_kernel: the table gradient A_hat^T dOut of the two "
                               "GGD views (bf16 dOut gathered over the out-edge CSR), d={d}",
    "gcn_propagate_views_kept": "spmm_items_kernel<{t},{p},1> + spmm_fixup_kernel: the two GGD views of an AUGMENTED step "
                                "(dropout_edge p = 0.4): the propagate over the view's kept edges only (bkg_csr_compact), d={d}",
    "gcn_propagate_views_bwd_kept": "spmm_items_kernel<{t},{p},1> + spmm_fixup_kernel: the table gradient of an augmented "
                                    "step's two views over the kept out-edges, d={d}",
    "kge_fused_bce": "kge_fused_bce_kernel<{t}>: DistMult scores + BCE terms + d score + d rel_emb partials, one triplet set, d={d}",
    "kge_bwd_nodes": "kge_incidence_streams_kernel + kge_bwd_stream_kernel<{t}> + fixup: dz by per-node segment reduce over the "
                     "triplet incidences (their other-endpoint / relation / g streams in list order), d={d}",
    "kge_onepass": "kge_onepass_kernel<{t}> + fixup: the whole DistMult training pass of the POSITIVES in one walk over their "
                   "(node, role, relation)-sorted incidence lists (scores, BCE terms, dz, one d rel_emb record row per run), d={d}",
    "kge_onepass_mx": "kge_onepass_mx_kernel: the whole DistMult training pass of the step's fresh NEGATIVES in one walk over "
                      "their (node, role)-sorted incidence lists (BCE terms, dz; d rel_emb as a one-hot product on the matrix "
                      "cores), d={d}",
}


def roofline_of(tag, rec, precision):
    """One tagged gather / aggregate call.  ``achieved`` = algorithmic bytes / HIP-event time on the launch
    stream.  The gather tables (66-133 MB) are resident in the 256 MiB Infinity Cache and their hot rows in the XCD L2s,
    so the ALGORITHMIC rate exceeds the 8 TB/s HBM figure (``algorithmic_over_hbm`` > 1): it is not an HBM fraction.
    What the memory side sees is ``traffic`` = the bytes that left the XCD L2s (PMC FETCH_SIZE x 2 + WRITE_SIZE per
    call); MI355X_MICROARCH.md "HBM" (l. 297): those counters are the L2's fabric-side requests and Infinity-Cache hits
    ARE counted in them, so ``l2_miss_GBs = traffic / time`` is L2-miss traffic served by Infinity Cache + HBM together,
    not HBM alone.

    The model behind ``frac`` (round 4): a row gather moves two streams that overlap in time -- ALL gathered bytes
    through the XCD L2s (the guide's L2 gather rate, 16.8 TB/s) and the L2 MISSES through the fabric out of the Infinity
    Cache (its gather rate out of the Infinity Cache, 8.6 TB/s) -- so the fastest this call could run at the measured hit
    rate is ``t_min = max(algorithmic / 16.8 TB/s, l2_miss_bytes / 8.6 TB/s)`` and ``frac = t_min / t``.  The three
    component fractions are kept: ``frac_l2_gather``, ``frac_mall_gather``, ``frac_hbm_spec`` (L2-miss traffic against
    the 8 TB/s HBM spec, north_star's "HBM roofline": an upper bound of the true HBM fraction)."""
    from biomedkg_amd import ops
    durs = [a.elapsed_time(b) for a, b in rec["events"]]
    avg_ms = sum(durs) / len(durs)
    ops.timer_bytes(rec)                    # (a dropout_edge view's launches: bytes of the entries it kept)
    gbs = rec["bytes"] / (avg_ms * 1e-3) / 1e9
    traffic, l2_hit, source, traced_ms = pmc_traffic(tag, precision)
    miss_gbs = (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None
    d = rec.get("d", 0)
    halves = rec.get("col_halves", 1)       # 2: the launch runs the two column halves as two problems (csrc/spmm.hip)
    pieces = max(d // halves // 8, 1) if precision == "bf16" else max(d // 4, 1)
    name = KERNEL_NAMES.get(tag, tag).format(t="__bf16" if precision == "bf16" else "float", p=min(pieces, 64), d=d)
    f_hbm = (miss_gbs / HBM_PEAK_GBS) if miss_gbs else None
    f_mall = (miss_gbs / MALL_GATHER_GBS) if miss_gbs else None
    f_l2 = gbs / L2_GATHER_GBS
    t_l2 = rec["bytes"] / (L2_GATHER_GBS * 1e9) * 1e3                       # ms: every gathered byte through the L2s
    t_mall = (traffic / (MALL_GATHER_GBS * 1e9) * 1e3) if traffic else None   # ms: the misses out of the Infinity Cache
    t_min = max(t_l2, t_mall) if t_mall is not None else None
    return {"bound": "hbm", "kernel": name, "tag": tag, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (t_min / avg_ms) if t_min is not None else None, "traffic": traffic,
            "t_min_ms": t_min, "t_min_l2_ms": t_l2, "t_min_mall_ms": t_mall,
            "bound_by": (None if t_min is None else ("l2_misses_vs_infinity_cache_gather_rate" if t_mall >= t_l2
                                                     else "gathered_bytes_vs_l2_gather_rate")),
            "l2_miss_GBs": miss_gbs, "frac_hbm_spec": f_hbm, "frac_l2_gather": f_l2, "frac_mall_gather": f_mall,
            "algorithmic_over_hbm": gbs / HBM_PEAK_GBS, "reuse": (rec["bytes"] / traffic) if traffic else None,
            "l2_hit_rate": l2_hit, "pmc_source": source,
            "avg_launch_ms": avg_ms, "avg_launch_ms_traced": traced_ms,
            "traced_over_live": (traced_ms / avg_ms) if traced_ms else None,
            "launches": len(durs), "ms_total": sum(durs), "bytes_per_launch": rec["bytes"],
            "grid_size": rec.get("grid"), "grid_size_second_launch": rec.get("grid2"), "pieces": pieces, "col_halves": halves,
            }


# tagged calls that launch the SAME kernel over the same grid (the forward and the backward propagate of the GGD views): the
# dominant kernel is picked by their summed time, the call with the larger share stands for it
SAME_KERNEL = {"gcn_propagate_views_bwd": "gcn_propagate_views", "gcn_propagate_views_kept": "gcn_propagate_views",
               "gcn_propagate_views_bwd_kept": "gcn_propagate_views"}


def rooflines(timers, precision):
    out = [roofline_of(tag, rec, precision) for tag, rec in timers.items() if rec["events"]]
    group = {}
    for r in out:
        g = SAME_KERNEL.get(r["tag"], r["tag"])
        group[g] = group.get(g, 0.0) + r["ms_total"]
    for r in out:
        r["ms_total_of_kernel"] = group[SAME_KERNEL.get(r["tag"], r["tag"])]
    out.sort(key=lambda r: (-r["ms_total_of_kernel"], -r["ms_total"]))
    return (out[0] if out else None), out[1:]



# ----------------------------------------------------------------------------------------------- the line the driver reads
LINE_LIMIT = 8192          # the driver keeps an 8 KB tail of stdout: the LAST line must fit it with room to spare
_ROOFLINE_KEYS = ("bound", "tag", "kernel", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_ms",
                  "avg_launch_ms_traced", "l2_hit_rate", "l2_miss_GBs", "frac_hbm_spec", "frac_l2_gather", "frac_mall_gather",
                  "launches", "pmc_source")
_CPU_KEYS = ("value", "unit", "cores", "kind", "cpu_model", "seconds_per_step", "sample", "full_size", "threads1")
_EXCHANGE_KEYS = ("wire_bytes_received_per_step", "wire_bytes_sent_per_step", "exchanges_per_step", "exchange_exposed_ms",
                  "exchange_serial_ms", "exposed_fraction", "ms_per_step_overlapped", "ms_per_step_serial",
                  "host_enqueue_ms_per_step", "host_enqueue_ms_per_step_graphed", "library_calls_per_step",
                  "padded_over_real_rows", "rows", "form", "bucket_bytes_ggd", "bucket_bytes_kge", "allreduce_alone_ms_ggd",
                  "allreduce_alone_ms_kge")
_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "ranks", "backend", "mode", "steps", "warmup", "ms_per_step",
             "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
_SIDE_KEYS = ("triplets_per_sec_kge_step", "edges_per_sec_gcl_step", "ms_per_step_median", "ms_per_step_max", "ms_gcl_step",
              "ms_kge_step", "ms_per_step_fp32", "ms_per_step_literal_ggd_last_layer", "ms_per_step_fresh_first_layer_means",
              "ms_per_step_grouped_neg_types", "ms_per_step_all_literal",
              "ms_kge_step_neg10", "triplets_per_sec_kge_step_neg10")


def _strict(o, digits=6):
    """JSON the strictest reader takes: non-finite floats -> null, floats to ``digits`` significant digits, tuples ->
    lists, anything else that json cannot carry -> its str()."""
    import math
    if isinstance(o, bool) or o is None or isinstance(o, (int, str)):
        return o
    if isinstance(o, float):
        if not math.isfinite(o):
            return None
        return float(f"{o:.{digits}g}")
    if isinstance(o, dict):
        return {str(k): _strict(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_strict(v, digits) for v in o]
    if hasattr(o, "item"):
        try:
            return _strict(o.item(), digits)
        except Exception:      # noqa: BLE001
            pass
    return str(o)


def _short(text, limit):
    text = " ".join(str(text).split())
    return text if len(text) <= limit else text[:limit - 3] + "..."


def _pick(d, keys, text_limit=160):
    if not isinstance(d, dict):
        return None
    out = {}
    for k in keys:
        if k in d:
            v = d[k]
            out[k] = _short(v, text_limit) if isinstance(v, str) else v
    return out


def compact_line(out, detail_path=None):
    """The ONE line the driver parses (a 26 KB line came back ``parsed: null``): the contract's keys, the
    dominant kernel's ``roofline`` and ``cpu_baseline`` as numbers plus short labels, ``exchange`` numbers for N > 1, a few
    headline side figures.  Everything else -- the other kernels, the step accounting, NT-Xent / fusion / sampled-regime
    figures, the prose definitions -- lives in ``bench_detail.json`` (``detail``).  -> a strict-JSON string, always shorter
    than LINE_LIMIT (the optional parts are dropped in order if a future field pushes it over)."""
    line = {k: out.get(k) for k in _TOP_KEYS}
    cfg = out.get("config") or {}
    line["config"] = {"workload": _short(cfg.get("workload", ""), 360), "parallelism": cfg.get("parallelism")}
    roof = _pick(out.get("roofline"), _ROOFLINE_KEYS, 200)
    line["roofline"] = roof
    cpu = out.get("cpu_baseline")
    if isinstance(cpu, dict) and "error" in cpu:
        line["cpu_baseline"] = {"error": _short(cpu["error"], 200)}
    else:
        cpu = _pick(cpu, _CPU_KEYS, 260)
        if cpu:
            for k in ("full_size", "threads1"):
                if isinstance(cpu.get(k), dict):
                    cpu[k] = {a: b for a, b in cpu[k].items() if not isinstance(b, (str, dict, list)) or a == "source"}
        line["cpu_baseline"] = cpu
    if out.get("exchange"):
        ex = _pick(out["exchange"], _EXCHANGE_KEYS)
        gs = out["exchange"].get("graphed_sub_steps")
        ex["graphed_sub_steps"] = ({k: (_short(v, 120) if isinstance(v, str) else v) for k, v in gs.items()}
                                   if isinstance(gs, dict) else None)
        line["exchange"] = ex
    for k in _SIDE_KEYS:
        if out.get(k) is not None:
            line[k] = out[k]
    for key in ("check_loss", "check_loss_after", "partition_bound"):
        if out.get(key):
            line[key] = {k: v for k, v in out[key].items() if not isinstance(v, str) or k == "source"}
    line["detail"] = detail_path
    line = _strict(line)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    for drop in ("check_loss_after", "check_loss", "partition_bound", "exchange.rows", "detail", "triplets_per_sec_kge_step_neg10", "ms_kge_step_neg10", "ms_per_step_grouped_neg_types", "ms_per_step_literal_ggd_last_layer", "ms_per_step_fresh_first_layer_means",
                 "ms_per_step_all_literal", "ms_per_step_fp32", "exchange.graphed_sub_steps", "cpu_baseline.threads1", "cpu_baseline.full_size"):
        if len(text) < LINE_LIMIT - 1024:
            break
        head, _, leaf = drop.partition(".")
        if leaf:
            if isinstance(line.get(head), dict):
                line[head].pop(leaf, None)
        else:
            line.pop(head, None)
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:      # cannot happen with the fields above; never print a line the driver cannot read
        line = {k: line.get(k) for k in _TOP_KEYS + ("config",)}
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    return text


def emit(out, detail_path):
    """Write the full record to ``detail_path`` (and to gpurun_out/ when that directory exists: it is what travels back
    from a GPU box), then print the compact line LAST on stdout."""
    detail = _strict(out, digits=9)
    wrote = None
    for path in [detail_path] + ([os.path.join(ROOT, "gpurun_out", os.path.basename(detail_path))]
                                  if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else []):
        try:
            with open(path, "w") as f:
                json.dump(detail, f, allow_nan=False, indent=1)
            wrote = wrote or os.path.relpath(path, ROOT)
        except OSError as why:
            sys.stderr.write(f"bench.py: could not write {path}: {why}\n")
    try:      # RCCL writes its version banner through C stdio: flush it out BEFORE the line, which must come last
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:      # noqa: BLE001
        pass
    sys.stdout.flush()
    print(compact_line(out, wrote), flush=True)


# ----------------------------------------------------------------------------------------------- sub-steps
def make_step(module, batch, world, overlap=False):
    """One sub-step: zero_grad, training_step, backward, gradient all-reduce, clip, Adam.

    ``overlap`` (multi-rank only): the all-reduce is started right after backward and finished -- wait, clip,
    Adam -- at the top of this sub-step's NEXT call (or by ``step.finish()``).  The two sub-steps of the bench own
    disjoint parameters, so the KGE bucket (32.5 MB) travels over xGMI while the next GGD sub-step computes; the
    update still lands before the KGE parameters are read again, i.e. the arithmetic of the loop is unchanged."""
    from biomedkg_amd.distributed import GradBucket
    params = [p for p in module.model.parameters()]
    opt = torch.optim.Adam(params, lr=1e-3, fused=True)   # torch's single-kernel Adam (same update; ~0.4 ms per step)
    bucket = GradBucket(params, world)   # one flat RCCL all-reduce per sub-step (<= 49 MB)
    state = {"pending": False}

    def finish():
        if state["pending"]:
            bucket.finish()
            clip_adam_step(opt, 1.0)
            state["pending"] = False

    def step():
        finish()
        opt.zero_grad(set_to_none=True)
        loss = module.training_step(batch)
        loss.backward()
        if overlap and bucket.live:
            bucket.begin()
            state["pending"] = True
        else:
            bucket.all_reduce()
            clip_adam_step(opt, 1.0)
        return loss

    step.finish = finish
    step.batch = batch
    step.params = params
    step.bucket = bucket
    return step


def build_partitioned_steps(args, rank, world, dev, simulate=False):
    """The same pair of sub-steps over ONE graph whose nodes are partitioned over the ranks: GGD with its
    augmentation (coin, feature mask, dropout_edge -- the replicated draws come from a generator every rank seeds alike)
    through the fused side-by-side views, R-GCN + DistMult with each rank scoring its share of the positives and its
    own negatives.

    ``simulate``: THIS process stands in for rank ``rank`` of ``world`` without a process group (scripts/simulate_ranks.py:
    the rank's real structures and kernels, every exchange a local copy of the same byte count -- partition.Comm)."""
    from biomedkg_amd import ops, partition as P, sampling, synthetic
    from biomedkg_amd.distributed import GradBucket
    from biomedkg_amd.gcl_module import GGDModule
    from biomedkg_amd.kge_module import KGEModule
    g = synthetic.primekg_like(args.nodes, args.edges, args.relations, seed=42)     # the same graph on every rank
    n, e, r, h = g["num_nodes"], g["edge_index"].size(1), g["num_relations"], args.hidden
    ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
    part = P.NodePartition(ei, n, rank, world, simulate=simulate)
    # refuse early, with a message, what would otherwise die inside a step: the gathered tables live in the padded slot
    # space [world * n_max, d] -- per conv layer a forward table and a gathered output gradient, two views wide for GGD,
    # plus the activations saved for the backward (a generous bound: fp32, every layer resident at once)
    need = part.world * part.n_max * (2 * h) * 4 * 2 * 4 * 2
    free, _ = torch.cuda.mem_get_info(dev)
    if need > 0.9 * free:
        raise SystemExit(f"bench.py: rank {rank} would need ~{need / 2**30:.1f} GiB for the gathered tables of {world} x "
                         f"{part.n_max} slots x {2 * h} columns but the device has {free / 2**30:.1f} GiB free "
                         "(BKG_BENCH_ONE_DEVICE shares one GPU among all ranks: use a smaller --nodes / --edges there)")
    torch.manual_seed(42)
    xk = part.shard(torch.randn(n, h, device=dev)).contiguous()                     # each rank keeps only its rows
    xg = part.shard(torch.randn(n, 768, device=dev)).contiguous()
    torch.manual_seed(1234)
    kge = KGEModule("rgcn", "dismult", h, h, h, 2, r, 2, "cosine", 1e-3, 0.2, "none", args.neg_ratio, "random").to(dev)
    gcl = GGDModule(in_dim=768, hidden_dim=h, out_dim=h, num_hidden_layers=2, fuse_method="none").to(dev)
    kge.train(); gcl.train()
    pg = P.PartitionedGraph(part, ei, P.HipBackend)
    prg = P.PartitionedRelGraph(part, ei, et, r, P.HipBackend)
    # both-owner scoring (partition.TripletShare): the rank scores every triplet that touches one of its nodes -- counted
    # by the owner of its head -- and reduces the embedding gradient of its own rows; no dz ever crosses the wire
    pos = P.TripletShare(part, ei, et)
    neg_sampler = sampling.BlockNegativeSampler(ei, et, part.bounds, rank, args.neg_ratio * e, num_relations=r,
                                                n_max=part.n_max)
    shared = torch.Generator(device=dev).manual_seed(7)       # replicated draws: identical on every rank
    shared_host = torch.Generator().manual_seed(7)            # ... the augmentation coin / the negatives' seed: host draws
    own = torch.Generator().manual_seed(1000 + rank)          # per-rank seeds for the in-kernel streams
    counter = [0]
    from biomedkg_amd import rand as R
    shared_rand = R.DeviceRand(seed=777)                      # replicated device draws (GGD's corruption permutation)

    def own_seed():
        return int(torch.randint(0, 1 << 31, (1,), generator=own))

    def dropout_of(rows, cols, device):
        counter[0] += 1
        return ops.Dropout(0.2, keep=None, seed=own_seed(), stream_id=counter[0] & 0xFFFFFFFF)

    # BKG_BENCH_PARTITION_GRAPH=1 (RCCL only): each sub-step -- zero_grad, loss, backward, the gradient all-reduce, clip,
    # Adam, every exchange in between -- is captured once per host-side branch (GGD's augmentation coin) as ONE hipGraph
    # and replayed: one host call per sub-step instead of ~350 launches (at one rank the host needs 3.6 ms to enqueue an
    # eager partitioned step: a rank whose share of the kernels is shorter than that -- P >= 4 -- is launch-bound).  What
    # makes it capturable: every extent is static (patterns, block-structured negatives), no host read inside a step, the
    # in-kernel random streams are re-keyed by the library's device replay counter, torch's device draws come from a
    # generator registered with the graph.  If a capture raises on any rank, all ranks leave (exit code 3) and the launcher restarts the job with the eager step.
    graph_mode = (os.environ.get("BKG_BENCH_PARTITION_GRAPH") == "1"
                  and (simulate or (torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl")))
    graph_stats = {"captured": 0, "replayed": 0, "eager": 0, "refused": None}

    def capture_agreed(ok):
        if simulate or not torch.distributed.is_initialized() or torch.distributed.get_world_size() == 1:
            return ok
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        return bool(flag.item())

    def make(params, loss_fn, draw_key, defer_update=False):
        """``defer_update`` (graphed mode, the KGE sub-step): its 32.5 MB gradient bucket -- the one message of the step that
        does not shrink with P -- is reduced on a stream and a communicator of its own WHILE the next GGD sub-step runs, and
        the update (scale, clip, Adam) lands at the top of this sub-step's next call, before its parameters are read again:
        three graphs (loss + backward + pack | all-reduce | update) instead of one, the arithmetic of the loop unchanged
        (what ``make_step(overlap=True)`` does for the replicated runs)."""
        from biomedkg_amd import _lib as L
        from biomedkg_amd.graph import as_captured
        opt = torch.optim.Adam(params, lr=1e-3, fused=True)
        if graph_mode:
            from biomedkg_amd.graphs import make_capturable
            opt = make_capturable(opt)
        defer = bool(defer_update and graph_mode and not simulate and os.environ.get("BKG_BENCH_DEFER_UPDATE", "1") != "0")
        group = torch.distributed.new_group() if defer else None      # (every rank makes it: the same call order everywhere)
        bucket = GradBucket(params, world, group=group, simulate=simulate)
        defer = defer and bucket.live
        epoch = torch.zeros(1, dtype=torch.int32, device=dev)
        graphs, losses = {}, {}
        side = torch.cuda.Stream() if graph_mode else None
        comm_stream = torch.cuda.Stream() if defer else None
        state = {"on": graph_mode, "pending": False}
        tail = {}                                       # the deferred update's two graphs: "reduce", "update"

        def body(key, whole=True):
            if graph_mode:
                epoch.add_(1)
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(key)
            loss.backward()
            if whole:
                bucket.all_reduce()
                clip_adam_step(opt, 1.0)
            else:
                bucket.pack()                           # (the reduce and the update follow as graphs of their own)
            return loss

        def reduce_body():
            torch.distributed.all_reduce(bucket._flat, group=group)

        def update_body():
            bucket._flat.div_(world)
            for p_, v_ in zip(bucket.params, bucket._views):
                p_.grad = v_                            # the update reads the averaged bucket in place
            clip_adam_step(opt, 1.0)

        def run_tail(name, fn, stream):
            """One of the deferred update's two pieces on ``stream``: eagerly the first time, captured the second, replayed
            from then on (the same ladder as the sub-steps' own graphs)."""
            with torch.cuda.stream(stream):
                g = tail.get(name)
                if g is None:
                    fn()
                    tail[name] = False                  # seen once: capture next time
                elif g is False:
                    torch.cuda.synchronize()
                    time.sleep(0.5)                     # (the watchdog and the eager collectives just issued: see below)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                        fn()
                    tail[name] = g
                    g.replay()
                else:
                    g.replay()

        def finish_pending():
            if not state["pending"]:
                return
            state["pending"] = False
            outer = torch.cuda.current_stream()
            side.wait_stream(comm_stream)               # the averaged bucket has arrived
            side.wait_stream(outer)
            run_tail("update", update_body, side)
            outer.wait_stream(side)

        def step():
            key = draw_key()
            if not state["on"] or step.force_eager:
                finish_pending()
                graph_stats["eager"] += 1
                return body(key)
            if defer:
                finish_pending()                        # the previous call's update, before the parameters are read
            L.call("bkg_set_rng_epoch", L.ptr(epoch))
            outer = torch.cuda.current_stream()
            side.wait_stream(outer)
            try:
                with torch.cuda.stream(side):
                    g = graphs.get(key)
                    if g is None:
                        with as_captured():
                            loss = body(key, whole=not defer).detach().clone()   # this kind's first step, eagerly (it also warms the pools)
                        torch.cuda.synchronize()
                        # the process group's watchdog thread polls the end events of the eager collectives just issued; let
                        # it retire them before a capture begins (an event query from that thread while this one captures
                        # ends the process: hipErrorCapturedEvent, seen once in four runs without the pause), and capture in
                        # thread-local mode so that its later calls are none of this capture's business
                        time.sleep(0.5)
                        g = torch.cuda.CUDAGraph()
                        g.register_generator_state(shared)
                        refused = None
                        try:
                            with as_captured(), torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                                losses[key] = body(key, whole=not defer).detach()
                        except Exception as why:                    # noqa: BLE001
                            torch.cuda.synchronize()
                            refused = f"{type(why).__name__}: {why}"[:300]
                        # the outcome is agreed on by ALL ranks before anyone goes on: a rank that dropped to
                        # the eager step on its own would issue another sequence of collectives than the ranks replaying
                        # their graphs, and the job would hang until the launcher's limit.  One eager MIN all-reduce of an
                        # ok flag per capture (three per run); if any rank's capture failed every rank leaves with exit
                        # code 3 and the launcher starts the job again with the eager step (launch_ranks).
                        if not capture_agreed(refused is None):
                            graph_stats["refused"] = refused or "another rank's capture failed"
                            sys.stderr.write(f"bench.py rank {rank}: capture of a partitioned sub-step refused "
                                             f"({graph_stats['refused']}); leaving so that the eager step can run\n")
                            sys.stderr.flush()
                            os._exit(3)
                        graphs[key] = g
                        graph_stats["captured"] += 1
                        if defer:
                            start_reduce()
                        return loss
                    g.replay()
                    ops.note_graph_replay()
                    graph_stats["replayed"] += 1
                    if defer:
                        start_reduce()
                    return losses[key]
            finally:
                outer.wait_stream(side)
                L.call("bkg_set_rng_epoch", None)

        def start_reduce():
            comm_stream.wait_stream(side)               # the packed bucket is complete
            run_tail("reduce", reduce_body, comm_stream)
            state["pending"] = True
            graph_stats["deferred_updates"] = graph_stats.get("deferred_updates", 0) + 1
        step.finish = finish_pending
        step.force_eager = False
        return step

    def ggd_key():
        return float(torch.rand(1, generator=shared_host)) < gcl.model.p             # reference gcl.py:75-76 (a host coin)

    def ggd_loss(aug):
        x, keep = xg, None
        if aug:
            x = ops.masked_bf16(xg, 0.4, own_seed(), rank) if ops.bf16_storage() else xg * (torch.rand_like(xg) >= 0.4)
            keep = torch.rand(e, generator=shared, device=dev) >= 0.4
        perm = shared_rand.perm(n, dev)            # a keyed bijection, the same key on every rank (one launch, no sort)
        return P.ggd_loss_partitioned(gcl.model, x, pg, P.HipBackend, perm, dropout_of, keep=keep)

    def kge_loss(_key=None):
        # reference kge_module.py:86-102 over a partition: every rank draws the blocks of the cell grid that touch its
        # nodes from the step's SHARED seed (the two owners of a block get the same triplets); the relation ids --
        # edge_type.repeat(k), dealt to the blocks once, relation-grouped inside a block -- are constants of the partition
        seed = int(torch.randint(0, 1 << 62, (1,), generator=shared_host))
        smp = neg_sampler
        neg = P.TripletShare(part, None, smp.types, n_own=smp.n_own, entries=smp.entries, slots=smp.draw(seed),
                             types_own=smp.types_own, types_foreign=smp.types_foreign)
        return P.kge_loss_partitioned(kge.model, xk, prg, P.HipBackend, pos, neg, kge.model.decoder.kind, dropout_of,
                                      n_pos_global=e, n_neg_global=args.neg_ratio * e)

    def check_losses():
        """The two losses on FIXED draws (one corruption permutation, one edge mask, world-independent negatives), exact
        fp32, dropout off, at the initial weights: the same numbers at every world size (tests compare N ranks with one)."""
        was = ops.get_matmul_precision()
        ops.set_matmul_precision("fp32")
        gcl.eval(); kge.eval()
        try:
            gen = torch.Generator().manual_seed(99)
            perm = torch.randperm(n, generator=gen).to(dev)
            keep = (torch.rand(e, generator=gen) >= 0.4).to(dev)
            nt = et[torch.randperm(e, generator=gen).to(dev)]
            with torch.no_grad():
                lg = P.ggd_loss_partitioned(gcl.model, xg, P.PartitionedGraph(part, ei, P.HipBackend), P.HipBackend, perm,
                                            None, keep=keep)
                neg = sampling.negative_sampling(ei, n, e, seed=99)
                lk = P.kge_loss_partitioned(kge.model, xk, prg, P.HipBackend, pos, P.TripletShare(part, neg, nt),
                                            kge.model.decoder.kind, None, n_pos_global=e, n_neg_global=e)
            return {"ggd": float(P.loss_value(lg, world)), "kge": float(P.loss_value(lk, world)),
                    "note": "exact fp32, dropout off, fixed draws, initial weights: independent of the world size"}
        finally:
            ops.set_matmul_precision(was)
            gcl.train(); kge.train()

    check = check_losses()
    # both outcomes of GGD's augmentation coin and the KGE step once, untimed: whichever comes first inside the timed
    # region then finds its buffers in the caching allocator (a first-time hipMalloc costs ~100 ms)
    for loss_fn in (lambda: ggd_loss(True), lambda: ggd_loss(False), kge_loss, kge_loss):
        loss_fn().backward()
    for p in list(gcl.model.parameters()) + list(kge.model.parameters()):
        p.grad = None
    step_g = make(list(gcl.model.parameters()), ggd_loss, ggd_key)
    step_k = make(list(kge.model.parameters()), kge_loss, lambda: None, defer_update=True)
    step_g.part = part
    step_g.check_loss = check
    step_g.check_fn = check_losses
    step_g.graph_stats = graph_stats if graph_mode else None
    return step_g, step_k, n, e, r


def build_replica_steps(args, rank, world, dev):
    from biomedkg_amd import synthetic
    from biomedkg_amd.gcl_module import GGDModule
    from biomedkg_amd.kge_module import KGEModule
    g = synthetic.primekg_like(args.nodes, args.edges, args.relations, seed=42 + rank)
    torch.manual_seed(42 + rank)
    n, e, r, h = g["num_nodes"], g["edge_index"].size(1), g["num_relations"], args.hidden
    ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
    bk, bg = Batch(), Batch()
    bk.x, bk.edge_index, bk.edge_type = torch.randn(n, h, device=dev), ei, et
    bg.x, bg.edge_index = torch.randn(n, 768, device=dev), ei
    torch.manual_seed(1234)  # identical initial weights on every rank
    kge = KGEModule("rgcn", "dismult", h, h, h, 2, r, 2, "cosine", 1e-3, 0.2, "none", args.neg_ratio, "random").to(dev)
    gcl = GGDModule(in_dim=768, hidden_dim=h, out_dim=h, num_hidden_layers=2, fuse_method="none").to(dev)
    kge.train(); gcl.train()
    # both outcomes of GGD's augmentation coin and the KGE step once, untimed: whichever the seeded coin brings first
    # inside the timed region then finds its buffers in the caching allocator (a first-time hipMalloc costs ~100 ms)
    from biomedkg_amd import rand as R

    class Coin(R.DeviceRand):
        def __init__(self, value):
            super().__init__()
            self.value = value

        def coin(self):
            return self.value
    for value in (0.0, 1.0):
        with R.use(Coin(value)):
            gcl.training_step(bg).backward()
    kge.training_step(bk).backward()
    for p in list(gcl.model.parameters()) + list(kge.model.parameters()):
        p.grad = None
    return make_step(gcl, bg, world), make_step(kge, bk, world, overlap=True), n, e, r


# ----------------------------------------------------------------------------------------------- extras (N = 1)
def _event_ms(fn, reps, warm=1):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(x.elapsed_time(y) for x, y in evs)
    return t[len(t) // 2]


def ntxent_figures(dev):
    """The NT-Xent op (normalise + Gram tiles + finish) against the MFMA peaks: executed flops = M^2 h forward
    (symmetric: upper-triangular tiles only) and 4 M^2 h backward (Gram recomputed + W Z), M = 2N."""
    from biomedkg_amd import ops
    out = []
    for n, h, mode, reps in ((5000, 128, "fp32", 10), (130_000, 256, "bf16", 3), (130_000, 256, "fp32", 1)):
        torch.manual_seed(0)
        h1 = torch.randn(n, h, device=dev, requires_grad=True)
        h2 = (0.6 * h1.detach() + 0.4 * torch.randn(n, h, device=dev)).requires_grad_()
        m = 2 * n
        fwd = _event_ms(lambda: ops.ntxent_loss(h1, h2, 0.2, operands=mode), reps)
        loss = ops.ntxent_loss(h1, h2, 0.2, operands=mode)
        bwd = _event_ms(lambda: torch.autograd.grad(loss, (h1, h2), retain_graph=True), reps)
        peak = MFMA_F32_TFLOPS if mode == "fp32" else MFMA_BF16_TFLOPS
        tf_f, tf_b = m * m * h / fwd / 1e9, 4.0 * m * m * h / bwd / 1e9
        out.append({"N": n, "h": h, "operands": mode, "fwd_ms": fwd, "bwd_ms": bwd, "fwd_TFLOPs": tf_f, "bwd_TFLOPs": tf_b,
                    "peak_TFLOPs": peak, "fwd_frac": tf_f / peak, "bwd_frac": tf_b / peak})
        del h1, h2, loss
    return {"note": "whole op by HIP events (normalise, Gram-tile kernel, fixed-order finish); TFLOP/s on EXECUTED flops: "
                    "forward M^2 h (symmetric), backward 4 M^2 h, M = 2N; kernel-only times and counters: profiles/r03_ntxent_*",
            "cases": out}


def grace_figures(dev):
    """BASELINE configs[1]: one GRACE training step (GCN 768->128, num_hidden_layers=2 = 4 conv layers, AttentionFusion
    over [N,2,768], NT-Xent tau=0.2) on the 5k-node / 50k-edge graph, exact fp32."""
    from biomedkg_amd import ops, synthetic
    from biomedkg_amd.gcl_module import GRACEModule
    prev = ops.get_matmul_precision()
    ops.set_matmul_precision("fp32")
    try:
        ei = synthetic.gcl_small(5000, 50_000, seed=42).to(dev)
        b = Batch()
        b.x, b.edge_index = synthetic.lm_features(5000, 2, 768, seed=42).to(dev), ei
        torch.manual_seed(1)
        mod = GRACEModule(in_dim=768, hidden_dim=128, out_dim=128, num_hidden_layers=2, fuse_method="attention").to(dev).train()
        params = list(mod.model.parameters())
        opt = torch.optim.Adam(params, lr=1e-3, fused=True)

        def step():
            opt.zero_grad(set_to_none=True)
            loss = mod.training_step(b)
            loss.backward()
            clip_adam_step(opt, 1.0)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        # the same step captured once into a hipGraph and replayed (biomedkg_amd/graphs.py): one host call per step
        from biomedkg_amd import graphs
        torch.manual_seed(1)
        mod2 = GRACEModule(in_dim=768, hidden_dim=128, out_dim=128, num_hidden_layers=2, fuse_method="attention").to(dev).train()
        gstep = graphs.GraphedStep(mod2, b, torch.optim.Adam(mod2.model.parameters(), lr=1e-3, fused=True), clip=1.0, warmup=3)
        try:
            for _ in range(5):
                gstep()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                gstep()
            torch.cuda.synchronize()
            ms_graph = (time.perf_counter() - t0) / 100 * 1e3
        finally:
            gstep.close()
        return {"workload": "GRACE step, N=5000 E=50000, GCN 768->128 (4 conv layers), attention fuser, fp32, fwd+bwd+clip+Adam",
                "ms_per_step": ms_graph, "edges_per_sec": ei.size(1) / (ms_graph * 1e-3),
                "ms_per_step_eager": ms, "how": "one hipGraph replay per step (captured training_step + backward + clip + Adam); "
                                               "eager = one host launch per kernel"}
    finally:
        ops.set_matmul_precision(prev)


def fusion_figures(dev, edge_index=None, n=130_000):
    """The modality projection (reference fusion.py:17-31) over [N,2,768]: first call (QKV product + 2x2 attention + mean,
    3 x 2N x 768 x 768 x 2 flops) and the steady-state call (the fuser is frozen -- SURVEY 0.3-1 -- so the result is cached
    per (input, weights) version)."""
    from biomedkg_amd import ops, synthetic
    from biomedkg_amd.utils.fusion import AttentionFusion
    x = synthetic.lm_features(n, 2, 768, seed=3).to(dev)
    torch.manual_seed(2)
    f = AttentionFusion(768).to(dev).eval()
    out = {}
    for mode in ("bf16", "fp32"):
        prev = ops.get_matmul_precision()
        ops.set_matmul_precision(mode)
        try:
            x.add_(0.0)                                   # bump the version: the next call recomputes
            f(x)                                          # (once untimed: its buffers then come from the caching allocator)
            x.add_(0.0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            f(x)
            torch.cuda.synchronize(); first = (time.perf_counter() - t0) * 1e3      # wall clock of ONE cold call (host included)
            x.add_(0.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(x); e1.record()
            torch.cuda.synchronize(); first_dev = e0.elapsed_time(e1)               # the same call on the device's clock
            steady = _event_ms(lambda: f(x), 5)
            # nominal flops = the reference's literal form (q, k and v projections of all 2N token rows); the bf16 mode
            # evaluates it in projected form (ops._attention_fusion_projected: X (Wq^T Wk) over 2N rows + Wv over the N
            # combined rows = half the multiply-adds) -- both rates are given
            executed = (2.0 * 2 * n * 768 * 768 + 2.0 * n * 768 * 768) if (mode == "bf16" and ops.PROJECTED_ATTENTION) \
                else 2.0 * 2 * n * 768 * 2304
            out[mode] = {"first_call_ms": first, "first_call_device_ms": first_dev, "steady_ms": steady, "first_call_TFLOPs": 2.0 * 2 * n * 768 * 2304 / first / 1e9,
                         "first_call_executed_TFLOPs": executed / first / 1e9,
                         "form": "projected (two identities of the reference's arithmetic, DESIGN 4)"
                                 if (mode == "bf16" and ops.PROJECTED_ATTENTION) else "literal (q | k | v product, 2 x 2 attention)"}
        finally:
            ops.set_matmul_precision(prev)
    # ReDAF (reference fusion.py:49-90): dropout makes it new every training step -- nothing to cache -- so its cost is
    # per step: the transform product + scale / dropout / ReLU / modality mean, in node chunks (ops.redaf_fusion)
    from biomedkg_amd.utils.fusion import ReDAF
    torch.manual_seed(3)
    r = ReDAF(768).to(dev).train()
    redaf = {}
    for mode in ("bf16", "fp32"):
        prev = ops.get_matmul_precision()
        ops.set_matmul_precision(mode)
        try:
            ms = _event_ms(lambda: r(x), 5, warm=2)
            redaf[mode] = {"step_ms": ms, "TFLOPs": 2.0 * 2 * n * 768 * 768 / ms / 1e9}
        finally:
            ops.set_matmul_precision(prev)
    ggd = None
    if edge_index is not None and edge_index.numel() and int(n) == 130_000:
        # the GGD step of the headline with the [N, 2, 768] language-model features through the ReDAF fuser in front
        from biomedkg_amd.gcl_module import GGDModule
        torch.manual_seed(1234)
        mod = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="redaf").to(dev).train()
        b = Batch()
        b.x, b.edge_index = x, edge_index
        params = list(mod.model.parameters())
        opt = torch.optim.Adam(params, lr=1e-3, fused=True)

        def step():
            opt.zero_grad(set_to_none=True)
            mod.training_step(b).backward()
            clip_adam_step(opt, 1.0)
        prev = ops.get_matmul_precision()
        ops.set_matmul_precision("bf16")
        try:
            for _ in range(4):
                step()
            ggd = {"ms_per_step_bf16": _event_ms(step, 6, warm=1),
                   "note": "GGD / GCN 768->256 step over [N, 2, 768] features, fuse_method=redaf (the fuser runs every step)"}
        finally:
            ops.set_matmul_precision(prev)
    return {"workload": f"AttentionFusion over [{n}, 2, 768]", **out,
            "redaf": {"workload": f"ReDAF over [{n}, 2, 768], training mode (fresh dropout mask per call)", **redaf,
                      "ggd_step": ggd}}


def sampled_figures(dev, ei, et, n, r, walks=128, steps=150):
    """The reference's own KGE regime (train_kge.py:86: GraphSAINT random-walk batches, walk_length 10, neg_ratio 10,
    R-GCN 256-256-256 + DistMult) on the bench graph: steps/s of the eager loop (device sampler + ordinary step) and of
    the padded, sync-free loop replayed as one hipGraph per step (biomedkg_amd/sampled.py), fp32."""
    from biomedkg_amd import ops
    from biomedkg_amd.kge_module import KGEModule
    from biomedkg_amd.sampled import SaintGraphedKge
    from biomedkg_amd.samplers import SaintBatchLoader
    prec = ops.get_matmul_precision()
    ops.set_matmul_precision("fp32")
    try:
        torch.manual_seed(7)
        x = torch.randn(n, 256, device=dev) * 0.05
        m = KGEModule("rgcn", "dismult", 256, 256, 256, 2, r, 2, "cosine", 1e-3, 0.0, "none", 10, "random").to(dev).train()
        params = list(m.model.parameters())
        opt = torch.optim.Adam(params, lr=1e-3, fused=True)

        def eager(k):
            it = iter(SaintBatchLoader(x, ei, et, walks, walk_length=10, num_steps=k, seed=3))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            nodes = edges = 0
            for _ in range(k):
                b = next(it)
                opt.zero_grad(set_to_none=True)
                loss = m.training_step(b)
                loss.backward()
                clip_adam_step(opt, 1.0)
                nodes += b.x.size(0); edges += b.edge_index.size(1)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k, nodes / k, edges / k
        eager(10)
        t_eager, nodes, edges = eager(steps)
        out = {"sampler": f"GraphSAINT random walk, {walks} roots x 10 steps", "mean_nodes": round(nodes, 1),
               "mean_edges": round(edges, 1), "neg_ratio": 10, "eager_steps_per_s": round(1.0 / t_eager, 1)}
        for key, det in (("graphed_steps_per_s", False), ("graphed_deterministic_steps_per_s", True)):
            ops.set_deterministic(det)
            sg = SaintGraphedKge(m, x, ei, et, walks, 10, torch.optim.Adam(params, lr=1e-3, fused=True), clip=1.0, seed=3)
            for _ in range(60):
                sg.step()
            before = dict(sg.stats)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                sg.step()
            torch.cuda.synchronize()
            out[key] = round(steps / (time.perf_counter() - t0), 1)
            out[key.replace("steps_per_s", "stats")] = {k: sg.stats[k] - before[k] for k in sg.stats}
        ops.set_deterministic(False)
        out["speedup"] = round(out["graphed_steps_per_s"] / out["eager_steps_per_s"], 2)
        return out
    finally:
        ops.set_deterministic(False)
        ops.set_matmul_precision(prec)


def sampled_gcl_figures(dev, n=27_000, e=640_000, seeds=128, steps=40):
    """The reference's GCL regime (train_gcl.py:106-122, data_module.py:93-99: NeighborLoader batches of 128 seeds, fan-out
    [30] * 3, GGD over GCN 768 -> 256 with the attention fuser) on a single-type graph of PrimeKG's gene/protein size:
    steps/s of the eager loop (device sampler + ordinary step, fp32), and whether the GPU or the host bounds it -- the
    host's time to ENQUEUE a step (no synchronisation inside the loop) against the step's wall time: a launch-bound loop has
    the two equal (the GPU waits for the host), a GPU-bound one has the host run ahead.  The kernel-trace busy fraction of
    the same loop: scripts/busy_fraction.py over a rocprofv3 --kernel-trace of scripts/dev_gcl_sampled.py (profiles/)."""
    from biomedkg_amd import gcl_module as GM, ops, synthetic
    from biomedkg_amd.gcl_module import GGDModule
    from biomedkg_amd.samplers import NeighborBatchLoader
    prec = ops.get_matmul_precision()
    ops.set_matmul_precision("fp32")
    try:
        ei = synthetic.gcl_small(n, e, seed=1).to(dev)
        x = synthetic.lm_features(n, 2, 768, seed=2).to(dev)
        torch.manual_seed(0)
        m = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="attention").to(dev).train()
        params = list(m.model.parameters())
        opt = torch.optim.Adam(params, lr=1e-3, fused=True)

        def loop(k, m=m, opt=opt):
            it = iter(NeighborBatchLoader(x, ei, None, seeds, (30, 30, 30), True, seed=3))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            nodes = edges = 0
            for _ in range(k):
                b = next(it)
                opt.zero_grad(set_to_none=True)
                loss = m.training_step(b)
                loss.backward()
                clip_adam_step(opt, 1.0)
                nodes += b.x.size(0); edges += b.edge_index.size(1)
            t_host = time.perf_counter() - t0              # everything enqueued (the sampler's two size reads included)
            torch.cuda.synchronize()
            return t_host / k, (time.perf_counter() - t0) / k, nodes / k, edges / k
        loop(5)
        t_host, t_wall, nodes, edges = loop(steps)
        out = {"sampler": f"NeighborLoader semantics on the device, {seeds} seeds x fan-out 30/30/30, GGD (GCN 768->256, "
                          f"attention fuser), fp32, graph N={n} E={e}", "mean_nodes": round(nodes, 1), "mean_edges": round(edges, 1),
               "eager_steps_per_s": round(1.0 / t_wall, 1), "ms_per_step": t_wall * 1e3, "host_enqueue_ms_per_step": t_host * 1e3,
               "host_share_of_wall": t_host / t_wall,
               "bound": "gpu" if t_host < 0.8 * t_wall else "host (launch-bound)"}
        # the same regime as padded, sync-free batches replayed from one captured graph per (edge bucket, augmentation
        # branch) -- biomedkg_amd/sampled_gcl.py; the first batches of each kind are captured before the timed loop
        from biomedkg_amd.sampled_gcl import NeighborGraphedGcl

        def graphed(precision, cls=GGDModule, steps=steps, warm=3):
            ops.set_matmul_precision(precision)
            torch.manual_seed(0)
            mg = cls(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="attention").to(dev).train()
            lp = NeighborGraphedGcl(mg, x, ei, seeds, (30, 30, 30), clip=1.0, seed=3)
            for _ in range(warm * steps):                   # capture what the regime uses
                lp.step()
            torch.cuda.synchronize()
            before = dict(lp.stats)
            t0 = time.perf_counter()
            for _ in range(steps):
                lp.step()
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / steps
            stats = {k: lp.stats[k] - before[k] for k in lp.stats}
            lp.close()
            return {"steps_per_s": round(1.0 / t, 1), "ms_per_step": t * 1e3, "host_enqueue_ms_per_step": t_host / steps * 1e3,
                    "stats": stats, "graphs_held": len(lp.execs)}
        out["graphed"] = graphed("fp32")
        out["graphed_bf16"] = graphed("bf16")
        out["speedup"] = round(out["graphed"]["steps_per_s"] / out["eager_steps_per_s"], 2)
        # the other two GCL models of train_gcl.py:15-38 over the same batches (round 4: masked readout / JSD for DGI, masked
        # NT-Xent for GRACE -- on ~25k-node batches GRACE's step is its 51k x 51k NT-Xent, whatever launches it)
        for name, cls, k in (("dgi", GM.DGIModule, 20), ("grace", GM.GRACEModule, 8)):
            try:
                ops.set_matmul_precision("fp32")
                torch.manual_seed(0)
                me = cls(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="attention").to(dev).train()
                oe = torch.optim.Adam(list(me.model.parameters()), lr=1e-3, fused=True)
                loop(2, me, oe)
                _, t_wall_e, _, _ = loop(k, me, oe)
                del me, oe
                warm = 6 if name == "grace" else 2       # (GRACE keys its graphs by a node rung too: more kinds to capture first)
                out[name] = {"eager_steps_per_s": round(1.0 / t_wall_e, 1), "graphed": graphed("fp32", cls, k, warm),
                             "graphed_bf16": graphed("bf16", cls, k, warm)}
                out[name]["speedup"] = round(out[name]["graphed"]["steps_per_s"] / out[name]["eager_steps_per_s"], 2)
            except Exception as why:      # noqa: BLE001  (a side figure)
                torch.cuda.synchronize()
                out[name] = {"error": f"{type(why).__name__}: {why}"[:300]}
        return out
    finally:
        ops.set_matmul_precision(prec)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_full_step(nodes, edges, relations, hidden, neg_ratio, threads, steps=1):
    """One GGD + one R-GCN/DistMult TRAINING step (fwd + bwd + clip + Adam) of the CPU oracle at the given size with
    ``threads`` threads, through ``oracle/lean.py`` -- the oracle's arithmetic with its ``[E, d]`` / per-relation ``[N, d]``
    temporaries recomputed in the backward instead of stored (the plain form needs > 62 GB at the bench size; proven equal
    on the 5,000-node graph: tests/test_oracle_lean.py).  -> seconds per step (each step timed, the fastest reported:
    the first one also pays the allocator's first touch of ~10 GB)."""
    from biomedkg_amd import synthetic
    from oracle import lean, ref_modules as R
    torch.set_num_threads(threads)
    g = synthetic.primekg_like(nodes, edges, relations, seed=42)
    torch.manual_seed(0)
    xk, xg = torch.randn(nodes, hidden), torch.randn(nodes, 768)
    kge = R.make_kge_model("rgcn", "dismult", hidden, hidden, hidden, 2, g["num_relations"])
    ggd = R.GGD(R.GCNEncoder(768, hidden, hidden, 2), hidden, 1, 0.5)
    ok, og = torch.optim.Adam(kge.parameters(), lr=1e-3), torch.optim.Adam(ggd.parameters(), lr=1e-3)
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        og.zero_grad(); lean.ggd_loss(ggd, xg, g["edge_index"]).backward()
        torch.nn.utils.clip_grad_norm_(ggd.parameters(), 1.0); og.step()
        t1 = time.perf_counter()
        ok.zero_grad(); lean.kge_step(kge, xk, g["edge_index"], g["edge_type"], neg_ratio)[0].backward()
        torch.nn.utils.clip_grad_norm_(kge.parameters(), 1.0); ok.step()
        times.append((time.perf_counter() - t0, t1 - t0))
    return min(times)


def cpu_baseline(args):
    """CPU oracle (kind "port": the torch restatement of the reference's PyG/PyGCL path -- PyG itself cannot be installed
    here) timed on this box's host cores.

    ``value`` is LIVE and BOUNDED: full training steps (GGD + R-GCN/DistMult, fwd + bwd + clip + Adam) of the memory-lean
    oracle (``cpu_full_step`` / oracle/lean.py) on a 1/``--cpu-scale`` graph of the bench graph's shape with all cores --
    one untimed step, then timed steps for ~15 s.  ``full_size``: the same step on the FULL bench graph, which takes 100-110 s
    of all 64 cores per step (180 s on one thread) and is therefore measured once per round on a GPU box of this pool by
    ``scripts/cpu_baseline_full.py`` and read from profiles/*cpu_baseline_full.json (the default run was two
    thirds CPU leg).  ``BKG_BENCH_CPU_FULL=1`` runs the full-size step live as well (``full_size_live``)."""
    from biomedkg_amd import synthetic
    from oracle import lean, ref_modules as R
    # the cores this process may actually run on (a cgroup-limited box reports more in cpu_count())
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64)
    scale = max(args.cpu_scale, 1)
    torch.set_num_threads(cores)
    n, e = args.nodes // scale, args.edges // scale
    g = synthetic.primekg_like(n, e, args.relations, seed=7)
    torch.manual_seed(0)
    xk, xg = torch.randn(n, args.hidden), torch.randn(n, 768)
    kge = R.make_kge_model("rgcn", "dismult", args.hidden, args.hidden, args.hidden, 2, g["num_relations"])
    ggd = R.GGD(R.GCNEncoder(768, args.hidden, args.hidden, 2), args.hidden, 1, 0.5)
    ok, og = torch.optim.Adam(kge.parameters(), lr=1e-3), torch.optim.Adam(ggd.parameters(), lr=1e-3)

    def step():
        og.zero_grad(); lean.ggd_loss(ggd, xg, g["edge_index"]).backward()
        torch.nn.utils.clip_grad_norm_(ggd.parameters(), 1.0); og.step()
        ok.zero_grad(); lean.kge_step(kge, xk, g["edge_index"], g["edge_type"], args.neg_ratio)[0].backward()
        torch.nn.utils.clip_grad_norm_(kge.parameters(), 1.0); ok.step()

    budget = float(os.environ.get("BKG_BENCH_CPU_BUDGET_S", "15"))
    t0 = time.perf_counter()
    step()                                   # untimed: the allocator's first touch
    warm = time.perf_counter() - t0
    times = []
    while not times or (sum(times) + times[-1] < budget and len(times) < 10):
        t0 = time.perf_counter(); step(); times.append(time.perf_counter() - t0)
    dt = min(times)
    out = {"value": e / dt, "unit": "edges/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
           "seconds_per_step": dt, "steps_timed": len(times), "seconds_first_step": warm,
           "sample": f"live: CPU oracle (oracle/lean.py), full GGD + R-GCN/DistMult training steps (fwd+bwd+clip+Adam) on a "
                     f"1/{scale}-scale graph of the bench shape (N={n}, E={e}, R={g['num_relations']}, "
                     f"neg_ratio={args.neg_ratio}), {cores} threads, fastest of {len(times)} steps = {dt:.2f} s; full_size = the "
                     f"same step on the whole bench graph, measured per round on this pool's box (profiles/)"}
    # the full-size step (all cores and ONE thread): measured by scripts/cpu_baseline_full.py
    try:
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*cpu_baseline_full.json")), reverse=True):
            off = json.load(open(path))
            out["full_size"] = {"value": off.get("edges_per_s_all_cores"), "seconds_per_step": off.get("seconds_all_cores"),
                                "cores": off.get("cores"), "nodes": off.get("nodes"), "edges": off.get("edges"),
                                "cpu_model": off.get("cpu_model"), "source": os.path.basename(path)}
            out["threads1"] = {"value": off.get("edges_per_s_1_thread"), "seconds_per_step": off.get("seconds_1_thread"),
                               "cores": 1, "thread_scaling": off.get("thread_scaling"), "source": os.path.basename(path)}
            break
    except Exception:  # noqa: BLE001
        pass
    if os.environ.get("BKG_BENCH_CPU_FULL") == "1":
        try:
            sec, sec_ggd = cpu_full_step(args.nodes, args.edges, args.relations, args.hidden, args.neg_ratio, cores)
            out["full_size_live"] = {"value": args.edges / sec, "seconds_per_step": sec, "seconds_ggd": sec_ggd,
                                     "seconds_kge": sec - sec_ggd, "cores": cores}
        except Exception as why:             # noqa: BLE001  (e.g. not enough host memory)
            out["full_size_live"] = {"error": f"{type(why).__name__}: {why}"[:200]}
    return out


# ----------------------------------------------------------------------------------------------- main
STEPS_MARK = "bench.py: structures built, steps start"      # rank 0 writes this to stderr right before its warm-up steps


def _run_ranks(args, extra_env, limit_s, setup_limit_s=None, argv=()):
    """Start N fresh interpreters on this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set and wait for them.
    -> (rank 0's stdout, exit codes, text for stderr).  ``argv``: arguments appended to this process's own (a later
    ``--mode`` / ``--detail`` wins: the two legs of the default N-GPU run).  ``limit_s``: end every rank when the STEPS (warm-up, timed loop,
    side measurements -- everything after rank 0's ``STEPS_MARK``) take longer; ``setup_limit_s``: the same for what comes
    before the mark (the first ``import torch`` on a fresh box pages the image in for 1-2 minutes, the rendezvous, the
    partition's structures), so that a short limit on the steps never ends a healthy run that was merely slow to start."""
    import socket
    import subprocess
    import tempfile
    n = args.gpus
    with socket.socket() as s:                      # a free rendezvous port on the loop-back interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BKG_BENCH_SPAWNED="1", **extra_env)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this stack
        # every rank's stderr goes to a file of its own: when one fails, ITS last lines are shown under its rank number
        # (eight interleaved tracebacks of ranks that were merely killed in a collective say nothing)
        logs.append(tempfile.TemporaryFile(mode="w+b"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=logs[-1]))
    out, first_dead, note, t_mark, killed = None, None, "", None, set()
    t0 = time.time()
    while out is None:                              # a rank that dies early leaves the others waiting in a collective:
        try:                                        # end them instead of waiting for the caller's own limit
            out, _ = procs[0].communicate(timeout=2)
        except subprocess.TimeoutExpired:
            dead = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            if t_mark is None and (limit_s is not None or setup_limit_s is not None):
                logs[0].seek(0)
                if STEPS_MARK.encode() in logs[0].read():
                    t_mark = time.time()
                logs[0].seek(0, 2)
            late = ((t_mark is not None and limit_s is not None and time.time() - t_mark > limit_s)
                    or (t_mark is None and setup_limit_s is not None and time.time() - t0 > setup_limit_s))
            if dead or late:
                if dead:
                    first_dead = dead[0]
                    note += f"bench.py --gpus {n}: rank {dead[0][0]} exited with {dead[0][1]}; ending the others\n"
                elif t_mark is not None:
                    note += f"bench.py --gpus {n}: no result {limit_s:.0f} s after the steps started; ending the ranks\n"
                else:
                    note += f"bench.py --gpus {n}: the ranks were not ready after {setup_limit_s:.0f} s; ending them\n"
                if dead:                            # a moment for ranks that are failing the same way to say so themselves
                    t_grace = time.time() + 3.0
                    while time.time() < t_grace and any(p.poll() is None for p in procs):
                        time.sleep(0.1)
                for r, p in enumerate(procs):
                    if p.poll() is None:
                        p.kill()
                        killed.add(r)
                out, _ = procs[0].communicate()
    codes = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            codes.append(p.wait(timeout=max(deadline - time.time(), 1)))
        except subprocess.TimeoutExpired:           # rank 0 is gone and this one is stuck in a collective
            p.kill()
            codes.append(p.wait() or 1)
    bad = [c for c in codes if c != 0]

    def tail(r, lines=40):
        logs[r].seek(0)
        text = logs[r].read().decode(errors="replace").splitlines()
        return "\n".join(text[-lines:])
    if bad:
        # the rank that failed FIRST (the others' errors are the collective timing out or the kill), else every bad one
        show = [r for r, c in enumerate(codes) if c != 0 and r not in killed] or ([first_dead[0]] if first_dead else [])
        for r in show:
            note += f"---- bench.py rank {r} (exit code {codes[r]}), last lines of its stderr ----\n{tail(r)}\n"
    else:
        note += tail(0, lines=200) + ("\n" if tail(0, 1) else "")     # rank 0's warnings (RCCL's banner etc.)
    return out.decode(), codes, note


def graphed_line_refused(stdout):
    """Why the graphed attempt's line must not stand, or None: it has to be there, parse strictly, say that replays
    happened, and show that they TRAINED -- both fixed-draw check losses finite and lower after the run than before."""
    import math
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    if not lines:
        return "no line"
    try:
        line = json.loads(lines[-1])
    except ValueError as why:
        return f"unparsable line: {why}"
    gs = (line.get("exchange") or {}).get("graphed_sub_steps") or {}
    if not gs.get("replayed") or gs.get("refused"):
        return f"no replays ({gs})"
    before, after = line.get("check_loss") or {}, line.get("check_loss_after") or {}
    for key in ("ggd", "kge"):
        a, b = before.get(key), after.get(key)
        if not (isinstance(a, (int, float)) and isinstance(b, (int, float)) and math.isfinite(a) and math.isfinite(b)):
            return f"check loss {key} missing or not finite ({a} -> {b})"
        if not b < a:
            return f"check loss {key} did not fall over the run ({a} -> {b})"
    return None


def _last_line(stdout):
    """The last JSON line of a leg's stdout as a dict, or None."""
    for l in reversed((stdout or "").splitlines()):
        if l.startswith("{"):
            try:
                return json.loads(l)
            except ValueError:
                return None
    return None


def run_partition_ladder(args, argv=(), deadline=None):
    """The node-partitioned step over N fresh ranks -> (rank 0's stdout, exit codes, text for stderr, attempts).
    ``deadline`` (time.time() value): the whole run's budget -- an attempt gets what is left of it, or is not started.

    Over RCCL the GRAPHED sub-steps go first (BKG_BENCH_PARTITION_GRAPH=1: every sub-step one hipGraph replay -- an eager
    partitioned step costs the host 3.6-5 ms to enqueue, more than a rank's kernels take from four ranks on).  Captured
    collectives have only ever run on a one-rank RCCL group here (no multi-GPU box), so that attempt is fenced three ways
    (ADVICE r04): (1) its steps get BKG_BENCH_GRAPH_ATTEMPT_S (default 75 s, counted from rank 0's "steps start" mark, not
    from process start) -- a run stuck in a collective is ended; (2) a rank whose capture fails takes every rank out with it
    (``capture_agreed``), never a mix of replaying and eager ranks; (3) the line is ACCEPTED only if it shows that the
    replays trained: both fixed-draw check losses finite and lower after the run than before it (a graph that replays
    wrongly -- stale buffers, a collective out of order -- does not pass that).  Otherwise the ranks are started again with
    the eager step, which is fenced as well (BKG_BENCH_EAGER_ATTEMPT_S, default 300 s from the same mark).

    Whole-table exchanges of unequal shards: the library's default is the packed ``all_to_all_single`` (the form the gloo
    worlds 2 / 3 / 4 / 8 of tests/test_partition_gloo.py run).  The in-place list form (``BKG_DIRECT_ALL_TO_ALL=1``: no
    (world - 1)-fold packed copy, no unpack pass) has only ever run in the simulated rank, so bench.py asks for it in its
    first two attempts -- both fenced -- and the last attempt is the packed form.  ``exchange.form`` / ``attempts`` on the
    line say which ran."""
    n = args.gpus
    backend = os.environ.get("BKG_BENCH_BACKEND", "nccl")
    setup = float(os.environ.get("BKG_BENCH_SETUP_LIMIT_S", "600"))
    eager_limit = float(os.environ.get("BKG_BENCH_EAGER_ATTEMPT_S", "300"))
    direct = ({} if "BKG_DIRECT_ALL_TO_ALL" in os.environ else {"BKG_DIRECT_ALL_TO_ALL": "1"}) if backend == "nccl" else {}
    try_graphs = "BKG_BENCH_PARTITION_GRAPH" not in os.environ and backend == "nccl"
    attempts, notes = [], ""

    def attempt(form, env, limit):
        """One fenced attempt inside what is left of the whole run's budget (``deadline``): -> (stdout, codes, note) or None
        when there is no time left for it (recorded in ``attempts``)."""
        left = (deadline - time.time()) if deadline is not None else float("inf")
        if left < 150:
            attempts.append({"form": form, "ok": False, "why_not": f"not started: {left:.0f} s of the run's budget left"})
            return None
        return _run_ranks(args, env, min(limit, left - 60), min(setup, left - 90), argv)

    out, codes = "", [1]
    if try_graphs:
        form = "graphed sub-steps" + (", in-place all-to-all" if direct else "")
        got = attempt(form, dict(direct, BKG_BENCH_PARTITION_GRAPH="1"), float(os.environ.get("BKG_BENCH_GRAPH_ATTEMPT_S", "75")))
        if got is not None:
            out, codes, note = got
            why = graphed_line_refused(out) if all(c == 0 for c in codes) else f"exit codes {codes}"
            attempts.append({"form": form, "ok": why is None, "why_not": why})
            if why is None:
                return out, codes, note, attempts
            notes += note + f"bench.py --gpus {n}: the graphed attempt is not used ({why}); running the eager step\n"
    eager_env = {"BKG_BENCH_PARTITION_GRAPH": "0"} if try_graphs else {}
    forms = [("eager step" + (", in-place all-to-all" if direct else ""), dict(direct, **eager_env))]
    if direct:
        forms.append(("eager step, packed all_to_all_single", dict(eager_env, BKG_DIRECT_ALL_TO_ALL="0")))
    for form, env in forms:
        got = attempt(form, env, eager_limit)
        if got is None:
            break
        out, codes, note = got
        ok = all(c == 0 for c in codes) and _last_line(out) is not None
        attempts.append({"form": form, "ok": ok, "why_not": None if ok else f"exit codes {codes}"})
        notes += note
        if ok:
            break
        notes += f"bench.py --gpus {n}: {form} failed (exit codes {codes})\n"
    return out, codes, notes, attempts


_LEG_KEYS = ("mode", "value", "unit", "ms_per_step", "ms_per_step_median", "scaling", "ranks", "backend", "steps", "warmup",
             "ms_gcl_step", "ms_kge_step", "triplets_per_sec_kge_step", "edges_per_sec_gcl_step")


def merge_legs(primary, secondary, key, attempts=None):
    """The ONE line of the default N-GPU run: ``primary`` (a leg's parsed line) with the other leg's headline figures under
    ``key`` -- both regimes explicitly, each with its own ``scaling``.  -> strict JSON text < LINE_LIMIT."""
    line = dict(primary)
    if isinstance(secondary, dict) and "error" not in secondary:
        leg = {k: secondary.get(k) for k in _LEG_KEYS if secondary.get(k) is not None}
        leg["parallelism"] = (secondary.get("config") or {}).get("parallelism")
        ex = secondary.get("exchange") or {}
        leg["exchange"] = {k: ex[k] for k in ("form", "exchanges_per_step", "wire_bytes_sent_per_step", "bucket_bytes_ggd",
                                              "bucket_bytes_kge", "allreduce_alone_ms_ggd", "allreduce_alone_ms_kge",
                                              "exposed_fraction", "host_enqueue_ms_per_step") if k in ex}
        leg["detail"] = secondary.get("detail")
    else:
        leg = secondary
    line[key] = leg
    if attempts is not None:
        line["attempts"] = attempts
    line = _strict(line)
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    for drop in ("attempts", "check_loss_after", "check_loss", "partition_bound"):
        if len(text) < LINE_LIMIT - 512:
            break
        line.pop(drop, None)
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_LIMIT and isinstance(line.get(key), dict):
        line[key].pop("exchange", None)
        text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    return text


def launch_ranks(args):
    """``python3 bench.py --gpus N`` with no launcher around it: this process becomes the launcher -- it never touches
    the GPU (nothing before this point does: ``import torch`` and ``torch.cuda.device_count()`` do not initialise HIP on
    this image) -- and starts N fresh interpreters on this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the
    way Lightning starts the reference's ranks from one ``python train_gcl.py`` (reference train_gcl.py:63-68,
    configs/gcl.yaml:11).  Rank 0's stdout (the single JSON line) is relayed; the exit code is the worst child's.

    ``--mode auto`` over N > 1 GPUs runs BOTH regimes, one after the other, each on N fresh ranks, and prints ONE line
    (rounds 1-4 reported the node partition, round 5 the replicas; the two figures are not comparable and the
    line must say which is which): the headline is the NODE-PARTITIONED step (BASELINE configs[3] / north_star: one graph,
    ``"scaling": "strong"``, ``run_partition_ladder``) and ``replicas`` carries the reference's own DDP regime (one graph
    per rank, flat gradient all-reduce, ``"scaling": "weak"``) with its own value / ms_per_step.  If every partition attempt
    fails the replicas leg becomes the headline and ``partition`` holds the attempts' outcomes -- a number the driver can
    read either way.  ``--mode partition`` / ``--mode replicas`` run one leg."""
    n = args.gpus
    one_device = os.environ.get("BKG_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count()
    if have < n and not one_device:
        raise SystemExit(f"bench.py --gpus {n}: this node shows {have} GPU(s) (BKG_BENCH_ONE_DEVICE=1 + "
                         "BKG_BENCH_BACKEND=gloo is the functional single-device check)")
    setup = float(os.environ.get("BKG_BENCH_SETUP_LIMIT_S", "600"))

    def finish(out, codes, note):
        sys.stderr.write(note)
        sys.stdout.write(out)
        sys.stdout.flush()
        bad = [c for c in codes if c != 0]
        raise SystemExit(max(abs(c) for c in bad) if bad else 0)

    # the whole run's budget (the driver's own limit on a bench run is 30 minutes): every attempt is cut to what is left
    deadline = time.time() + float(os.environ.get("BKG_BENCH_TOTAL_S", "1400"))
    if args.mode == "partition":
        out, codes, note, _ = run_partition_ladder(args, deadline=deadline)
        finish(out, codes, note)
    replicas_limit = float(os.environ.get("BKG_BENCH_REPLICAS_LIMIT_S", "400"))
    if args.mode == "replicas":
        finish(*_run_ranks(args, {}, replicas_limit, setup))
    # auto: both legs, one line
    base, ext = os.path.splitext(args.detail)
    out_r, codes_r, note_r = _run_ranks(args, {}, replicas_limit, setup,
                                        ["--mode", "replicas", "--detail", base + "_replicas" + ext])
    line_r = _last_line(out_r) if all(c == 0 for c in codes_r) else None
    sys.stderr.write(note_r)
    if line_r is None:
        sys.stderr.write(f"bench.py --gpus {n}: the replicas leg gave no line (exit codes {codes_r})\n")
    out_p, codes_p, note_p, attempts = run_partition_ladder(args, ["--mode", "partition"], deadline=deadline)
    line_p = _last_line(out_p) if all(c == 0 for c in codes_p) else None
    sys.stderr.write(note_p)
    try:      # (RCCL's banner etc. went through the children; nothing of ours is buffered in C stdio here)
        sys.stderr.flush()
    except Exception:      # noqa: BLE001
        pass
    if line_p is not None:
        other = line_r if line_r is not None else {"error": f"the replicas leg gave no line (exit codes {codes_r})"}
        print(merge_legs(line_p, other, "replicas", attempts), flush=True)
        raise SystemExit(0)
    if line_r is not None:
        sys.stderr.write(f"bench.py --gpus {n}: no partition attempt gave a line (exit codes {codes_p}); the replicas leg is "
                         "the headline\n")
        print(merge_legs(line_r, {"error": f"no attempt of the node-partitioned step gave a line (exit codes {codes_p})",
                                  "attempts": attempts}, "partition"), flush=True)
        raise SystemExit(0)
    bad = [c for c in list(codes_r) + list(codes_p) if c != 0]
    raise SystemExit(max(abs(c) for c in bad) if bad else 1)


# what an outer launcher (torch.distributed.run) puts into its workers' environment
_OUTER_RANK_KEYS = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
                    "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT")


def take_over_from_outer_launcher(args):
    """``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` (the driver's SCALE form): N copies of this
    file arrive with RANK / WORLD_SIZE set.  Running the step in THOSE processes would mean the eager partitioned step --
    host-bound from four ranks on (3.6-5 ms to enqueue against 1.4-2.8 ms of kernels) -- with no way back if the graphed
    form failed: a rank cannot restart itself.  So the outer launcher's rank 0, which has not touched the GPU (nothing
    before this point does), becomes bench.py's OWN launcher exactly as if it had been started bare -- ``launch_ranks``:
    N fresh ranks on a rendezvous port of their own, the graphed sub-steps first, the eager step if that attempt fails
    or stalls -- and relays its rank 0's single JSON line; the outer launcher's other workers leave at once with exit
    code 0.  ``BKG_BENCH_OUTER_RANKS=1`` keeps the outer launcher's processes as the ranks (the eager step, or the
    graphed one with ``BKG_BENCH_PARTITION_GRAPH=1``)."""
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {os.environ.get('WORLD_SIZE')} rank(s)")
    if int(os.environ.get("RANK", "0")) != 0:
        raise SystemExit(0)
    for key in list(os.environ):
        if key in _OUTER_RANK_KEYS or key.startswith("TORCHELASTIC_"):
            del os.environ[key]
    launch_ranks(args)                              # does not return


def main():
    args = parse()
    # dmabuf IPC: RCCL across processes needs it on this stack (the launcher sets it for the ranks it starts; ranks started
    # by an outer torch.distributed.run get it here, before anything touches the GPU)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                          # does not return
    if (args.gpus > 1 and os.environ.get("BKG_BENCH_SPAWNED") != "1" and os.environ.get("BKG_BENCH_OUTER_RANKS") != "1"
            and args.mode in ("partition", "auto")):
        take_over_from_outer_launcher(args)         # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # BKG_BENCH_ONE_DEVICE=1 / BKG_BENCH_BACKEND=gloo: functional check of the multi-rank code path on a 1-GPU box
    # (all ranks share cuda:0, gloo instead of RCCL); never used for reported numbers.
    if os.environ.get("BKG_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # BKG_PARTITION_FORCE_COLLECTIVES=1 with --gpus 1 --mode partition | replicas: a 1-rank RCCL group, every collective of the
    # N-rank path issued for real (functional check of the RCCL calls on a single-GPU box; never a reported number)
    forced = world == 1 and os.environ.get("BKG_PARTITION_FORCE_COLLECTIVES") == "1" and args.mode in ("partition", "replicas")
    backend = None
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        backend = os.environ.get("BKG_BENCH_BACKEND", "nccl")
        if os.environ.get("BKG_BENCH_PARTITION_GRAPH") == "1":
            # captured collectives: no pooled events (an event of the pool may be handed from a finished eager collective to
            # a captured one while the watchdog still looks at it)
            os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")
        kw = {"device_id": dev} if backend == "nccl" else {}
        if forced:
            kw.update(rank=0, world_size=1)
        # a rank that never arrives must not hold the others until the caller's limit: the rendezvous and every collective
        # give up after BKG_BENCH_TIMEOUT_S (default 5 minutes) with an error naming the call
        import datetime
        kw["timeout"] = datetime.timedelta(seconds=float(os.environ.get("BKG_BENCH_TIMEOUT_S", "300")))
        torch.distributed.init_process_group(backend, **kw)
        if os.environ.get("BKG_BENCH_FAIL_RANK") == str(rank):      # fault injection (tests): a rank that dies after rendezvous
            os._exit(7)

    from biomedkg_amd import ops
    mode = args.mode if args.mode != "auto" else ("replicas" if world > 1 else "single")
    ops.set_matmul_precision(args.precision)
    if mode == "partition":
        step_g, step_k, n, e, r = build_partitioned_steps(args, rank, world, dev)
    else:
        step_g, step_k, n, e, r = build_replica_steps(args, rank, world, dev)
    h = args.hidden

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # everything the timed loop needs is made BEFORE the warm-up steps -- the per-step marks, the tagged calls' events, the
    # cycle collection -- so that nothing but the bracket's synchronisation stands between the last warm-up step and the
    # first timed one
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step GPU times (no sync inside)
    for ev in marks:                                    # (created for real here, not inside the timed loop)
        ev.record()
    ops.reserve_events(64 * args.steps + 64)            # ... and the tagged calls' events (~50 per step)
    import gc
    gc.collect()                                        # the interpreter's cycle collector stays out of the timed loop (a full
    gc.disable()                                        # collection over the warm-up's tensors showed up as one 21 ms step)
    # ... and out of the gap between warm-up and timed loop: a collection THERE hands the warm-up's cyclic garbage (buffers
    # of the last steps) back to the caching allocator, the first timed step then finds another block layout and runs
    # 1.0-1.3 ms long (10.9-11.2 against 9.75 ms, every run; scripts/dev_first_step_syncs.py takes the loop apart).  So:
    # collect what the set-up left, switch the collector off, THEN warm up -- the timed steps see the allocator the warm-up
    # steps left
    if rank == 0 and world > 1:
        sys.stderr.write(STEPS_MARK + "\n"); sys.stderr.flush()
    for _ in range(args.warmup):
        step_g(); step_k()
    step_k.finish()
    sync()
    ops.KERNEL_TIMERS = None if os.environ.get("BKG_BENCH_NO_KERNEL_TIMERS") == "1" else {}
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step_g(); step_k()
        marks[i + 1].record()
    step_k.finish()                                     # the last KGE update belongs to the timed region
    sync()
    dt = time.perf_counter() - t0
    gc.enable()
    per_step_raw = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
    per_step = sorted(per_step_raw)
    timers, ops.KERNEL_TIMERS = ops.KERNEL_TIMERS or {}, None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3

    # separately timed sub-steps (not part of the headline; one extra pass each)
    def timed(fn, reps=3):
        sync(); t = time.perf_counter()
        for _ in range(reps):
            fn()
        getattr(fn, "finish", lambda: None)()
        sync()
        return (time.perf_counter() - t) / reps

    ms_g, ms_k = timed(step_g) * 1e3, timed(step_k) * 1e3

    exchange = None
    if mode == "partition":
        # what crosses the wire per step and how long the compute stream waits for it (events around every wait); then
        # the same steps with every exchange waited for at once (BKG_PARTITION_SERIAL semantics): all of the collective
        # time shows as exposed -- the ratio is how much of the exchange the schedule hides
        part = step_g.part
        comm = part.comm
        reps = 3

        graphed = getattr(step_g, "graph_stats", None) is not None
        host_graphed = None
        if graphed:                      # the host's time per step in the mode the timed loop ran in (replays)
            sync(); t = time.perf_counter()
            for _ in range(reps):
                step_g(); step_k()
            host_graphed = (time.perf_counter() - t) / reps * 1e3
            sync()
        step_g.force_eager = step_k.force_eager = True      # the wire counters and waits below are those of the eager step

        def measure(serial):
            comm.serial, comm.timed = serial, True
            comm.take_stats()
            sync(); t = time.perf_counter()
            for _ in range(reps):
                step_g(); step_k()
            host = (time.perf_counter() - t) / reps * 1e3       # everything ENQUEUED (no synchronisation inside a step)
            sync()
            wall = (time.perf_counter() - t) / reps * 1e3
            rx, tx, count, exposed = comm.take_stats()
            comm.serial, comm.timed = False, False
            return wall, rx / reps, tx / reps, count / reps, exposed / reps, host
        wall_o, rx, tx, count, exposed_o, host_o = measure(False)
        wall_s, _, _, _, exposed_s, _ = measure(True)
        launches = [0]
        from biomedkg_amd import _lib as _L
        real_call = _L.call

        def counting(name, *a):
            launches[0] += 1
            return real_call(name, *a)
        _L.call = ops.L.call = counting
        try:
            step_g(); step_k()
        finally:
            _L.call = ops.L.call = real_call
        sync()
        step_g.force_eager = step_k.force_eager = False
        exchange = {"wire_bytes_received_per_step": rx, "wire_bytes_sent_per_step": tx, "exchanges_per_step": count,
                    "exchange_exposed_ms": exposed_o, "exchange_serial_ms": exposed_s,
                    "exposed_fraction": (exposed_o / exposed_s) if exposed_s else None,
                    "ms_per_step_overlapped": wall_o, "ms_per_step_serial": wall_s,
                    "host_enqueue_ms_per_step": host_o, "library_calls_per_step": launches[0],
                    "graphed_sub_steps": getattr(step_g, "graph_stats", None),
                    "form": "graphed sub-steps (hipGraph replays)" if graphed else "eager step",
                    "host_enqueue_ms_per_step_graphed": host_graphed,
                    "host_note": "host_enqueue_ms_per_step: the host's time to enqueue one step (no synchronisation inside it); a "
                                 "rank whose share of the kernels is shorter than that is launch-bound, whatever the wire does",
                    "halo": {k: {"used": comm.patterns.get(k) is not None, "max_share_of_remote_rows_read": v}
                             for k, v in part.halo_fraction.items()},
                    "rows": {"n_local": part.n_local, "n_max": part.n_max, "bounds": part.bounds},
                    "padded_over_real_rows": part.padding_waste,
                    "note": "rank 0's counts; gathers only: forward tables and the backward's gated output gradients (bf16 in "
                            "bf16 mode), the scorer's embedding table -- its gradient needs no exchange (both-owner scoring) -- "
                            "the first R-GCN layer's constant input gathered once per graph, not per step; the flat gradient "
                            "all-reduce is not counted here"}
    if mode == "replicas" and getattr(step_g, "bucket", None) is not None and step_g.bucket.live:
        # what crosses the wire per step in the replicas regime: the two flat gradient buckets, all-reduced (ring: 2 (P - 1) / P
        # of the bucket per rank); timed alone on the buckets as they stand after the run (3 repetitions, every rank together)
        def reduce_ms(bucket):
            bucket.pack()
            sync(); t = time.perf_counter()
            for _ in range(3):
                bucket.all_reduce_packed()
            sync()
            return (time.perf_counter() - t) / 3 * 1e3
        by_g = sum(p.numel() for p in step_g.bucket.params) * 4
        by_k = sum(p.numel() for p in step_k.bucket.params) * 4
        ring = 2.0 * (world - 1) / max(world, 1)
        exchange = {"form": "flat gradient all-reduce per sub-step (the KGE bucket under the next GGD sub-step)",
                    "exchanges_per_step": 2, "bucket_bytes_ggd": by_g, "bucket_bytes_kge": by_k,
                    "wire_bytes_sent_per_step": ring * (by_g + by_k), "wire_bytes_received_per_step": ring * (by_g + by_k),
                    "allreduce_alone_ms_ggd": reduce_ms(step_g.bucket), "allreduce_alone_ms_kge": reduce_ms(step_k.bucket)}
    partition_bound = None
    if world > 1 and mode == "replicas":
        # why this is not the node-partitioned step: its measured bound at this world size (profiles/*partition_model.json:
        # each rank's compute measured on one GPU by scripts/simulate_ranks.py + the wire model)
        try:
            import glob
            path = sorted(glob.glob(os.path.join(ROOT, "profiles", "*partition_model.json")), reverse=True)[0]
            row = next(r_ for r_ in json.load(open(path))["ranks"] if r_["world"] == world)
            partition_bound = {"world": world, "compute_ms_slowest_rank_measured": row.get("compute_ms_measured"),
                               "step_ms_as_coded": row.get("step_ms_as_coded"), "speedup_over_one_gpu": row.get("speedup_as_coded"),
                               "wire_MB_per_step_per_rank": row.get("wire_MB"), "source": os.path.basename(path)}
        except Exception:      # noqa: BLE001 - reporting only
            partition_bound = None
    # the same two losses after the run's optimizer steps (every rank: collectives inside): training moved them
    check_after = step_g.check_fn() if mode == "partition" else None
    n_neg = args.neg_ratio * e
    strong = mode == "partition"
    total_edges = e if strong or world == 1 else world * e
    roof, others = rooflines(timers, args.precision)

    ms_det = None
    if world == 1 and not args.no_extras:
        ops.set_deterministic(True)                     # Trainer(deterministic=True): fixed-order weight-gradient sums
        for _ in range(2):
            step_g(); step_k()
        sync(); t1 = time.perf_counter()
        for _ in range(5):
            step_g(); step_k()
        sync()
        ms_det = (time.perf_counter() - t1) / 5 * 1e3
        ops.set_deterministic(False)

    ms_fresh = None
    if world == 1 and mode == "single" and not args.no_extras:
        # the reference's loaders hand over a NEW edge_index every batch: nothing cached on the tensor survives, so the
        # CSRs, (relation, target) segments, scorer incidence lists and the positive hash set are rebuilt inside the step
        def fresh():
            ei = step_k.batch.edge_index.clone()
            step_k.batch.edge_index = step_g.batch.edge_index = ei
            step_k.batch.edge_type = step_k.batch.edge_type.clone()
            step_g(); step_k()
        try:
            fresh(); fresh()
            ms_fresh = timed(fresh, reps=5) * 1e3
        except Exception:                      # noqa: BLE001  (a side figure: never costs the headline line)
            torch.cuda.synchronize()

    ms_linear_head = ms_literal_last = ms_fresh_means = ms_all_literal = None
    if world == 1 and not args.no_extras and args.precision == "bf16":
        def timed5():           # (GGD's augmentation coin makes the steps of a short run differ: as many steps as the headline)
            for _ in range(3):
                step_g(); step_k()
            sync(); t1 = time.perf_counter()
            for _ in range(max(args.steps, 10)):
                step_g(); step_k()
            sync()
            return (time.perf_counter() - t1) / max(args.steps, 10) * 1e3
        # the same step with the first R-GCN layer's segment means of the CONSTANT node features aggregated every step
        # instead of kept on the graph (ops._RgcnConv)
        ops.RGCN_KEEP_SEGMENT_MEANS = False
        try:
            ms_fresh_means = timed5()
        finally:
            ops.RGCN_KEEP_SEGMENT_MEANS = True
        # the same bf16 step with GGD's LAST CONV LAYER as a layer (its [2N, 256] product and [N, 512] aggregation, forward
        # and backward) instead of folded into the linear head behind it (model/gcl.py GGD._forward_views) ...
        was_fold = ops.GGD_FOLD_LAST
        ops.GGD_FOLD_LAST = False
        try:
            ms_literal_last = timed5()
            # ... and, on top, with the head in the reference's own form (the [2N, 256] Linear, then the row sum)
            ok_fn = ops.rowsum_linear_ok
            ops.rowsum_linear_ok = lambda *a, **k: False
            try:
                ms_linear_head = timed5()
                # ... and with EVERY switchable identity / kept constant of the bf16 mode off at once: the layer as a layer, the
                # head as a Linear + row sum, the first R-GCN layer's segment means recomputed every step (the negatives'
                # relation ids are the literal repeat(k)[randperm] by default) -- the reference's expressions literally, on the
                # same kernels (what stays is the graph's own bookkeeping: CSRs, segments, incidence lists, built once for a
                # graph that stays)
                ops.RGCN_KEEP_SEGMENT_MEANS = False
                try:
                    ms_all_literal = timed5()
                finally:
                    ops.RGCN_KEEP_SEGMENT_MEANS = True
            finally:
                ops.rowsum_linear_ok = ok_fn
        finally:
            ops.GGD_FOLD_LAST = was_fold

    ms_grouped_types = None
    if world == 1 and mode == "single" and not args.no_extras and args.precision == "bf16":
        # the same step with the negatives' relation ids GROUPED BY RELATION (BKG_GROUPED_NEG_TYPES=1, biomedkg_amd/ops.py
        # grouped_types: the same distribution of typed negatives, the default of rounds 3-5) instead of the reference's literal
        # repeat(..)[randperm(..)], which the headline draws since round 6
        from biomedkg_amd import kge_module as _km
        was = _km.GROUPED_NEG_TYPES
        _km.GROUPED_NEG_TYPES = True
        try:
            ms_grouped_types = timed5()
        finally:
            _km.GROUPED_NEG_TYPES = was

    neg10 = None
    if world == 1 and mode == "single" and not args.no_extras and args.neg_ratio != 10:
        # the reference's DEFAULT KGE workload at full size (configs/kge.yaml:14 ``neg_ratio: 10``; kge_module.py:86-102):
        # T = E + 10 E = 44 M triplets per step on the bench graph.  SURVEY 8d sizes the scorer's traffic at 45.8 GB per
        # step -> 5.7 ms at 8 TB/s, i.e. 7.7 G triplets/s as the HBM-spec bound of the whole KGE sub-step
        def kge_neg10():
            from biomedkg_amd.kge_module import KGEModule
            torch.manual_seed(1234)
            m = KGEModule("rgcn", "dismult", h, h, h, 2, r, 2, "cosine", 1e-3, 0.2, "none", 10, "random").to(dev)
            m.train()
            step10 = make_step(m, step_k.batch, 1)
            for _ in range(2):
                step10()
            per = []
            for _ in range(5):
                sync(); t1 = time.perf_counter()
                step10()
                sync()
                per.append((time.perf_counter() - t1) * 1e3)
            ms10 = sorted(per)[len(per) // 2]
            t10 = e + 10 * e
            return {"neg_ratio": 10, "triplets": t10, "ms_kge_step": ms10, "triplets_per_sec_kge_step": t10 / (ms10 * 1e-3),
                    "hbm_spec_bound_triplets_per_sec": 7.7e9, "frac_of_hbm_spec_bound": t10 / (ms10 * 1e-3) / 7.7e9,
                    "note": "R-GCN (4 conv layers, d = 256) + DistMult training step, fwd + bwd + clip + Adam, the reference's "
                            "default neg_ratio; bound: SURVEY 8d's 45.8 GB per step at the 8 TB/s HBM spec"}
        try:
            neg10 = kge_neg10()
        except Exception as why:               # noqa: BLE001  (a side figure: never costs the headline line)
            torch.cuda.synchronize()
            neg10 = {"error": f"{type(why).__name__}: {why}"[:300]}
        torch.cuda.empty_cache()

    ms_fp32 = None
    if world == 1 and not args.no_extras and args.precision == "bf16":
        ops.set_matmul_precision("fp32")                # the exact parity mode on the same modules / graph
        for _ in range(3):
            step_g(); step_k()
        # per-step times, median: this mode's first steps allocate new buffer sizes, and a first-time hipMalloc inside
        # a three-step mean showed up as 40 ms/step on one run (28 on the others)
        per = []
        for _ in range(5):
            sync(); t1 = time.perf_counter()
            step_g(); step_k()
            sync()
            per.append((time.perf_counter() - t1) * 1e3)
        ms_fp32 = sorted(per)[len(per) // 2]
        ops.set_matmul_precision(args.precision)

    if rank == 0:
        what = {"single": "the whole graph on one GPU",
                "partition": f"ONE graph, nodes partitioned over {world} ranks (gathers of the rows a rank's edges read, "
                             "forward tables and output gradients alike; every triplet scored by the owners of both its "
                             "endpoints; no reduce-scatter anywhere)",
                "replicas": "one graph per rank, gradient all-reduce only (the reference's DDP regime)"}[mode]
        out = {
            "metric": "edges/sec, GCL (GGD) step + KGE (R-GCN+DistMult) step, PrimeKG++ scale",
            "value": total_edges / (ms * 1e-3), "unit": "edges/s", "n_gpus": world,
            "ranks": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
            "backend": ("rccl" if backend == "nccl" else backend), "mode": mode, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32",
            "dtype_note": ("bf16 storage / fp32 accumulate: matrix-core operands, gather tables, hidden activations "
                           "between conv layers (and their gradients) and intermediates that only feed products are "
                           "bf16; every accumulation, encoder outputs, scorer, losses, weights, weight gradients and "
                           "optimizer state are fp32" if args.precision == "bf16"
                           else "exact fp32 everywhere (fp32-input MFMA)"),
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] shape: synthetic PrimeKG++-like graph, "
                                   f"N={n} E={e} R={r}; GGD/GCN 768->{h} (4 conv layers) step + R-GCN d={h} "
                                   f"(4 conv layers) + DistMult, neg_ratio={args.neg_ratio}, step; "
                                   f"{args.precision} matmuls; fwd+bwd+clip+Adam; {what}",
                       "parallelism": {"single": "dp1", "partition": f"np{world}", "replicas": f"dp{world}"}[mode],
                       "cached_structures": "the step re-uses one graph: CSRs, (relation, target) segments, incidence lists, "
                                            "the positive-edge hash set and the bf16 copy of the constant input features are "
                                            "built once (the reference's per-batch loaders rebuild PyG's COO bookkeeping)",
                       "identities": "bf16 mode evaluates three expressions of the reference in an algebraically equal form, "
                                     "nothing is skipped: the corrupted view's first product x[perm] W^T as (x W^T)[perm], "
                                     "GGD's head (W h + b).sum(1) as h . W.sum(0) + b.sum(), and (round 5) -- no activation "
                                     "follows the last conv layer, so the scores are linear in its INPUT H -- that layer "
                                     "folded into the head: A_hat (H W_L^T) . v = A_hat (H w'), w' = W_L^T W_m^T 1, one column "
                                     "per view through the aggregation (ms_per_step_literal_ggd_last_layer times the layer as a "
                                     "layer; every parameter keeps its exact gradient).  Constants of a run are computed once: "
                                     "the features' bf16 copy and the first R-GCN layer's segment means of the constant node "
                                     "features (ms_per_step_fresh_first_layer_means recomputes them) (every gradient included; "
                                     "the exact-fp32 mode shares the first product too -- there the identity holds bit for "
                                     "bit, rows of a product do not depend on their position -- and keeps the reference's "
                                     "head).  The negatives' relation ids are the reference's literal "
                                     "edge_type.repeat(k)[randperm] (round 6; ms_per_step_grouped_neg_types times the opt-in form "
                                     "of rounds 3-5: the ids grouped by relation, the same distribution of typed negatives since "
                                     "the sampler's order is exchangeable).  ms_per_step_all_literal: every one of these "
                                     "switched off at once"},
            "triplets_per_sec_kge_step": (total_edges + (n_neg if strong or world == 1 else world * n_neg)) / (ms_k * 1e-3),
            "edges_per_sec_gcl_step": total_edges / (ms_g * 1e-3),
            "ms_per_step_median": per_step[len(per_step) // 2], "ms_per_step_max": per_step[-1],
            "ms_per_step_each": [round(v, 3) for v in per_step_raw],
            "ms_per_step_slowest_index": per_step_raw.index(per_step[-1]),
            "ms_gcl_step": ms_g, "ms_kge_step": ms_k, "ms_per_step_fp32": ms_fp32, "ms_per_step_deterministic": ms_det,
            "ms_per_step_literal_ggd_last_layer": ms_literal_last, "ms_per_step_fresh_first_layer_means": ms_fresh_means,
            "ms_per_step_linear_ggd_head": ms_linear_head, "ms_per_step_grouped_neg_types": ms_grouped_types,
            "ms_per_step_all_literal": ms_all_literal,
            "ms_per_step_fresh_structures": ms_fresh,
            "kge_step_neg10": neg10,
            "ms_kge_step_neg10": (neg10 or {}).get("ms_kge_step"),
            "triplets_per_sec_kge_step_neg10": (neg10 or {}).get("triplets_per_sec_kge_step"),
            "roofline": roof, "roofline_model": ROOFLINE_MODEL, "kernels": others, "exchange": exchange,
            "step_accounting": None,
            "partition_bound": partition_bound,
            "check_loss": getattr(step_g, "check_loss", None), "check_loss_after": check_after,
            "cpu_baseline": None,
        }
        if world == 1:
            live = [roof] + others if roof else []
            out["step_accounting"] = {
                "live": {"tagged_bytes_per_step": sum(r["bytes_per_launch"] * r["launches"] for r in live) / args.steps,
                         "tagged_ms_per_step": sum(r["ms_total"] for r in live) / args.steps,
                         "ms_per_step": ms,
                         "tagged_GBs_over_the_step": (sum(r["bytes_per_launch"] * r["launches"] for r in live) / args.steps
                                                      / (ms * 1e-3) / 1e9) if live else None,
                         "note": "HIP events of this run: the tagged gather / aggregate / scorer calls' algorithmic bytes and "
                                 "time per step; the remainder of the step (products, casts, sorts, optimizer, glue) is "
                                 "broken down by kernel family in `traced`"},
                "traced": step_accounting(args.precision)}

        def extra(fn, *a):      # the side figures never cost the headline line: a failure is reported in its place
            try:
                return fn(*a)
            except Exception as why:          # noqa: BLE001
                torch.cuda.synchronize()
                return {"error": f"{type(why).__name__}: {why}"[:300]}
        if world == 1 and not args.no_extras:
            out["grace_step"] = extra(grace_figures, dev)
            out["ntxent"] = extra(ntxent_figures, dev)
            out["fusion"] = extra(fusion_figures, dev, step_g.batch.edge_index if mode == "single" else None)
            if mode == "single":
                out["sampled_regime"] = extra(sampled_figures, dev, step_k.batch.edge_index, step_k.batch.edge_type, n, r)
                out["sampled_regime_gcl"] = extra(sampled_gcl_figures, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = extra(cpu_baseline, args)
    if world > 1 or forced:
        torch.distributed.destroy_process_group()
    if rank == 0:
        emit(out, args.detail)


if __name__ == "__main__":
    main()
