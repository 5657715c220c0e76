"""The line bench.py prints LAST on stdout is what the driver parses (round 4's 26 KB line came back ``parsed: null``):
it must be one compact strict-JSON object below 8 KB carrying the contract's keys, ``roofline`` and ``cpu_baseline``
(and ``exchange`` for N > 1), whatever the full record holds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _no_constants(name):
    raise AssertionError(f"non-strict JSON constant {name} in the bench line")


def _check(text, n_gpus):
    assert "\n" not in text and len(text) < 8192, len(text)
    line = json.loads(text, parse_constant=_no_constants)
    for key in CONTRACT:
        assert key in line, key
    assert line["n_gpus"] == n_gpus and isinstance(line["config"]["workload"], str) and line["config"]["parallelism"]
    return line


def _canned(n_gpus):
    """A record shaped like round 4's, with its worst habits: ~1 KB of prose per kernel, a long kernel list, NaN / inf
    values, numpy-ish scalars and a sampled-regime block."""
    import bench
    prose = "x" * 1100
    roof = {"bound": "hbm", "kernel": "spmm_items_kernel<__bf16,32,1> + spmm_fixup_kernel: " + "y" * 400, "tag": "gcn_propagate_views",
            "achieved": 12747.7, "peak": 8000.0, "unit": "GB/s", "frac": 0.7588, "traffic": 1.88e9, "bytes_per_launch": 4261640000,
            "avg_launch_ms": 0.3343, "avg_launch_ms_traced": 0.33125, "l2_hit_rate": float("nan"), "pmc_source": "r04.json",
            "launches": 36, "ms_total": 12.0, "frac_definition": prose, "guide_GBs": {"a": 1.0}}
    out = {"metric": "edges/sec, GCL (GGD) step + KGE (R-GCN+DistMult) step, PrimeKG++ scale", "value": 4.2e8, "unit": "edges/s",
           "n_gpus": n_gpus, "ranks": n_gpus, "backend": "rccl" if n_gpus > 1 else None, "steps": 20, "warmup": 5,
           "ms_per_step": 9.48, "higher_is_better": True, "scaling": "strong" if n_gpus > 1 else "weak", "vs_baseline": None,
           "dtype": "bf16", "dtype_note": prose, "data": "synthetic",
           "config": {"workload": "w" * 900, "parallelism": f"np{n_gpus}" if n_gpus > 1 else "dp1", "identities": prose,
                      "cached_structures": prose},
           "roofline": roof, "roofline_model": bench.ROOFLINE_MODEL, "kernels": [dict(roof, tag=f"k{i}") for i in range(12)],
           "step_accounting": {"traced": {"kernels": [{"kernel": "k" * 120, "ms_per_step": 0.01}] * 60}},
           "ms_per_step_each": [9.4] * 200, "ms_gcl_step": 3.5, "ms_kge_step": float("inf"), "ms_per_step_fp32": 24.5,
           "check_loss": {"ggd": 1.49, "kge": 1.73, "note": prose}, "check_loss_after": {"ggd": 1.2, "kge": 1.5},
           "cpu_baseline": {"value": 39280.7, "unit": "edges/s", "cores": 64, "kind": "port", "cpu_model": "AMD EPYC 9575F",
                            "seconds_per_step": 12.7, "sample": "s" * 900,
                            "full_size": {"value": 36058.0, "seconds_per_step": 110.9, "cores": 64, "source": "r04_cpu.json",
                                          "note": prose},
                            "threads1": {"value": 22117.5, "seconds_per_step": 180.9, "cores": 1, "sample": prose}},
           "ntxent": {"cases": [{"fwd_frac": 0.457, "note": prose}] * 4}, "fusion": {"redaf": {"note": prose}},
           "sampled_regime": {"note": prose}, "sampled_regime_gcl": {"note": prose * 3}, "exchange": None}
    if n_gpus > 1:
        out["exchange"] = {"wire_bytes_received_per_step": 6.61e8, "wire_bytes_sent_per_step": 6.6e8, "exchanges_per_step": 34.0,
                           "exchange_exposed_ms": 0.4, "exchange_serial_ms": 1.9, "exposed_fraction": 0.21,
                           "ms_per_step_overlapped": 2.2, "ms_per_step_serial": 3.1, "host_enqueue_ms_per_step": 4.1,
                           "host_enqueue_ms_per_step_graphed": 0.3, "library_calls_per_step": 350,
                           "graphed_sub_steps": {"captured": 3, "replayed": 40, "eager": 0, "refused": None},
                           "host_note": prose, "note": prose, "halo": {f"h{i}": {"used": True, "v": 0.5} for i in range(30)},
                           "rows": {"n_local": 16250, "n_max": 16300, "bounds": list(range(0, 130001, 130000 // n_gpus))},
                           "padded_over_real_rows": 1.003}
    return out


@pytest.mark.parametrize("n_gpus", [1, 8])
def test_compact_line_is_small_strict_and_complete(n_gpus, tmp_path, capsys):
    import bench
    full = _canned(n_gpus)
    assert len(json.dumps(full)) > 30_000                     # the record itself is as fat as round 4's
    line = _check(bench.compact_line(full, "bench_detail.json"), n_gpus)
    assert line["roofline"]["frac"] == pytest.approx(0.7588) and line["roofline"]["l2_hit_rate"] is None   # NaN -> null
    assert "frac_definition" not in line["roofline"] and "kernels" not in line and "step_accounting" not in line
    assert line["cpu_baseline"]["cores"] == 64 and line["cpu_baseline"]["kind"] == "port"
    assert line["cpu_baseline"]["full_size"]["seconds_per_step"] == pytest.approx(110.9)
    assert "ms_kge_step" not in line or line["ms_kge_step"] is None                 # inf -> null
    if n_gpus > 1:
        assert line["exchange"]["wire_bytes_received_per_step"] > 0 and "halo" not in line["exchange"]
        assert len(line["exchange"]["rows"]["bounds"]) == n_gpus + 1
        assert line["exchange"]["graphed_sub_steps"]["captured"] == 3
    # emit(): the full record goes to the detail file, the compact line comes LAST on stdout
    path = tmp_path / "bench_detail.json"
    bench.emit(full, str(path))
    printed = capsys.readouterr().out.splitlines()
    _check(printed[-1], n_gpus)
    detail = json.loads(path.read_text(), parse_constant=_no_constants)
    assert len(detail["kernels"]) == 12 and detail["roofline_model"]["frac_definition"]


def test_a_record_that_keeps_growing_still_fits():
    import bench
    full = _canned(8)
    full["exchange"]["rows"]["bounds"] = list(range(4000))            # a field that got out of hand
    full["check_loss"] = {f"k{i}": float(i) for i in range(600)}
    _check(bench.compact_line(full, "bench_detail.json"), 8)


@pytest.mark.gpu
def test_bench_prints_a_line_the_driver_can_read(tmp_path):
    """``python bench.py --gpus 1 --steps 2 --warmup 1`` (every extra and a short CPU leg included): the last stdout line
    parses strictly, is below 8 KB and carries roofline + cpu_baseline; the detail file holds the rest."""
    detail = tmp_path / "detail.json"
    env = dict(os.environ, BKG_BENCH_CPU_BUDGET_S="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--cpu-scale", "32", "--detail", str(detail)], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    assert sum(1 for l in lines if l.startswith("{")) == 1, [len(l) for l in lines]
    line = _check(lines[-1], 1)
    assert line["steps"] == 2 and line["warmup"] == 1 and line["value"] > 0 and line["ms_per_step"] > 0
    roof, cpu = line["roofline"], line["cpu_baseline"]
    assert roof["bound"] == "hbm" and roof["achieved"] > 0 and roof["avg_launch_ms"] > 0 and roof["bytes_per_launch"] > 0
    assert cpu["value"] > 0 and cpu["cores"] >= 1 and cpu["kind"] == "port"
    full = json.loads(detail.read_text(), parse_constant=_no_constants)
    for key in ("kernels", "ntxent", "fusion", "grace_step", "sampled_regime", "sampled_regime_gcl", "step_accounting"):
        assert key in full, key


def test_a_graphed_attempt_must_show_that_it_trained():
    """launch_ranks accepts the graphed attempt's line only if replays happened and both fixed-draw check losses fell."""
    import bench
    ok = {"exchange": {"graphed_sub_steps": {"captured": 3, "replayed": 40, "refused": None}},
          "check_loss": {"ggd": 1.5, "kge": 1.7}, "check_loss_after": {"ggd": 1.2, "kge": 1.5}}
    assert bench.graphed_line_refused(json.dumps(ok)) is None
    assert bench.graphed_line_refused("RCCL banner\n") == "no line"
    assert "unparsable" in bench.graphed_line_refused('{"a": ')
    bad = json.loads(json.dumps(ok)); bad["check_loss_after"]["kge"] = 1.9
    assert "did not fall" in bench.graphed_line_refused(json.dumps(bad))
    bad = json.loads(json.dumps(ok)); bad["check_loss_after"]["ggd"] = None
    assert "not finite" in bench.graphed_line_refused(json.dumps(bad))
    bad = json.loads(json.dumps(ok)); bad["exchange"]["graphed_sub_steps"]["replayed"] = 0
    assert "no replays" in bench.graphed_line_refused(json.dumps(bad))
    bad = json.loads(json.dumps(ok)); bad["exchange"]["graphed_sub_steps"]["refused"] = "RuntimeError: x"
    assert "no replays" in bench.graphed_line_refused(json.dumps(bad))


# ---------------------------------------------------------------------------------- the default N-GPU run: both regimes, one line
def _leg_line(mode, n_gpus, graphed=True):
    """What a leg's rank 0 prints: the compact line of a canned record in that mode."""
    import bench
    full = _canned(n_gpus)
    full.update(mode=mode, scaling="strong" if mode == "partition" else "weak",
                value=9.0e8 if mode == "partition" else 3.6e9, ms_per_step=4.4 if mode == "partition" else 8.9)
    full["config"]["parallelism"] = (f"np{n_gpus}" if mode == "partition" else f"dp{n_gpus}")
    if mode == "replicas":
        full["exchange"] = {"form": "flat gradient all-reduce per sub-step", "exchanges_per_step": 2, "bucket_bytes_ggd": 1.8e6,
                            "bucket_bytes_kge": 3.25e7, "wire_bytes_sent_per_step": 6.0e7, "allreduce_alone_ms_ggd": 0.05,
                            "allreduce_alone_ms_kge": 0.4}
    else:
        full["exchange"]["form"] = "graphed sub-steps (hipGraph replays)" if graphed else "eager step"
        if not graphed:
            full["exchange"]["graphed_sub_steps"] = None
    return bench.compact_line(full, "bench_detail.json")


class _Args:
    gpus, mode, detail = 8, "auto", "/tmp/bench_detail.json"


def _launch(monkeypatch, capsys, script):
    """bench.launch_ranks with ``_run_ranks`` replaced by ``script(extra_env, argv) -> (stdout, codes)``; -> (exit code,
    the calls made, stdout lines)."""
    import bench
    calls = []

    def fake(args, extra_env, limit_s, setup_limit_s=None, argv=()):
        calls.append({"env": dict(extra_env), "argv": list(argv), "limit": limit_s})
        out, codes = script(dict(extra_env), list(argv))
        return out, codes, ""
    monkeypatch.setattr(bench, "_run_ranks", fake)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    for key in ("BKG_BENCH_BACKEND", "BKG_BENCH_PARTITION_GRAPH", "BKG_DIRECT_ALL_TO_ALL", "BKG_BENCH_ONE_DEVICE"):
        monkeypatch.delenv(key, raising=False)
    with pytest.raises(SystemExit) as stop:
        bench.launch_ranks(_Args())
    return stop.value.code, calls, [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]


def test_the_default_n_gpu_run_reports_both_regimes_on_one_line(monkeypatch, capsys):
    """``--mode auto`` over N GPUs = the replicas leg, then the node-partitioned leg, ONE line: headline the
    partition (BASELINE configs[3], strong scaling), ``replicas`` beside it with its own value / scaling."""
    def script(env, argv):
        mode = argv[argv.index("--mode") + 1]
        return "RCCL banner\n" + _leg_line(mode, 8) + "\n", [0] * 8
    code, calls, lines = _launch(monkeypatch, capsys, script)
    assert code == 0 and len(lines) == 1
    line = _check(lines[0], 8)
    assert line["mode"] == "partition" and line["scaling"] == "strong" and line["config"]["parallelism"] == "np8"
    rep = line["replicas"]
    assert rep["mode"] == "replicas" and rep["scaling"] == "weak" and rep["value"] == pytest.approx(3.6e9)
    assert rep["ms_per_step"] == pytest.approx(8.9) and rep["parallelism"] == "dp8" and rep["exchange"]["bucket_bytes_kge"] > 0
    assert [a["ok"] for a in line["attempts"]] == [True]
    # the replicas leg first (a detail file of its own), then the graphed partition attempt with the in-place exchanges
    assert calls[0]["argv"][:2] == ["--mode", "replicas"] and calls[0]["argv"][3].endswith("bench_detail_replicas.json")
    assert calls[1]["env"] == {"BKG_DIRECT_ALL_TO_ALL": "1", "BKG_BENCH_PARTITION_GRAPH": "1"} and calls[1]["limit"] == 75.0
    assert all(c["limit"] is not None for c in calls)                 # every attempt is time-fenced


def test_the_partition_ladder_ends_on_the_packed_exchanges(monkeypatch, capsys):
    """Graphed attempt refused (its check losses did not fall), eager attempt with the in-place all-to-all dies, the packed
    ``all_to_all_single`` form (the library's default) runs: its line is the headline and ``attempts`` says what happened."""
    def script(env, argv):
        mode = argv[argv.index("--mode") + 1]
        if mode == "replicas":
            return _leg_line(mode, 8) + "\n", [0] * 8
        if env.get("BKG_BENCH_PARTITION_GRAPH") == "1":
            bad = json.loads(_leg_line(mode, 8))
            bad["check_loss_after"] = bad["check_loss"]
            return json.dumps(bad) + "\n", [0] * 8
        if env.get("BKG_DIRECT_ALL_TO_ALL") == "1":
            return "", [-9] * 8
        return _leg_line(mode, 8, graphed=False) + "\n", [0] * 8
    code, calls, lines = _launch(monkeypatch, capsys, script)
    assert code == 0 and len(lines) == 1
    line = _check(lines[0], 8)
    assert line["mode"] == "partition" and line["exchange"]["form"] == "eager step"
    assert [a["ok"] for a in line["attempts"]] == [False, False, True]
    assert "did not fall" in line["attempts"][0]["why_not"] and "packed" in line["attempts"][2]["form"]
    assert calls[-1]["env"] == {"BKG_BENCH_PARTITION_GRAPH": "0", "BKG_DIRECT_ALL_TO_ALL": "0"}


def test_without_a_partition_line_the_replicas_leg_is_the_headline(monkeypatch, capsys):
    def script(env, argv):
        mode = argv[argv.index("--mode") + 1]
        return (_leg_line(mode, 8) + "\n", [0] * 8) if mode == "replicas" else ("", [1] + [-9] * 7)
    code, calls, lines = _launch(monkeypatch, capsys, script)
    assert code == 0 and len(lines) == 1
    line = _check(lines[0], 8)
    assert line["mode"] == "replicas" and line["scaling"] == "weak"
    assert "error" in line["partition"] and len(line["partition"]["attempts"]) == 3
    assert not any(a["ok"] for a in line["partition"]["attempts"])


def test_attempts_stay_inside_the_runs_budget(monkeypatch, capsys):
    """BKG_BENCH_TOTAL_S: what is left of the whole run's budget caps every attempt's limits, and an attempt that could not
    get two and a half minutes is not started (the driver ends a bench run after 30 minutes: a line must be out before)."""
    import time as _time
    clock = {"t": 1000.0}
    monkeypatch.setattr(_time, "time", lambda: clock["t"])
    monkeypatch.setenv("BKG_BENCH_TOTAL_S", "1000")

    def script(env, argv):
        mode = argv[argv.index("--mode") + 1]
        clock["t"] += 300.0 if mode == "replicas" else 350.0            # every leg / attempt takes this long
        if mode == "replicas":
            return _leg_line(mode, 8) + "\n", [0] * 8
        return "", [-9] * 8                                              # every partition attempt fails
    code, calls, lines = _launch(monkeypatch, capsys, script)
    line = _check(lines[0], 8)
    assert code == 0 and line["mode"] == "replicas"
    att = line["partition"]["attempts"]
    assert [a["ok"] for a in att] == [False, False, False] and "not started" in att[2]["why_not"]
    assert len(calls) == 3                                               # replicas, graphed, eager; the packed retry had no time left
    assert calls[1]["limit"] == 75.0 and calls[2]["limit"] <= 350.0 - 60.0 + 1e-9     # (1000 - 650 s left, minus the margin)


def test_no_leg_at_all_is_a_failure(monkeypatch, capsys):
    code, calls, lines = _launch(monkeypatch, capsys, lambda env, argv: ("", [1] * 8))
    assert code != 0 and not lines
