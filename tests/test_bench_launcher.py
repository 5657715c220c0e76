"""bench.py's own rank launcher (``python3 bench.py --gpus N`` with no torch.distributed.run around it), as far as a
box without a GPU can take it: the parent never initialises HIP, refuses a node with too few devices, starts N children
with the rendezvous environment and hands back the worst child's exit code."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=300)


def test_too_few_devices_is_refused_by_the_parent():
    import torch
    if torch.cuda.device_count() >= 2:
        return
    out = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert out.returncode != 0 and "GPU(s)" in out.stderr


def test_children_get_the_rendezvous_environment_and_their_exit_code_comes_back(tmp_path):
    import torch
    if torch.cuda.is_available():
        return                      # on a GPU box the real thing runs (tests/test_partition_hip.py)
    # no GPU here: each child must stop at "bench.py needs an MI355X" -- AFTER it has read its rank environment
    for mode, legs in (("replicas", 1), ("partition", 1), ("auto", 2)):      # auto (round 6): both regimes, one after the other
        out = _run({"BKG_BENCH_ONE_DEVICE": "1", "BKG_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "1", "--warmup", "0",
                   "--mode", mode)
        assert out.returncode != 0
        assert out.stderr.count("needs an MI355X") == 2 * legs, (mode, out.stderr[-1500:])
        assert out.stdout.strip() == ""


def test_world_size_must_match_gpus():
    out = _run({"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2")
    assert out.returncode != 0 and "started 3 rank" in out.stderr
    out = _run({"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0", "BKG_BENCH_OUTER_RANKS": "1"}, "--gpus", "2")
    assert out.returncode != 0 and "started 3 rank" in out.stderr


def test_under_an_outer_launcher_rank_zero_takes_over_and_the_others_leave():
    """``torch.distributed.run --nproc-per-node N bench.py --gpus N --mode partition``: the outer launcher's workers other than rank 0 exit
    0 at once (no GPU touched, nothing printed); rank 0 becomes bench.py's own launcher with a clean rank environment --
    here, without GPUs, its children stop at "needs an MI355X" after reading THEIR rendezvous (two of them, ranks 0 and
    1 on a port of their own, not the outer launcher's)."""
    import torch
    if torch.cuda.is_available():
        return                      # on a GPU box the real thing runs (tests/test_partition_hip.py)
    outer = {"WORLD_SIZE": "2", "LOCAL_WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1",
             "TORCHELASTIC_USE_AGENT_STORE": "True", "TORCHELASTIC_RUN_ID": "x", "BKG_BENCH_ONE_DEVICE": "1",
             "BKG_BENCH_BACKEND": "gloo"}
    part = ("--gpus", "2", "--steps", "1", "--warmup", "0", "--mode", "partition")
    other = _run(dict(outer, RANK="1", LOCAL_RANK="1"), *part)
    assert other.returncode == 0 and other.stdout.strip() == "" and other.stderr.strip() == ""
    lead = _run(dict(outer, RANK="0", LOCAL_RANK="0"), *part)
    assert lead.returncode != 0
    assert lead.stderr.count("needs an MI355X") == 2, lead.stderr[-1500:]
    # kept as the ranks on request: this process IS rank 1 then (and stops at the missing GPU itself)
    kept = _run(dict(outer, RANK="1", LOCAL_RANK="1", BKG_BENCH_OUTER_RANKS="1"), *part)
    assert kept.returncode != 0 and kept.stderr.count("needs an MI355X") == 1
    # the default mode over N GPUs (round 6: both regimes on fresh ranks): rank 0 takes over there too, the others leave
    own = _run(dict(outer, RANK="1", LOCAL_RANK="1"), "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert own.returncode == 0 and own.stdout.strip() == "" and own.stderr.strip() == ""
    lead = _run(dict(outer, RANK="0", LOCAL_RANK="0"), "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert lead.returncode != 0 and lead.stderr.count("needs an MI355X") == 4, lead.stderr[-1500:]
    # --mode replicas alone runs in the outer launcher's own processes (nothing to fall back from)
    own = _run(dict(outer, RANK="1", LOCAL_RANK="1"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--mode", "replicas")
    assert own.returncode != 0 and own.stderr.count("needs an MI355X") == 1
