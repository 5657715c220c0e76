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
    out = _run({"BKG_BENCH_ONE_DEVICE": "1", "BKG_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert out.returncode != 0
    assert out.stderr.count("needs an MI355X") == 2, out.stderr[-1500:]
    assert out.stdout.strip() == ""


def test_world_size_must_match_gpus():
    out = _run({"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2")
    assert out.returncode != 0 and "started 3 rank" in out.stderr
