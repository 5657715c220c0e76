#!/usr/bin/env python3
"""Dev: host synchronisations inside the bench's steps after a W-step warm-up (torch's sync debug mode), step by step."""
import sys
import warnings

import torch

sys.path.insert(0, ".")
sys.argv = sys.argv[:1]
import bench  # noqa: E402
from biomedkg_amd import ops  # noqa: E402

args = bench.parse()
dev = torch.device("cuda:0")
ops.set_matmul_precision("bf16")
step_g, step_k, n, e, r = bench.build_replica_steps(args, 0, 1, dev)
for _ in range(3):
    step_g(); step_k()
step_k.finish()
import gc
import os
if os.environ.get("DEV_GC") == "1":
    gc.collect(); gc.disable()
if os.environ.get("DEV_EVENTS") == "1":
    ops.reserve_events(700)
torch.cuda.synchronize()
ops.KERNEL_TIMERS = {} if os.environ.get("DEV_TIMERS", "1") == "1" else None
torch.cuda.set_sync_debug_mode("warn")
for i in range(4):
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); step_g(); step_k(); b.record()
    torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    print("step", i, round(a.elapsed_time(b), 3), "ms; syncs:", len(w))
    for x in w[:6]:
        print("   ", str(x.message)[:150], "|", x.filename.split("/")[-1], x.lineno)
    torch.cuda.set_sync_debug_mode("warn")
torch.cuda.set_sync_debug_mode("default")
