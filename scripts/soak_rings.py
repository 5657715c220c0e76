#!/usr/bin/env python3
"""Run-to-run repeatability of the full-size bench step in deterministic bf16 mode over many optimizer steps: the same
seeded GGD + KGE modules are trained twice for `steps` steps (default 60) and every parameter must agree bit for bit at
the end.  A lost wait in one of the pipelined (LDS-DMA ring) kernels -- or any other race -- shows up as a difference.
python scripts/soak_rings.py [steps]  (MI355X)"""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
import bench                                   # noqa: E402
from biomedkg_amd import ops, rand as R        # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extras"]
args = bench.parse()
dev = torch.device("cuda:0")
ops.set_matmul_precision("bf16")
ops.set_deterministic(True)


def run():
    torch.manual_seed(0)
    R.reseed_default(123)
    gstep, kstep = bench.build_replica_steps(args, 0, 1, dev)[:2]
    losses = []
    for _ in range(steps):
        losses.append((float(gstep().detach()), float(kstep().detach())))
    torch.cuda.synchronize()
    h = hashlib.sha256()
    count = 0
    for st in (gstep, kstep):
        for p in st.params:
            h.update(p.detach().cpu().numpy().tobytes())
            count += 1
    assert count > 0
    return h.hexdigest(), losses


a, la = run()
b, lb = run()
print("first losses", la[:2], "last", la[-1])
print("hash A", a[:16], "hash B", b[:16], "losses equal:", la == lb)
assert la == lb and a == b, "run-to-run difference in deterministic mode"
print(f"soak_rings: {steps} steps x 2 runs, bit-identical")
