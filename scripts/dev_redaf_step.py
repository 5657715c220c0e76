#!/usr/bin/env python3
"""ReDAF's training-mode call at the bench size (bench.py's fusion.redaf figures alone), and the drop rate of its masks.
python scripts/dev_redaf_step.py  (MI355X)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from biomedkg_amd import ops, synthetic  # noqa: E402
from biomedkg_amd.utils.fusion import ReDAF  # noqa: E402

dev = "cuda"
n = 130_000
x = synthetic.lm_features(n, 2, 768, seed=3).to(dev)
torch.manual_seed(3)
r = ReDAF(768).to(dev).train()
for mode in ("bf16", "fp32"):
    ops.set_matmul_precision(mode)
    ms = bench._event_ms(lambda: r(x), 20, warm=3)
    print(f"{mode}: {ms:.3f} ms per call")
ops.set_matmul_precision("fp32")
r.eval()
ops.set_matmul_precision("bf16")
x.add_(0.0)
print(f"bf16, eval mode (no masks): {bench._event_ms(lambda: (x.add_(0.0), r(x)), 20, warm=3):.3f} ms per call incl. an add_ over x")
print(f"the add_ alone: {bench._event_ms(lambda: x.add_(0.0), 20, warm=3):.3f} ms")
ops.set_matmul_precision("fp32")
