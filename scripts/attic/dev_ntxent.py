#!/usr/bin/env python3
"""NT-Xent v2 timings (HIP events, median): whole op and kernel-only flops accounting.
    python scripts/dev_ntxent.py [fp32|bf16] ..."""
import json
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops  # noqa: E402
from scripts.bench_kernels import timeit  # noqa: E402


def main():
    modes = sys.argv[1:] or ["fp32", "bf16"]
    torch.manual_seed(0)
    for mode in modes:
        peak = 155.0 if mode == "fp32" else 2500.0
        for n, h in ((5000, 128), (16384, 128), (32768, 256), (130000, 256)):
            if mode == "fp32" and n > 40000:
                reps = 2
            else:
                reps = 10
            h1 = torch.randn(n, h, device="cuda", requires_grad=True)
            h2 = (0.6 * h1.detach() + 0.4 * torch.randn(n, h, device="cuda")).requires_grad_()
            m = 2 * n
            fwd = timeit(lambda: ops.ntxent_loss(h1, h2, 0.2, operands=mode), reps=reps, warm=1)
            loss = ops.ntxent_loss(h1, h2, 0.2, operands=mode)
            bwd = timeit(lambda: torch.autograd.grad(loss, (h1, h2), retain_graph=True), reps=reps, warm=1)
            f_fwd, f_bwd = 1.0 * m * m * h, 4.0 * m * m * h       # executed: symmetric forward, full backward
            print(json.dumps({"kernel": "ntxent2", "operands": mode, "N": n, "h": h, "fwd_ms": round(fwd, 4), "bwd_ms": round(bwd, 4),
                              "fwd_TFLOPs_executed": round(f_fwd / fwd / 1e9, 1), "bwd_TFLOPs_executed": round(f_bwd / bwd / 1e9, 1),
                              "fwd_frac_peak": round(f_fwd / fwd / 1e9 / peak, 3), "bwd_frac_peak": round(f_bwd / bwd / 1e9 / peak, 3),
                              "loss": float(loss)}), flush=True)


if __name__ == "__main__":
    main()
