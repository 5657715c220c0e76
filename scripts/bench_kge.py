#!/usr/bin/env python3
"""KGE scorer + BCE forward / backward at bench scale (N=130k, T=4M+4M, d=256); run under
rocprofv3 --kernel-trace and summarise with scripts/ktrace.py.   python scripts/bench_kge.py [fp32|bf16] [kind]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops, sampling, synthetic  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    kind = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ops.set_matmul_precision(prec)
    dev = "cuda"
    g = synthetic.primekg_like(130_000, 4_000_000, 30, seed=42)
    ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
    torch.manual_seed(0)
    z = (torch.randn(130_000, 256, device=dev) * 0.3).requires_grad_()
    rel = (torch.randn(30, 256, device=dev) * 0.3).requires_grad_()
    neg = sampling.negative_sampling(ei, 130_000, ei.size(1))
    nt = et[torch.randperm(et.numel(), device=dev)]
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, _, _ = ops.kge_bce_loss(z, rel, ei, et, neg, nt, kind)
        loss.backward()
        torch.cuda.synchronize()
        print(f"iter {it}: {1e3 * (time.perf_counter() - t0):.2f} ms  loss {float(loss.detach()):.5f}")
        z.grad = None; rel.grad = None


if __name__ == "__main__":
    main()
