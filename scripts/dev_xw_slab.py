#!/usr/bin/env python3
"""The K = 768 products with the weight slab in registers (gemm_xw8k_kernel) against the streaming kernel (BKG_XW_SLAB=0):
ReDAF over [130000, 2, 768] bf16 rows, the AttentionFusion-sized product 260000 x 768 x 768, the first GCN layer's
130000 x 768 -> 256."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biomedkg_amd import ops  # noqa: E402

DEV = "cuda"


def ms(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ops.set_matmul_precision("bf16")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 130_000
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(n, 2, 768, device=DEV, generator=g).bfloat16()
    w = torch.randn(768, 768, device=DEV, generator=g) / 768 ** 0.5
    w1 = torch.randn(256, 768, device=DEV, generator=g) / 768 ** 0.5
    b = torch.randn(768, device=DEV, generator=g) * 0.1
    scale = torch.rand(2, 768, device=DEV, generator=g) + 0.25
    drop = ops.Dropout(0.1, keep=None, seed=5, stream_id=1)
    x2 = x.view(2 * n, 768)
    x1 = x2[:n].contiguous()
    out = {}
    for slab in ("1", "0"):
        os.environ["BKG_XW_SLAB"] = slab
        rec = {}
        rec["redaf_train_ms"] = ms(lambda: ops.redaf_fusion(x, w, b, scale, drop))
        rec["redaf_eval_ms"] = ms(lambda: ops.redaf_fusion(x, w, b, scale, None))
        rec["xw_260k_768_768_bf16out_ms"] = ms(lambda: ops.gemm(x2, w, trans_b=True, out_dtype=torch.bfloat16))
        rec["xw_260k_768_768_f32out_ms"] = ms(lambda: ops.gemm(x2, w, trans_b=True))
        rec["xw_130k_768_256_bf16out_ms"] = ms(lambda: ops.gemm(x1, w1, trans_b=True, out_dtype=torch.bfloat16))
        fl = 2.0 * 2 * n * 768 * 768
        rec["redaf_train_TF"] = fl / rec["redaf_train_ms"] / 1e9
        rec["redaf_eval_TF"] = fl / rec["redaf_eval_ms"] / 1e9
        rec["xw_260k_bf16out_TF"] = fl / rec["xw_260k_768_768_bf16out_ms"] / 1e9
        rec["xw_130k_256_TF"] = 2.0 * n * 768 * 256 / rec["xw_130k_768_256_bf16out_ms"] / 1e9
        out["slab" if slab == "1" else "streaming"] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
