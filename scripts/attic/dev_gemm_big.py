"""Dev: the fusers' products (M = 8192 .. 260k, K = 768, N = 768 / 2304) through the streaming kernel vs the LDS-tiled one."""
import sys
import torch
sys.path.insert(0, ".")
from biomedkg_amd import ops, _lib as L
ops.set_matmul_precision("bf16")
dev = "cuda"


def t(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for m in (8192, 65536, 260_000):
    for n in (768, 2304):
        k = 768
        a32 = torch.randn(m, k, device=dev); a16 = a32.bfloat16()
        w = torch.randn(n, k, device=dev) * 0.03
        bias = torch.randn(n, device=dev)
        fl = 2.0 * m * n * k
        for name, a in (("fp32 A", a32), ("bf16 A", a16)):
            ms = t(lambda: ops.gemm(a, w, trans_b=True, bias=bias))
            print(f"M={m} N={n} streaming {name}: {ms:.3f} ms {fl / ms / 1e9:.0f} TF", flush=True)
        out = torch.empty(m, n, device=dev)
        def tiled():
            L.call("bkg_gemm_f32", 0, 1, m, n, k, 1.0, L.ptr(a32), a32.stride(0), L.ptr(w), w.stride(0), 0.0, L.ptr(out), out.stride(0),
                   L.ptr(bias), 0, None, 0, 1.0, 1, L.stream())
        ms = t(tiled)
        ref = ops.gemm(a32, w, trans_b=True, bias=bias)
        tiled()
        print(f"M={m} N={n} LDS-tiled (fp32 A, bf16 MFMA): {ms:.3f} ms {fl / ms / 1e9:.0f} TF  max diff {float((out - ref).abs().max()):.3e}", flush=True)
