#!/usr/bin/env python3
"""Weight-gradient products over two bf16 tables (gemm_tn16.hip): result against an fp64 product of the same bf16
values, and HIP-event timings (each includes the zero fill of the result and one launch gap)."""
import json
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops  # noqa: E402
sys.path.insert(0, "scripts")
from bench_gemm import timeit            # noqa: E402


def check(n=130_000, s=300_000, g=30):
    dev = "cuda"
    torch.manual_seed(0)
    out = {}
    for (k, m, nn) in ((n, 256, 256), (n, 256, 768), (5000, 128, 128), (4100, 256, 128), (4097, 128, 256)):
        a, b = torch.randn(k, m, device=dev).bfloat16(), torch.randn(k, nn, device=dev).bfloat16()
        c = ops.gemm(a, b, trans_a=True)
        ref = a.double().t() @ b.double()
        out[f"tn {k}x{m}x{nn}"] = float((c.double() - ref).abs().max() / ref.abs().max())
    w = torch.arange(1, g + 1, dtype=torch.float64).pow(-1.2)
    sizes = (w / w.sum() * s).long()
    sizes[0] += s - sizes.sum()
    sizes[5] = 0
    sizes[0] += s - sizes.sum()
    gptr = torch.cat([torch.zeros(1, dtype=torch.long), sizes.cumsum(0)]).to(torch.int32).to(dev)
    h = torch.randn(s, 256, device=dev).bfloat16()
    table = torch.randn(n, 256, device=dev).bfloat16()
    rows = torch.randint(0, n, (s,), device=dev, dtype=torch.int32)
    dy = table[rows.long()]
    c0 = ops.gemm_grouped_k(h, dy, gptr, g)
    c1 = ops.gemm_grouped_k(h, table, gptr, g, b_rows=rows)
    ref = torch.stack([h[gptr[i]:gptr[i + 1]].double().t() @ dy[gptr[i]:gptr[i + 1]].double() for i in range(g)])
    out["grouped"] = float((c0.double() - ref).abs().max() / ref.abs().max())
    out["grouped gathered"] = float((c1.double() - ref).abs().max() / ref.abs().max())
    return out, (h, dy, table, rows, gptr, g)


def times(pack, n=130_000):
    h, dy, table, rows, gptr, g = pack
    dev = "cuda"
    r = {}
    for (m, nn) in ((256, 256), (256, 768), (512, 256)):
        a, b = torch.randn(n, m, device=dev).bfloat16(), torch.randn(n, nn, device=dev).bfloat16()
        r[f"tn {m}x{nn}"] = round(timeit(lambda: ops.gemm(a, b, trans_a=True)) * 1e3, 1)
    r["grouped"] = round(timeit(lambda: ops.gemm_grouped_k(h, dy, gptr, g)) * 1e3, 1)
    r["gathered"] = round(timeit(lambda: ops.gemm_grouped_k(h, table, gptr, g, b_rows=rows)) * 1e3, 1)
    return r


def main():
    ops.set_matmul_precision("bf16")
    err, pack = check()
    print(json.dumps({"rel_err": err}))
    assert all(v < 2e-3 for v in err.values()), err
    print(json.dumps({"us": times(pack)}))
    ops.set_deterministic(True)
    err, _ = check()
    print(json.dumps({"deterministic rel_err": err, "us": times(pack)}))
    ops.set_deterministic(False)


if __name__ == "__main__":
    main()
