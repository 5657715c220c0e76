#!/usr/bin/env python3
"""Randomised shapes / layouts / dtypes for the bf16-mode kernels (streaming GEMM, weight-gradient GEMMs, bf16 SpMM
results, column sums) against fp64 algebra on the bf16-rounded operands.  python scripts/fuzz_kernels.py  (MI355X)"""
import sys, random, torch
sys.path.insert(0, ".")
from biomedkg_amd import ops
dev = "cuda"
ops.set_matmul_precision("bf16")
random.seed(0); torch.manual_seed(0)
bad = 0
for it in range(120):
    m = random.choice([1024, 1025, 1500, 2047, 4096, 5000, 9999, 33333])
    n = 32 * random.randint(1, 12)
    k = 256 * random.randint(1, 3)
    tb = random.random() < 0.5
    a16 = random.random() < 0.5
    c16 = random.random() < 0.5
    lda = k + random.choice([0, 8, 64])
    abuf = torch.randn(m, lda, device=dev)
    a = abuf[:, :k]
    if a16:
        abuf16 = abuf.bfloat16(); a = abuf16[:, :k]
    b = torch.randn((n, k) if tb else (k, n), device=dev)
    bias = torch.randn(n, device=dev) if random.random() < 0.3 else None
    act = random.choice([0, 0, 1, 2])
    if not ops._xw_ok(a, m, n, k):
        continue
    out = ops.gemm(a, b, trans_b=tb, bias=bias, act=act, out_dtype=torch.bfloat16 if c16 else torch.float32)
    ref = a.double() if not a16 else a.double()
    ref = (a.bfloat16().double() @ (b.bfloat16().double().t() if tb else b.bfloat16().double()))
    if bias is not None: ref = ref + bias.double()
    if act == 1: ref = ref.relu()
    if act == 2: ref = torch.nn.functional.elu(ref)
    err = (out.double() - ref).abs().max().item()
    tol = (2 ** -7) * ref.abs().max().item() + 1e-2 if c16 else 1e-3 * k ** 0.5
    if not (err <= tol):
        bad += 1; print("XW FAIL", m, n, k, tb, a16, c16, lda, act, err, tol)
# TN fuzz
for it in range(60):
    k = random.choice([4096, 5000, 8191, 20000, 70001])
    m = 128 * random.randint(1, 3); n = 128 * random.randint(1, 6)
    a16 = random.random() < 0.7; b16 = random.random() < 0.7
    a = torch.randn(k, m, device=dev); b = torch.randn(k, n, device=dev)
    aa = a.bfloat16() if a16 else a; bb = b.bfloat16() if b16 else b
    out = ops.gemm(aa, bb, trans_a=True)
    ref = a.bfloat16().double().t() @ b.bfloat16().double()
    err = (out.double() - ref).abs().max().item()
    if not (err <= 2e-3 * k ** 0.5):
        bad += 1; print("TN FAIL", m, n, k, a16, b16, err)
# colsum / spmm bf16-out fuzz
from biomedkg_amd import graph as G
for it in range(30):
    nn_ = random.choice([500, 3000, 20000]); e = nn_ * random.randint(2, 30)
    d = random.choice([32, 64, 128, 256, 768])
    ei = torch.randint(0, nn_, (2, e), device=dev)
    gr = G.Graph(ei, nn_); gw = gr.gcn_weights(); c = gr.by_dst
    x = torch.randn(nn_, d, device=dev)
    for tab in (x, ops.to_bf16(x)):
        r32 = ops.spmm(c.rowptr, c.col, gw.w_dst, tab, nn_, self_w=gw.self_w, relu=True, split=c.split)
        r16 = ops.spmm(c.rowptr, c.col, gw.w_dst, tab, nn_, self_w=gw.self_w, relu=True, split=c.split, out_dtype=torch.bfloat16)
        if not torch.equal(r16, r32.bfloat16()):
            bad += 1; print("SPMM16 FAIL", nn_, e, d, tab.dtype)
    cs = ops.colsum(x)
    if not torch.allclose(cs.double(), x.double().sum(0), rtol=1e-5, atol=1e-4 * nn_ ** 0.5):
        bad += 1; print("COLSUM FAIL", nn_, d)
print("fuzz done, failures:", bad)
