#!/usr/bin/env python3
"""Randomised shapes / strides / group layouts for the LDS-DMA ring kernels (gemm_xw8 plain / grouped / gathered,
gemm_tn16 split-K / grouped / gathered) against fp64 algebra on the same bf16 values, each case launched several times
among other work: the activation x weight kernels have a fixed accumulation order, so two launches must agree bit for bit
(a lost wait in the ring would show up as run-to-run differences).  python scripts/fuzz_rings.py [iters]  (MI355X)"""
import random
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import ops  # noqa: E402

dev = "cuda"
ops.set_matmul_precision("bf16")
random.seed(1)
torch.manual_seed(1)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0


def group_ptr(total, g):
    cuts = sorted(random.choice([0, total, random.randint(0, total)]) if random.random() < 0.3 else random.randint(0, total)
                  for _ in range(g - 1))
    return torch.tensor([0] + cuts + [total], dtype=torch.int32, device=dev)


noise = torch.randn(1 << 22, device=dev)
for it in range(iters):
    kind = random.choice(["plain", "gated", "grouped", "gathered", "tn", "tn_grouped", "tn_gathered"])
    m = random.choice([256, 257, 320, 1000, 4095, 4160, 20_000, 33_333, 70_001, 150_000])
    c16 = random.random() < 0.5
    cdt = torch.bfloat16 if c16 else torch.float32
    tol = dict(rtol=2 ** -7, atol=3e-2) if c16 else dict(rtol=1e-4, atol=3e-3)
    try:
        if kind == "plain":
            lda = 256 + random.choice([0, 8, 256])
            a = torch.randn(m, lda, device=dev).bfloat16()[:, lda - 256:]
            w = torch.randn(256, 256, device=dev) * 0.1
            tb = random.random() < 0.5
            kw = {}
            if random.random() < 0.4:
                kw = dict(bias=torch.randn(256, device=dev), act=random.choice([0, 1, 2]), alpha=random.choice([1.0, 0.25]))
            outs = []
            for rep in range(3):
                outs.append(ops.gemm(a, w, trans_b=tb, out_dtype=cdt, **kw))
                noise.mul_(1.0001)
            ref = a.double() @ (w.bfloat16().double().t() if tb else w.bfloat16().double())
            ref = ref * kw.get("alpha", 1.0) + (kw["bias"].double() if "bias" in kw else 0.0)
            if kw.get("act") == 1:
                ref = ref.clamp_min(0)
            elif kw.get("act") == 2:
                ref = torch.where(ref > 0, ref, torch.expm1(ref))
            torch.testing.assert_close(outs[0].double(), ref, **tol)
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "run-to-run difference"
        elif kind == "gated":
            # the gated product (hand-issued gate loads counted into the ring's vmcnt ladder): the table bit for bit against
            # the product + gate pass, and run-to-run (its column sums have a fixed order too)
            mg = max(m, 4096)
            a = torch.randn(mg, 256, device=dev).bfloat16()
            w = torch.randn(256, 256, device=dev) * 0.1
            act = torch.relu(torch.randn(mg, 256, device=dev)).bfloat16()
            scale = random.choice([1.0, 1.25])
            outs = []
            for rep in range(3):
                outs.append(ops.gemm_gated(a, w, act, scale))
                noise.mul_(1.0001)
            ref = ops.to_bf16(ops.gemm(a, w, out_dtype=torch.bfloat16), gate=act, gate_scale=scale)
            assert torch.equal(outs[0][0].view(torch.int16), ref.view(torch.int16)), "gated table differs from the gate pass"
            torch.testing.assert_close(outs[0][1].double(), ref.double().sum(0), rtol=1e-4, atol=1e-5 * float(ref.float().abs().sum(0).max()))
            for o in outs[1:]:
                assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]), "run-to-run difference"
        elif kind in ("grouped", "gathered"):
            g = random.choice([1, 2, 7, 30, 63, 64, 100])
            gp = group_ptr(m, g)
            w = torch.randn(g, 256, 256, device=dev) * 0.1
            tb = random.random() < 0.5
            if kind == "gathered":
                n = random.choice([500, 6000, 130_000])
                table = torch.randn(n, 256, device=dev).bfloat16()
                idx = torch.randint(0, n, (m,), device=dev, dtype=torch.int32)
                rows, kw = table[idx.long()], dict(a_rows=idx)
            else:
                table = rows = torch.randn(m, 256, device=dev).bfloat16()
                kw = {}
            outs = []
            for rep in range(3):
                outs.append(ops.gemm_grouped_rows(table, w, gp, trans_b=tb, out_dtype=cdt, **kw))
                noise.mul_(1.0001)
            seg = torch.repeat_interleave(torch.arange(g, device=dev), (gp[1:] - gp[:-1]).long())
            wd = w.bfloat16().double()
            ref = torch.empty(m, 256, dtype=torch.float64, device=dev)
            for i in range(g):
                sl = slice(int(gp[i]), int(gp[i + 1]))
                ref[sl] = rows[sl].double() @ (wd[i].t() if tb else wd[i])
            torch.testing.assert_close(outs[0].double(), ref, **tol)
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "run-to-run difference"
        else:
            k = max(m, 4096)
            mm, nn = random.choice([(128, 128), (256, 256), (256, 768), (512, 256), (128, 384)])
            a = torch.randn(k, mm, device=dev).bfloat16()
            if kind == "tn":
                b = torch.randn(k, nn, device=dev).bfloat16()
                out = ops.gemm(a, b, trans_a=True)
                ref = a.double().t() @ b.double()
                torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=2e-3 * k ** 0.5)
            else:
                g = random.choice([1, 3, 30, 70])
                gp = group_ptr(k, g)
                if kind == "tn_gathered":
                    n = random.choice([500, 6000, 130_000])
                    table = torch.randn(n, nn, device=dev).bfloat16()
                    idx = torch.randint(0, n, (k,), device=dev, dtype=torch.int32)
                    b = table[idx.long()]
                    if not ops.gathered_k_ok(a, table):
                        continue
                    out = ops.gemm_grouped_k(a, table, gp, g, b_rows=idx)
                else:
                    b = torch.randn(k, nn, device=dev).bfloat16()
                    out = ops.gemm_grouped_k(a, b, gp, g)
                for i in range(g):
                    sl = slice(int(gp[i]), int(gp[i + 1]))
                    ref = a[sl].double().t() @ b[sl].double()
                    torch.testing.assert_close(out[i].double(), ref, rtol=1e-4, atol=2e-3 * max(1, sl.stop - sl.start) ** 0.5)
    except AssertionError as e:
        bad += 1
        print("FAIL", it, kind, m, str(e).splitlines()[0][:160])
torch.cuda.synchronize()
print(f"fuzz_rings: {iters} cases, {bad} failures")
