#!/usr/bin/env python3
"""The (node, role)-sorted one-pass walk with d rel_emb on the matrix cores (csrc/kge_onepass_mx.hip) on uniform negatives of
the bench shape: against a chunked fp64 restatement on the same bf16 table, and timed next to the two-pass form (fused scorer
+ incidence streams + node-side stream reduce).  python scripts/dev_kge_mx.py [n_nodes] [triplets]  (MI355X)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from biomedkg_amd import _lib as L, ops  # noqa: E402
from biomedkg_amd.graph import _Sorter  # noqa: E402

dev = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 130_000
t = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
r, d, label = 30, 256, 0.0
torch.manual_seed(0)
ops.set_matmul_precision("bf16")
z = torch.randn(n, d, device=dev) * 0.3
zt = ops.to_bf16(z)
rel = (torch.randn(r, d, device=dev) * 0.3).contiguous()
ei = torch.randint(0, n, (2, t), device=dev)
et = torch.randint(0, r, (t,), device=dev)
inv_t = 1.0 / (2 * t)


def structures_with_positions():
    """(node, role)-sorted lists WITH the entries' positions (the walk writes scores by them): the generic sort + streams."""
    role = torch.cat([torch.zeros(t, dtype=torch.int64, device=dev), torch.ones(t, dtype=torch.int64, device=dev)])
    _, perm, rowptr2 = _Sorter.sort_rowptr(role, ei.view(-1), 2, n, 2 * n + 1, 1)
    other = torch.empty(2 * t, dtype=torch.int32, device=dev)
    relid = torch.empty(2 * t, dtype=torch.int32, device=dev)
    L.call("bkg_kge_incidence_streams", L.ptr(perm), 2 * t, t, L.ptr(ei[0]), L.ptr(ei[1]), L.ptr(et), None, None,
           L.ptr(other), L.ptr(relid), None, L.stream())
    return perm, rowptr2, other | (relid << 24)


def structures():
    """What a training step builds: bkg_kge_role_incidence (keys + packed payload, one radix sort of pairs, lower bound)."""
    ws_bytes = L.lib().bkg_sort_workspace_bytes(2 * t)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    keys, vals, skeys, packed = (torch.empty(2 * t, dtype=torch.int32, device=dev) for _ in range(4))
    rowptr2 = torch.empty(2 * n + 1, dtype=torch.int32, device=dev)
    L.call("bkg_kge_role_incidence", L.ptr(ei[0]), L.ptr(ei[1]), L.ptr(et), t, n, L.ptr(keys), L.ptr(vals), L.ptr(skeys),
           L.ptr(packed), L.ptr(rowptr2), L.ptr(ws), ws_bytes, L.stream())
    return rowptr2, packed


def walk(perm, rowptr2, packed, scores=True):
    blocks = L.lib().bkg_kge_onepass_mx_blocks(n)
    score = torch.empty(t, device=dev) if scores else None
    loss_part = torch.empty(4 * blocks, device=dev)
    dz = torch.empty(n, d, device=dev)
    drel_part = torch.empty(blocks, 32, d, device=dev)
    L.call("bkg_kge_onepass_mx_f32", L.ptr(zt), zt.stride(0), n, L.ptr(rel), rel.stride(0), r, L.ptr(rowptr2), L.ptr(packed),
           L.ptr(perm), 2 * t, label, inv_t, L.ptr(score), L.ptr(loss_part), L.ptr(dz), dz.stride(0),
           L.ptr(drel_part), blocks, L.stream())
    return score, loss_part, dz, drel_part


perm, rowptr2, packed = structures_with_positions()
torch.cuda.synchronize()
score, loss_part, dz, drel_part = walk(perm, rowptr2, packed)
torch.cuda.synchronize()
drel = drel_part.sum(0)[:r] * inv_t

# ---- fp64 restatement on the same bf16 table, in chunks
z64, rel64 = zt.double(), rel.double()
s_ref = torch.empty(t, dtype=torch.float64, device=dev)
dz_ref = torch.zeros(n, d, dtype=torch.float64, device=dev)
drel_ref = torch.zeros(r, d, dtype=torch.float64, device=dev)
loss_ref = torch.zeros((), dtype=torch.float64, device=dev)
for lo in range(0, t, 1 << 19):
    hi = min(lo + (1 << 19), t)
    h, tl, rr = ei[0, lo:hi], ei[1, lo:hi], et[lo:hi]
    zh, ztl, rw = z64[h], z64[tl], rel64[rr]
    s = (zh * rw * ztl).sum(1)
    s_ref[lo:hi] = s
    g = (torch.sigmoid(s) - label) * inv_t
    loss_ref += (torch.clamp(s, min=0) - s * label + torch.log1p(torch.exp(-s.abs()))).sum()
    dz_ref.index_add_(0, h, g[:, None] * rw * ztl)
    dz_ref.index_add_(0, tl, g[:, None] * rw * zh)
    drel_ref.index_add_(0, rr, g[:, None] * zh * ztl)


def rel_err(a, b):
    return float((a.double() - b).norm() / b.norm())


print("score  max |diff|", float((score.double() - s_ref).abs().max()))
print("loss   rel", abs(float(loss_part.double().sum()) - float(loss_ref)) / float(loss_ref))
print("dz     rel", rel_err(dz, dz_ref), " max |diff| / max", float((dz.double() - dz_ref).abs().max() / dz_ref.abs().max()))
print("drel   rel", rel_err(drel, drel_ref), " worst row rel",
      float(((drel.double() - drel_ref).norm(dim=1) / drel_ref.norm(dim=1)).max()))

rowptr2_b, packed_b = structures()
print("bkg_kge_role_incidence: rowptr2 equal", bool(torch.equal(rowptr2_b, rowptr2)), " lists equal as multisets per node:",
      bool(torch.equal(torch.sort(packed_b.long() + (torch.repeat_interleave(torch.arange(2 * n, device=dev), (rowptr2_b[1:] - rowptr2_b[:-1]).long()) << 32))[0],
                       torch.sort(packed.long() + (torch.repeat_interleave(torch.arange(2 * n, device=dev), (rowptr2[1:] - rowptr2[:-1]).long()) << 32))[0])))
_, lp2, dz2, dp2 = walk(None, rowptr2_b, packed_b, scores=False)
print("without scores, library lists: dz rel diff", rel_err(dz2, dz.double()), " drel rel diff", rel_err(dp2.sum(0), drel_part.sum(0).double()))


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print("structures (sort by (node, role) + streams)  %.3f ms" % timed(structures))
print("walk with scores                            %.3f ms" % timed(lambda: walk(perm, rowptr2, packed)))
print("walk without scores                         %.3f ms" % timed(lambda: walk(None, rowptr2_b, packed_b, scores=False)))

# ---- the two-pass form through the op (positives empty: the negatives alone)
e0 = torch.empty(2, 0, dtype=torch.int64, device=dev)
zq = z.clone().requires_grad_()
rq = rel.clone().requires_grad_()


def two_pass():
    zq.grad = rq.grad = None
    loss = ops.kge_bce_loss(zq, rq, e0, e0[0], ei, et, ops.KGE_DISTMULT)[0]
    loss.backward()


was = getattr(ops, "KGE_ONEPASS_MX", None)
if was is not None:
    ops.KGE_ONEPASS_MX = False
print("two-pass op (fwd + bwd, incl. its incidence sort) %.3f ms" % timed(two_pass))
if was is not None:
    ops.KGE_ONEPASS_MX = was
