#!/usr/bin/env python3
"""Dev: the run-based one-pass DistMult kernel for the POSITIVES (bkg_kge_onepass_f32) against the two-pass form on the
bench graph: values (loss, scores, dz, d rel_emb) and time.  Needs an MI355X."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from biomedkg_amd import _lib as L
if os.environ.get("BKG_DEV_LIB"):
    L.LIB_PATH = os.environ["BKG_DEV_LIB"]
from biomedkg_amd import ops, synthetic

dev = "cuda"
n, e, r, d = 130_000, 4_000_000, 30, 256
if len(sys.argv) > 1 and sys.argv[1] == "small":
    n, e = 9_000, 200_000
g = synthetic.primekg_like(n, e, r, seed=42)
ei, et = g["edge_index"].to(dev), g["edge_type"].to(dev)
e = ei.size(1)
ops.set_matmul_precision("bf16")
torch.manual_seed(0)
z = (torch.randn(n, d, device=dev) * 0.5).requires_grad_(True)
rel = (torch.randn(r, d, device=dev) * 0.5).requires_grad_(True)
empty_ei = torch.zeros(2, 0, dtype=torch.int64, device=dev)
empty_et = torch.zeros(0, dtype=torch.int64, device=dev)


def two_pass():
    z.grad = rel.grad = None
    loss, pp, _ = ops.kge_bce_loss(z, rel, ei, et, empty_ei, empty_et, ops.KGE_DISTMULT, total=2 * e)
    loss.backward()
    return loss.detach(), pp, z.grad.clone(), rel.grad.clone()


nb = L.lib().bkg_kge_onepass_batch(d, 1)
lists = ops.onepass_lists_of(ei, et, n, r, nb)
print("items", lists.n_items, "batches", lists.n_batches, "records", lists.n_records, "nb", nb)


def one_pass(timing=None):
    zt = ops.to_bf16(z.detach())
    total = 2 * e
    dz = torch.empty(n, d, dtype=torch.float32, device=dev)
    blocks = L.lib().bkg_kge_onepass_blocks(lists.n_items)
    part = torch.empty(blocks * 4, dtype=torch.float32, device=dev)
    records = torch.empty(lists.n_records, d, dtype=torch.float32, device=dev)
    score = torch.empty(e, dtype=torch.float32, device=dev)
    sp = lists.split.use()
    partials = sp.partials(d, dev)
    ev0, ev1, ev2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    ev0.record()
    L.call("bkg_kge_onepass_f32", L.ptr(zt), 1, zt.stride(0), n, d, L.ptr(rel.detach()), rel.stride(0), r, L.ptr(lists.other),
           L.ptr(lists.tid), L.ptr(lists.batches), L.ptr(lists.items), lists.n_items, 1.0, 1.0 / total, L.ptr(score), L.ptr(part),
           L.ptr(records), L.ptr(dz), dz.stride(0), L.sref(sp.c), L.ptr(partials), L.stream())
    ev1.record()
    drel = ops.spmm(lists.rec_ptr, lists.rec_order, None, records, r, split=lists.rec_split)
    ev2.record()
    if timing is not None:
        timing.append((ev0, ev1, ev2))
    loss = part.double().sum() / total
    return loss.float(), score, dz, drel


a = two_pass()
b = one_pass()
torch.cuda.synchronize()
print("loss", float(a[0]), float(b[0]))
print("scores max abs diff", float((a[1] - b[1]).abs().max()), "scale", float(a[1].abs().max()))
print("dz rel err", float((a[2] - b[2]).norm() / a[2].norm()), "drel rel err", float((a[3] - b[3]).norm() / a[3].norm()))
for _ in range(3):
    one_pass()
tm = []
for _ in range(10):
    one_pass(tm)
torch.cuda.synchronize()
print("one-pass ms: walk %.3f  records->drel %.3f" % (sum(x.elapsed_time(y) for x, y, _ in tm) / 10, sum(y.elapsed_time(w) for _, y, w in tm) / 10))
ops.KERNEL_TIMERS = {}
for _ in range(3):
    two_pass()
ops.KERNEL_TIMERS = {}
for _ in range(10):
    two_pass()
torch.cuda.synchronize()
for k, rec in ops.KERNEL_TIMERS.items():
    t = [x.elapsed_time(y) for x, y in rec["events"]]
    print("two-pass", k, "ms per call %.3f x %d per step" % (sum(t) / len(t), len(t) // 10))
