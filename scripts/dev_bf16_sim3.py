"""Dev: ONE R-GCN layer, forward + backward through ops.rgcn_conv in bf16 mode, against tests/bf16_model.rgcn_layer."""
import sys

import torch

sys.path.insert(0, ".")
from biomedkg_amd import graph as G, ops, synthetic  # noqa: E402
from tests import bf16_model as B  # noqa: E402

DEV = "cuda"
n, r, d = 130_000, 30, 256
g = synthetic.primekg_like(n, 4_000_000, r, seed=42)
ei, et = g["edge_index"].to(DEV), g["edge_type"].to(DEV)
torch.manual_seed(42)
ops.set_matmul_precision("bf16")
rg = G.rel_graph_of(ei, et, n, r)


def err(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


src, dst = ei[0], ei[1]
key = et * n + dst
uniq, seg_of_edge, cnt = torch.unique(key, return_inverse=True, return_counts=True)
seg_dst, seg_rel = uniq % n, uniq // n
inv_cnt = 1.0 / cnt[seg_of_edge].double()

for hidden in (True, False):
    x = torch.randn(n, d, device=DEV).bfloat16().requires_grad_()
    weight = (torch.randn(r, d, d, device=DEV) * 0.06).requires_grad_()
    root = (torch.randn(d, d, device=DEV) * 0.06).requires_grad_()
    bias = (torch.randn(d, device=DEV) * 0.05).requires_grad_()
    out = ops.rgcn_conv(x, weight, root, bias, rg, relu=hidden, drop=None, hidden=hidden)
    gout = torch.randn(n, d, device=DEV) * 0.01
    if hidden:
        gout = gout.bfloat16()
    out.backward(gout)
    xm = x.detach().double().requires_grad_()
    Wm, Rm, bm = B.leaf(weight), B.leaf(root), B.leaf(bias)
    om = B.rgcn_layer(xm, src, dst, seg_of_edge, seg_dst, seg_rel, inv_cnt, Wm, Rm, bm, n, relu=hidden, hidden=hidden)
    om.backward(gout.double())
    print(f"hidden={hidden}: out {err(out.detach(), om.detach()):.2e} dx {err(x.grad, B.bf(xm.grad)):.2e} "
          f"(unrounded model dx {err(x.grad, xm.grad):.2e}) dW {err(weight.grad, Wm.grad):.2e} droot {err(root.grad, Rm.grad):.2e} "
          f"db {err(bias.grad, bm.grad):.2e}", flush=True)
    # variants of the model for dx: which rounding is off?
    for name, kw in (("no rb on root branch", dict(root_rb=False)), ("rb on table branch too", dict(table_rb=True))):
        xm2 = x.detach().double().requires_grad_()
        W2, R2, b2 = B.leaf(weight), B.leaf(root), B.leaf(bias)
        n_seg = uniq.numel()
        xin_t = B.rb(xm2) if kw.get("table_rb") else xm2
        xin_r = B.rb(xm2) if kw.get("root_rb", True) else xm2
        h = B.rb(B.rf(B.aggregate(xin_t, src, seg_of_edge, inv_cnt, None, n_seg)))
        w16 = B.rf(W2)
        counts = torch.bincount(seg_rel, minlength=r).tolist()
        parts, a = [], 0
        for rr, cc in enumerate(counts):
            if cc:
                parts.append(h[a:a + cc] @ w16[rr])
            a += cc
        y = B.rf(torch.cat(parts))
        xr = xin_r @ B.rf(R2)
        if hidden:
            xr = B.rf(xr)
        pre = B.rb(B.aggregate(y, torch.arange(n_seg, device=DEV), seg_dst, torch.ones(n_seg, dtype=torch.float64, device=DEV), None, n) + xr + b2)
        o2 = torch.relu(pre) if hidden else pre
        o2 = B.rb(B.rf(o2)) if hidden else o2
        o2.backward(gout.double())
        print(f"   {name}: dx {err(x.grad, B.bf(xm2.grad)):.2e}", flush=True)
