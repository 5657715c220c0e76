"""Skewed / large inputs that take the long-row-splitting, split-K and relation-sorted kernel paths,
checked against fp64 algebra on the same inputs.  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _skewed_graph(n, e, hub_share=0.3, seed=0):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    hub = int(e * hub_share)
    dst[:hub] = 7                      # one destination with ~hub in-edges
    src[hub:2 * hub] = 11              # one source with ~hub out-edges
    perm = torch.randperm(e, generator=g)
    return torch.stack([src[perm], dst[perm]])


@pytest.mark.parametrize("d", [128, 256, 768])
def test_spmm_row_split_matches_fp64(d):
    from biomedkg_amd import graph as G, ops
    n, e = 3000, 200_000
    ei = _skewed_graph(n, e)
    torch.manual_seed(1)
    x = torch.randn(n, d)
    gr = G.Graph(ei.to(DEV), n)
    assert gr.by_dst.split is not None and int(gr.by_dst.split.totals[2]) >= 1   # long rows exist
    gw = gr.gcn_weights()
    from oracle import pyg_ops as P
    e2, w = P.gcn_norm(ei, n)
    ref = torch.zeros(n, d, dtype=torch.float64).index_add_(0, e2[1], x.double()[e2[0]] * w.double()[:, None])
    bias = torch.randn(d)
    out = ops.spmm(gr.by_dst.rowptr, gr.by_dst.col, gw.w_dst, x.to(DEV), n, self_w=gw.self_w, bias=bias.to(DEV),
                   relu=True, split=gr.by_dst.split)
    torch.testing.assert_close(out.cpu().double(), (ref + bias.double()).relu(), rtol=1e-4, atol=1e-5)
    # same result (bitwise) from the unsplit kernel's row order is not required, but run-to-run it is
    out2 = ops.spmm(gr.by_dst.rowptr, gr.by_dst.col, gw.w_dst, x.to(DEV), n, self_w=gw.self_w, bias=bias.to(DEV),
                    relu=True, split=gr.by_dst.split)
    assert torch.equal(out, out2)
    refT = torch.zeros(n, d, dtype=torch.float64).index_add_(0, e2[0], x.double()[e2[1]] * w.double()[:, None])
    outT = ops.spmm(gr.by_src.rowptr, gr.by_src.col, gw.w_src, x.to(DEV), n, self_w=gw.self_w, split=gr.by_src.split)
    torch.testing.assert_close(outT.cpu().double(), refT, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("ta,tb,m,n,k", [(True, False, 256, 256, 50_000), (True, False, 256, 768, 20_000),
                                         (False, False, 96, 40, 9_000), (False, True, 300, 64, 5_000)])
def test_gemm_split_k(ta, tb, m, n, k):
    from biomedkg_amd import ops
    torch.manual_seed(2)
    a = torch.randn((k, m) if ta else (m, k), device=DEV)
    b = torch.randn((n, k) if tb else (k, n), device=DEV)
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
    out = ops.gemm(a, b, trans_a=ta, trans_b=tb)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize("kind", ["dismult", "transe"])
def test_kge_backward_split_and_sorted_paths(kind):
    from biomedkg_amd import ops
    from oracle import ref_modules as R
    n, t, r, d = 2000, 120_000, 7, 128
    ei = _skewed_graph(n, t, hub_share=0.25, seed=3)
    torch.manual_seed(3)
    et = torch.randint(0, r, (t,))
    et[et == 4] = 5
    z, rel, gout = torch.randn(n, d) * 0.3, torch.randn(r, d) * 0.3, torch.randn(t)
    zo, ro = z.double().requires_grad_(True), rel.double().requires_grad_(True)
    fn = R.distmult_score if kind == "dismult" else R.transe_score
    so = fn(zo, ei, et, ro)
    so.backward(gout.double())
    zd, rd = z.to(DEV).requires_grad_(True), rel.to(DEV).requires_grad_(True)
    s = ops.kge_score(zd, rd, ei.to(DEV), et.to(DEV), ops.KGE_DISTMULT if kind == "dismult" else ops.KGE_TRANSE)
    torch.testing.assert_close(s.detach().cpu().double(), so.detach(), rtol=1e-4, atol=1e-4)
    s.backward(gout.to(DEV))
    if kind == "dismult":
        torch.testing.assert_close(zd.grad.cpu().double(), zo.grad, rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(rd.grad.cpu().double(), ro.grad, rtol=1e-4, atol=1e-3)
    else:
        # TransE's gradient is a sum of sign() terms: an fp32-vs-fp64 sign flip at u ~ 0 moves one term by
        # 2|g|, so compare in the Frobenius norm instead of elementwise
        for got, exp in ((zd.grad, zo.grad), (rd.grad, ro.grad)):
            err = (got.cpu().double() - exp).norm() / exp.norm()
            assert err < 2e-3, err
