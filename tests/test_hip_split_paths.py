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


@pytest.mark.parametrize("tight", [True, False])
@pytest.mark.parametrize("mix", ["mixed", "no_short_rows", "only_short_rows", "empty_rows_at_the_end"])
@pytest.mark.parametrize("d,dtype", [(256, torch.bfloat16), (512, torch.bfloat16), (128, torch.float32), (96, torch.float32)])
def test_spmm_row_classes(mix, tight, d, dtype):
    """The three row classes of the split table (csrc/rowsplit.hip, spmm.hip): rows of <= 8 entries finished several per
    wave from the descriptor table, rows of 9..256 entries one wave each, longer rows in slices + a fix-up -- every mix of
    them, with the launch bounds cut to the table's counts (``tight``) and with the array-size bounds a captured step
    keeps (whole waves past the lists), epilogue included, against an fp64 scatter-add."""
    from biomedkg_amd import graph as G, ops
    gen = torch.Generator(device=DEV).manual_seed(17)
    rows, n_table = 40_000, 50_000
    lens = torch.randint(0, 9, (rows,), device=DEV, generator=gen)
    if mix == "mixed":
        pick = torch.rand(rows, device=DEV, generator=gen)
        lens[pick < 0.06] = torch.randint(9, 60, (rows,), device=DEV, generator=gen)[pick < 0.06]
        lens[:4] = torch.tensor([900, 0, 1, 300], device=DEV)
    elif mix == "no_short_rows":
        lens = torch.randint(9, 20, (rows,), device=DEV, generator=gen)
        lens[5] = 700
    elif mix == "only_short_rows":
        lens = lens.clamp_min(1)
    else:
        lens[-300:] = 0
    rowptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), torch.cumsum(lens, 0)]).to(torch.int32)
    nnz = int(rowptr[-1])
    assert nnz >= G.SPLIT_MIN_NNZ
    col = torch.randint(0, n_table, (nnz,), device=DEV, dtype=torch.int32, generator=gen)
    w = torch.rand(nnz, device=DEV, generator=gen)
    x = torch.randn(n_table, d, device=DEV, generator=gen).to(dtype)
    bias = torch.randn(d, device=DEV, generator=gen)
    was = (G.TIGHT_GRIDS, G.TIGHT_AFTER)
    G.TIGHT_GRIDS, G.TIGHT_AFTER = tight, 1          # (the host read normally waits for a table's 3rd launch: RowSplit.use)
    try:
        sp = G.RowSplit(rowptr, rows, nnz, pack=True, col=col)
        assert not sp.exact
        _row_classes_body(sp, tight, lens, rows, d, dtype, rowptr, col, w, x, bias, nnz)
    finally:
        G.TIGHT_GRIDS, G.TIGHT_AFTER = was


def _row_classes_body(sp, tight, lens, rows, d, dtype, rowptr, col, w, x, bias, nnz):
    from biomedkg_amd import ops
    sp.use()
    assert sp.exact == tight
    n_items, _, n_long, n_short = sp.totals.tolist()
    assert n_short == int((lens <= 8).sum()) and n_long == int((lens > 256).sum())
    row_of = torch.repeat_interleave(torch.arange(rows, device=DEV), lens.long())
    ref = torch.zeros(rows, d, dtype=torch.float64, device=DEV)
    ref.index_add_(0, row_of, x.double()[col.long()] * w.double()[:, None])
    out = ops.spmm(rowptr, col, w, x, rows, split=sp, bias=bias, relu=True)
    torch.testing.assert_close(out.double(), (ref + bias.double()).relu(), rtol=1e-4, atol=1e-4)     # (fp32 sums of up to 900 terms)
    plain = ops.spmm(rowptr, col, None, x, rows, split=sp, out_dtype=torch.bfloat16 if dtype == torch.bfloat16 else torch.float32)
    ref1 = torch.zeros(rows, d, dtype=torch.float64, device=DEV).index_add_(0, row_of, x.double()[col.long()])
    torch.testing.assert_close(plain.double(), ref1, rtol=1e-2 if dtype == torch.bfloat16 else 1e-5, atol=1e-2 if dtype == torch.bfloat16 else 1e-5)
    assert torch.equal(out, ops.spmm(rowptr, col, w, x, rows, split=sp, bias=bias, relu=True))      # run to run
