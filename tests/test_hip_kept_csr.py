"""A ``dropout_edge`` view as CSRs of its own (``bkg_gcn_view`` / graph.KeptCSR): reference biomedkg/model/gcl.py:42-43,76
-- PyG ``dropout_edge`` hands GCNConv ``edge_index[:, keep]``.  The view's CSRs and weights against the oracle's gcn_norm
of the kept edges entry by entry, the propagates over them, and a whole augmented GGD / GRACE step against the same step
over the masked weights (the form every graph below ``KEPT_CSR_MIN_EDGES`` keeps).  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _hub_graph(n, e, seed=0):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    hub = e // 4
    ei[1, :hub] = 7                      # a target with ~e/4 in-edges, a source with as many out-edges: long rows
    ei[0, hub:2 * hub] = 11
    ei[:, -50:] = torch.arange(50).repeat(2, 1)                     # self loops already in the edge list
    return ei[:, torch.randperm(e, generator=g)]


@pytest.mark.parametrize("n,e,p_drop", [(1, 0, 0.4), (5, 3, 0.5), (7, 40, 1.0), (7, 40, 0.0), (1000, 50_000, 0.4), (130_000, 1_000_003, 0.4),
                                        (3, 70_000, 0.4)])
def test_the_views_csrs_are_gcn_norm_of_the_kept_edges_entry_by_entry(n, e, p_drop):
    """Both CSRs of the view: rows in order, the entries of a row in edge order (the graph's stable order), weights
    ``deg^-1/2[s] deg^-1/2[d]`` of the KEPT edges' degrees, existing self loops gone, the unit loop's weight ``1 / deg``."""
    from biomedkg_amd import graph as G
    from oracle import pyg_ops as P
    g = torch.Generator().manual_seed(n * 31 + e)
    ei = torch.randint(0, n, (2, e), generator=g)
    if n > 10:
        ei[1, : e // 3] = 0                                         # a long first row; the last rows stay (nearly) empty
        ei[:, e // 3:] %= max(n - 3, 1)
    keep = torch.rand(e, generator=g) >= p_drop
    gr = G.Graph(ei.to(DEV), n)
    gw = gr.gcn_view(keep.to(DEV))
    kept = keep & (ei[0] != ei[1])
    e2, w = P.gcn_norm(ei[:, keep], n)                              # (kept non-loop edges in order, then one loop per node)
    m = int(kept.sum())
    assert torch.equal(e2[:, :m], ei[:, kept])
    torch.testing.assert_close(gw.self_w.cpu(), w[m:], rtol=1e-6, atol=0)
    for view, row_of, col_of in ((gw.dst, 1, 0), (gw.src, 0, 1)):
        assert int(view.count) == m
        order = torch.argsort(e2[row_of, :m], stable=True)           # rows ascending, edge order inside a row
        rowptr = torch.zeros(n + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(torch.bincount(e2[row_of, :m], minlength=n), 0)
        assert torch.equal(view.rowptr.cpu().long(), rowptr)
        assert torch.equal(view.col[:m].cpu().long(), e2[col_of, :m][order])
        torch.testing.assert_close(view.w[:m].cpu(), w[:m][order], rtol=1e-6, atol=0)


@pytest.mark.parametrize("tight_parent", [False, True])
@pytest.mark.parametrize("d,bf16", [(256, False), (512, True), (128, True)])
def test_propagates_over_the_kept_csr_match_gcn_norm_of_the_kept_edges(monkeypatch, d, bf16, tight_parent):
    from biomedkg_amd import graph as G, ops
    from oracle import pyg_ops as P
    monkeypatch.setattr(G, "KEPT_CSR_MIN_EDGES", 0)
    n, e = 3000, 120_000
    ei = _hub_graph(n, e)
    torch.manual_seed(3)
    x = torch.randn(n, d)
    keep = torch.rand(e) >= 0.4
    gr = G.Graph(ei.to(DEV), n)
    if tight_parent:                     # the graph's tables know their counts: they bound the view's launches
        gr.by_dst.split.tighten(); gr.by_src.split.tighten()
    gw = gr.gcn_weights(keep.to(DEV))
    if tight_parent:
        for view, full in ((gw.dst, gr.by_dst), (gw.src, gr.by_src)):
            assert view.split.n_items_max == full.split.n_items_max + G.VIEW_GRID_MARK
            assert view.split.c.n_slots_max == full.split.n_slots_max
            t = view.split.totals.tolist()
            assert t[0] <= view.split.n_items_max and t[1] <= view.split.n_slots_max and t[2] <= view.split.n_long_max
            assert t[3] <= view.split.n_short_max
    assert isinstance(gw.dst, G.KeptCSR) and gw.dst.split is not None and gw.w_dst is gw.dst.w
    loops = int((ei[0] == ei[1])[keep].sum())
    assert int(gw.dst.count) == int(keep.sum()) - loops == int(gw.src.count)    # existing self loops carry weight 0
    xt = x.to(DEV).to(torch.bfloat16) if bf16 else x.to(DEV)
    xr = xt.double().cpu()
    e2, w = P.gcn_norm(ei[:, keep], n)
    tol = dict(rtol=1e-4, atol=1e-5)
    for side, a, b in (("dst", 1, 0), ("src", 0, 1)):
        c = getattr(gw, side)
        ref = torch.zeros(n, d, dtype=torch.float64).index_add_(0, e2[a], xr[e2[b]] * w.double()[:, None])
        for _ in range(4):                                          # (never tightened: same bounds at every launch)
            out = ops.spmm(c.rowptr, c.col, c.w, xt, n, self_w=gw.self_w, split=c.split, live=c.count)
        torch.testing.assert_close(out.cpu().double(), ref, **tol)
        assert not c.split.exact
        # the masked form of the same view: same sums
        full = gr.by_dst if side == "dst" else gr.by_src
        w_edge, _ = gr.gcn_norm_raw(keep.to(DEV))
        masked = ops.spmm(full.rowptr, full.col, full.permute_f32(w_edge), xt, n, self_w=gw.self_w, split=full.split)
        torch.testing.assert_close(out, masked, **tol)
    assert gr.gcn_weights(keep.to(DEV)) is not gw                   # another mask tensor: another view
    assert gr.gcn_weights() .dst is None


@pytest.fixture
def _mode(request):
    from biomedkg_amd import ops
    was = ops.get_matmul_precision()
    ops.set_matmul_precision(request.param)
    yield request.param
    ops.set_matmul_precision(was)


@pytest.mark.parametrize("_mode", ["bf16", "fp32"], indirect=True)
@pytest.mark.parametrize("kind", ["ggd", "grace"])
def test_an_augmented_step_over_kept_edges_equals_the_masked_step(monkeypatch, kind, _mode):
    """Loss and every parameter gradient of one augmented training step: the compacted view against the masked weights
    (same draws; the sums differ only in how long rows are cut into work items).  Both modes (bf16: the fused views with
    their tagged launches; exact fp32: the fp32 side-by-side layers / GRACE's one-view layers)."""
    from biomedkg_amd import graph as G, ops, rand as R
    from biomedkg_amd.gcl_module import GGDModule, GRACEModule
    n, e = 6000, 100_000
    ei = _hub_graph(n, e, seed=12).to(DEV)
    x = torch.randn(n, 768, generator=torch.Generator().manual_seed(12)).to(DEV)
    torch.manual_seed(0)
    cls = GGDModule if kind == "ggd" else GRACEModule
    m = cls(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(DEV).train()
    if kind == "ggd":
        m.model.p = 1.0                                             # always augment
    res = []
    for floor in (1 << 40, 0):
        monkeypatch.setattr(G, "KEPT_CSR_MIN_EDGES", floor)
        ei_run = ei.clone()                                         # (structures are cached on the tensor)
        m.zero_grad(set_to_none=True)
        torch.manual_seed(99)                                       # (the draws torch's generator makes: the same)
        timers, ops.KERNEL_TIMERS = ops.KERNEL_TIMERS, {}
        try:
            with R.use(R.DeviceRand(seed=5)):
                loss = m.calculate_loss(x, ei_run)
            loss.backward()
            tags = {t: ops.timer_bytes(rec) for t, rec in ops.KERNEL_TIMERS.items()}
        finally:
            ops.KERNEL_TIMERS = timers
        kept_tags = [t for t in tags if t.endswith("_kept")]
        if _mode == "bf16" or kind == "grace":                     # (the fp32 side-by-side layer's launches carry no tag)
            assert bool(kept_tags) == (floor == 0), tags
        res.append((float(loss.detach()), [p.grad.clone() for p in m.model.parameters()], tags))
    (la, ga, ta), (lb, gb, tb) = res
    assert abs(la - lb) <= (1e-4 if _mode == "bf16" else 2e-6) * abs(la), (la, lb)
    for a, b in zip(ga, gb):
        assert float((a - b).norm() / a.norm().clamp_min(1e-12)) < (5e-3 if _mode == "bf16" else 2e-5)
    # the kept launches are priced on the entries they walk: ~60 % of the gathered bytes of the masked launch
    for t in (t for t in tb if t.endswith("_kept")):
        assert 0.5 < tb[t] / ta[t[:-5]] < 0.75, (t, tb[t], ta[t[:-5]])
