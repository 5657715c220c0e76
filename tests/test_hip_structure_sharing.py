"""Structures that lend each other their sorts (round 4): the R-GCN structure's (relation, target) sort as the scorer's
relation order while a graph is young, the by-source sort shared by the GCN and the R-GCN structure of one ``edge_index``.
Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _graph(n, e, r, seed):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    et = torch.randint(0, r, (e,), generator=g)
    et[: e // 2] = 1                                   # a dominant relation, an empty one (r - 1 never drawn below)
    et[et == r - 1] = 0
    return ei.to(DEV), et.to(DEV)


def test_planted_relation_order_is_a_grouping_and_gives_way_to_the_stable_one():
    from biomedkg_amd import graph as G, ops
    n, e, r = 3000, 50_000, 7
    ei, et = _graph(n, e, r, 1)
    G.RelGraph(ei, et, n, r)
    counts = torch.bincount(et.cpu(), minlength=r)
    ptr_ref = torch.zeros(r + 1, dtype=torch.int64)
    ptr_ref[1:] = torch.cumsum(counts, 0)
    stable = torch.argsort(et.cpu(), stable=True)
    seen = []
    for ask in range(4):
        order, ptr = ops.relation_order_of(et, r)
        order, ptr = order.cpu().long(), ptr.cpu().long()
        assert torch.equal(ptr, ptr_ref)
        assert torch.equal(torch.sort(order).values, torch.arange(e))            # a permutation of the triplets
        assert torch.equal(et.cpu()[order], torch.sort(et.cpu()).values)         # grouped by relation
        seen.append(torch.equal(order, stable))
    assert seen == [False, False, True, True]          # two steps on the structure's sort, then the stable order
    # a structure-less id tensor sorts at once; an id tensor changed in place forgets what was planted
    et2 = et.clone()
    assert torch.equal(ops.relation_order_of(et2, r)[0].cpu().long(), stable)
    G.RelGraph(ei, et2, n, r)
    et2[:10] = 3
    order, _ = ops.relation_order_of(et2, r)
    assert torch.equal(et2.cpu()[order.cpu().long()], torch.sort(et2.cpu()).values)


def test_scorer_gives_the_same_loss_and_gradients_under_either_relation_order():
    from biomedkg_amd import graph as G, ops
    n, e, r, d = 3000, 60_000, 7, 128
    ei, et = _graph(n, e, r, 2)
    g = torch.Generator().manual_seed(3)
    z0 = torch.randn(n, d, generator=g).to(DEV)
    rel0 = torch.randn(r, d, generator=g).to(DEV)
    neg = torch.randint(0, n, (2, e), generator=g).to(DEV)
    res = []
    for planted in (False, True):
        et_run = et.clone()
        if planted:
            G.RelGraph(ei, et_run, n, r)
        z, rel = z0.clone().requires_grad_(True), rel0.clone().requires_grad_(True)
        loss, pos, _ = ops.kge_bce_loss(z, rel, ei, et_run, neg, et_run, ops.KGE_DISTMULT, reg=1e-2)
        loss.backward()
        res.append((float(loss.detach()), z.grad.clone(), rel.grad.clone(), pos.clone()))
    (la, za, ra, pa), (lb, zb, rb, pb) = res
    assert abs(la - lb) <= 1e-6 * abs(la)
    assert torch.equal(pa, pb)                         # scores come back in the caller's order either way
    torch.testing.assert_close(za, zb, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(ra, rb, rtol=1e-4, atol=1e-7)


def test_gcn_and_rgcn_structures_share_the_by_source_sort():
    from biomedkg_amd import graph as G
    n, e, r = 2000, 40_000, 5
    ei, et = _graph(n, e, r, 4)
    alone = G.Graph(ei.clone(), n).by_src
    g = G.graph_of(ei, n)
    rg = G.rel_graph_of(ei, et, n, r)
    assert rg.src_csr.perm is g.by_src.perm and rg.src_csr.rowptr is g.by_src.rowptr          # one sort, two readers
    assert torch.equal(g.by_src.perm, alone.perm) and torch.equal(g.by_src.rowptr, alone.rowptr)
    assert torch.equal(g.by_src.col, alone.col)
    # the R-GCN side keeps its own columns (segments, not targets) over the shared order
    order = torch.argsort(ei[0].cpu(), stable=True)
    assert torch.equal(rg.src_csr.perm.cpu().long(), order)
    # another row count (a partitioned structure's slot space) does not take the cached sort
    other = G.CSR(ei[0], ei[1], n + 7, shared=(ei, "sort_by_source"))
    assert other.rowptr.numel() == n + 8 and other.perm is not g.by_src.perm
    # in-place edits of the tensor invalidate it
    ei[0, 0] = (ei[0, 0] + 1) % n
    again = G.CSR(ei[0], ei[1], n, shared=(ei, "sort_by_source"))
    assert again.perm is not g.by_src.perm
    assert torch.equal(again.perm.cpu().long(), torch.argsort(ei[0].cpu(), stable=True))


def test_positives_of_a_graph_that_stays_are_scored_in_their_sorted_frame(monkeypatch):
    """From the step after the planted order is used up, the scorer walks a copy of the positives stored in (relation,
    target) order: same loss, same gradients, scores handed back in the caller's order."""
    from biomedkg_amd import graph as G, ops
    monkeypatch.setattr(ops, "_SORTED_MIN_TRIPLETS", 1000)
    monkeypatch.setattr(ops, "KGE_ONEPASS", False)      # (the two-pass form: DistMult training passes walk the positives once by default)
    n, e, r, d = 3000, 60_000, 7, 128
    ei, et = _graph(n, e, r, 5)
    g = torch.Generator().manual_seed(6)
    z0 = torch.randn(n, d, generator=g).to(DEV)
    rel0 = torch.randn(r, d, generator=g).to(DEV)
    neg = torch.randint(0, n, (2, e), generator=g).to(DEV)
    neg_t = torch.randint(0, r - 1, (e,), generator=g).to(DEV)
    G.RelGraph(ei, et, n, r)
    res, framed = [], []
    for step in range(4):
        z, rel = z0.clone().requires_grad_(True), rel0.clone().requires_grad_(True)
        loss, pos, _ = ops.kge_bce_loss(z, rel, ei, et, neg, neg_t, ops.KGE_DISTMULT, reg=1e-2)
        loss.backward()
        framed.append("sorted_triplets" in et._bkg_cache)
        res.append((float(loss.detach()), z.grad.clone(), rel.grad.clone(), pos.clone()))
    assert framed == [False, False, True, True]
    frame = et._bkg_cache["sorted_triplets"][1]
    inv = frame.inverse.long()
    assert torch.equal(frame.index[:, inv], ei) and torch.equal(frame.types[inv], et)
    keys = frame.types * n + frame.index[1]
    assert bool((keys[1:] >= keys[:-1]).all())                    # (relation, target) order
    l0, z0g, r0g, p0 = res[0]
    for l, zg, rg, p in res[1:]:
        assert abs(l - l0) <= 1e-6 * abs(l0)
        assert torch.equal(p, p0)                                  # the caller's order, the same scores
        torch.testing.assert_close(zg, z0g, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(rg, r0g, rtol=1e-4, atol=1e-7)
    # an id tensor edited in place drops the frame with everything else cached on it
    et[:5] = 2
    loss, pos, _ = ops.kge_bce_loss(z0.clone().requires_grad_(True), rel0, ei, et, neg, neg_t, ops.KGE_DISTMULT)
    ref = ops.kge_score(z0, rel0, ei, et, ops.KGE_DISTMULT).detach()        # (exact fp32 rows; the loss reads a bf16 table)
    assert float((pos - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    assert float((pos[:5] - p0[:5]).abs().max()) > 1e-3                   # ... and they are the edited triplets' scores
