"""Edge cases of the hot path on the HIP side, each against the CPU oracle: empty and ragged inputs,
isolated nodes, duplicate edges and duplicate self loops, sizes that are not multiples of any tile,
embedding widths that take the generic kernels, an all-empty relation set, zero negatives.  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = dict(rtol=1e-4, atol=1e-5)


def _gcn_both(x, ei, w, b):
    from biomedkg_amd import graph as G, ops
    from oracle import pyg_ops as P
    n = x.size(0)
    xo, wo, bo = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = P.gcn_conv(xo, ei, wo, bo)
    ref.sum().backward()
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    gr = G.Graph(ei.to(DEV), n)
    out = ops.gcn_conv(xd, wd, bd, gr, gr.gcn_weights())
    out.sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), **TOL)
    for got, exp in ((xd, xo), (wd, wo), (bd, bo)):
        torch.testing.assert_close(got.grad.cpu(), exp.grad, **TOL)


def test_gcn_empty_edge_set_and_single_node():
    torch.manual_seed(0)
    _gcn_both(torch.randn(7, 12), torch.zeros(2, 0, dtype=torch.long), torch.randn(5, 12), torch.randn(5))
    _gcn_both(torch.randn(1, 8), torch.zeros(2, 0, dtype=torch.long), torch.randn(4, 8), torch.randn(4))
    _gcn_both(torch.randn(1, 8), torch.tensor([[0, 0], [0, 0]]), torch.randn(4, 8), torch.randn(4))   # only self loops


def test_gcn_duplicates_isolated_and_odd_sizes():
    torch.manual_seed(1)
    n = 67                                   # not a multiple of 4 / 64
    ei = torch.randint(0, 40, (2, 333))      # nodes 40..66 isolated
    ei = torch.cat([ei, ei[:, :50], torch.tensor([[3, 3, 9], [3, 3, 9]])], dim=1)   # duplicate edges + duplicate loops
    _gcn_both(torch.randn(n, 19), ei, torch.randn(13, 19), torch.randn(13))         # odd widths: generic spmm path


@pytest.mark.parametrize("n,e,r,din,dout", [(50, 0, 3, 8, 8), (33, 200, 1, 20, 12), (40, 300, 6, 16, 24)])
def test_rgcn_edge_cases(n, e, r, din, dout):
    from biomedkg_amd import graph as G, ops
    from oracle import pyg_ops as P
    torch.manual_seed(2)
    ei = torch.randint(0, max(n - 5, 1), (2, e))
    et = torch.randint(0, r, (e,))
    if r > 2:
        et[et == 1] = 0                      # an empty relation in the middle
    x, w = torch.randn(n, din), torch.randn(r, din, dout) * 0.3
    root, b = torch.randn(din, dout) * 0.3, torch.randn(dout)
    xo, wo, ro, bo = (t.clone().requires_grad_(True) for t in (x, w, root, b))
    ref = P.rgcn_conv(xo, ei, et, wo, ro, bo)
    ref.pow(2).sum().backward()
    xd, wd, rd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, root, b))
    rg = G.RelGraph(ei.to(DEV), et.to(DEV), n, r)
    out = ops.rgcn_conv(xd, wd, rd, bd, rg)
    out.pow(2).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), **TOL)
    for got, exp in ((xd, xo), (wd, wo), (rd, ro), (bd, bo)):
        torch.testing.assert_close(got.grad.cpu(), exp.grad, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("kind", ["dismult", "transe"])
def test_scorer_empty_and_repeated_triplets(kind):
    from biomedkg_amd import ops
    from oracle import ref_modules as R
    torch.manual_seed(3)
    z, rel = torch.randn(9, 16), torch.randn(2, 16)
    k = ops.KGE_DISTMULT if kind == "dismult" else ops.KGE_TRANSE
    s = ops.kge_score(z.to(DEV), rel.to(DEV), torch.zeros(2, 0, dtype=torch.long, device=DEV),
                      torch.zeros(0, dtype=torch.long, device=DEV), k)
    assert s.shape == (0,)
    ei = torch.tensor([[0, 0, 0, 4, 4], [0, 0, 1, 4, 2]])     # h == t pairs and exact repeats
    et = torch.tensor([0, 0, 1, 1, 0])
    zo, ro = z.clone().requires_grad_(True), rel.clone().requires_grad_(True)
    fn = R.distmult_score if kind == "dismult" else R.transe_score
    so = fn(zo, ei, et, ro)
    so.sum().backward()
    zd, rd = z.to(DEV).requires_grad_(True), rel.to(DEV).requires_grad_(True)
    sd = ops.kge_score(zd, rd, ei.to(DEV), et.to(DEV), k)
    sd.sum().backward()
    torch.testing.assert_close(sd.detach().cpu(), so.detach(), **TOL)
    torch.testing.assert_close(zd.grad.cpu(), zo.grad, **TOL)
    torch.testing.assert_close(rd.grad.cpu(), ro.grad, **TOL)


def test_ntxent_ragged_sizes():
    from biomedkg_amd import ops
    from oracle import pygcl_ops as G
    for n, h in ((1, 4), (3, 5), (65, 33), (130, 96)):
        torch.manual_seed(n)
        a, b = torch.randn(n, h), torch.randn(n, h)
        ao, bo = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        if n == 1:
            continue   # a single pair has no negatives: log(0) in the reference as well
        ref = G.dual_branch_l2l_intraview(ao, bo, 0.2)
        ref.backward()
        ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        loss = ops.ntxent_loss(ad, bd, 0.2)
        loss.backward()
        torch.testing.assert_close(loss.detach().cpu(), ref.detach(), rtol=0, atol=1e-5)
        torch.testing.assert_close(ad.grad.cpu(), ao.grad, rtol=1e-3, atol=1e-6)


def test_kge_module_step_with_saturated_graph_returns_no_negatives():
    """negative_sampling on a complete graph yields zero negatives; the step must still run."""
    from biomedkg_amd.kge_module import KGEModule
    n = 6
    ei = torch.cartesian_prod(torch.arange(n), torch.arange(n)).t().contiguous()
    mod = KGEModule("rgcn", "dismult", 8, 8, 8, 1, 2, 2, "cosine", 1e-3, 0.2, "none", 1, "random").to(DEV).train()

    class B:
        pass
    b = B()
    b.x, b.edge_index, b.edge_type = torch.randn(n, 8, device=DEV), ei.to(DEV), torch.randint(0, 2, (n * n,), device=DEV)
    neg_ei, neg_et = mod.sample_neg_edges(b.edge_index, b.edge_type)
    assert neg_ei.shape == (2, 0)
    z = mod(b.x, b.edge_index, b.edge_type)
    assert torch.isfinite(z).all()


def test_redaf_training_dropout_draws_a_fresh_mask_every_call():
    """reference fusion.py:44,84: nn.Dropout(0.1) draws a new mask per forward.  Under the default DeviceRand the
    in-kernel Philox stream is keyed by the provider's per-call counter (a constant key would zero the same 10 % of
    positions on every training step)."""
    from biomedkg_amd import rand as R
    from biomedkg_amd.utils.fusion import ReDAF
    torch.manual_seed(0)
    f = ReDAF(64).to(DEV).train()
    with torch.no_grad():
        f.transform_layer.weight.copy_(torch.eye(64))     # t = relu(x): positive x stays > 0 unless dropped
        f.transform_layer.bias.zero_()
    x = torch.rand(512, 2, 64, device=DEV) + 0.5
    with R.use(R.DeviceRand(seed=3)):
        a, b = f(x), f(x)
    with R.use(R.DeviceRand(seed=3)):
        a2 = f(x)
    assert torch.equal(a, a2)                               # a pure function of (seed, call number)
    assert not torch.equal(a, b)                            # ... and the second call draws differently
    # per (row, modality, feature) drop rate ~ 0.1: the mean of two modalities loses one of them w.p. ~0.18
    f.eval()
    full = f(x)
    partial = ((a < 0.75 * full) & (full > 0)).float().mean()
    assert 0.12 < float(partial) < 0.25, float(partial)


def test_dropout_without_relu_gates_by_the_mask_not_by_the_output_sign():
    """A kept element whose pre-activation is exactly 0 still passes its gradient: GCNConv with
    relu=False and an explicit keep mask against the oracle; the in-kernel stream is refused for this shape."""
    from biomedkg_amd import graph as G, ops
    from oracle import pyg_ops as P
    torch.manual_seed(5)
    n, d_in, d_out = 50, 8, 6
    ei = torch.randint(0, n, (2, 200))
    x, w = torch.randn(n, d_in), torch.randn(d_out, d_in)
    x[:10] = 0.0                                            # rows whose neighbourhoods may sum to exactly 0
    b = torch.zeros(d_out)
    keep = torch.rand(n, d_out) >= 0.3
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = P.gcn_conv(xo, ei, wo, b) * keep / 0.7
    (ref * torch.arange(1, d_out + 1)).sum().backward()
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    gr = G.Graph(ei.to(DEV), n)
    out = ops.gcn_conv(xd, wd, b.to(DEV), gr, gr.gcn_weights(), relu=False, drop=ops.Dropout(0.3, keep=keep.to(DEV)))
    (out * torch.arange(1, d_out + 1, device=DEV)).sum().backward()
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), **TOL)
    torch.testing.assert_close(xd.grad.cpu(), xo.grad, **TOL)
    torch.testing.assert_close(wd.grad.cpu(), wo.grad, **TOL)
    with pytest.raises(NotImplementedError):
        ops.gcn_conv(xd, wd, b.to(DEV), gr, gr.gcn_weights(), relu=False, drop=ops.Dropout(0.3, seed=1, stream_id=1))


def test_mismatched_triplet_arrays_are_refused_not_dereferenced():
    """negative_sampling may come back with fewer pairs than neg_ratio * E (a 2-node batch has one free cell): the
    reference then dies on a broadcast error in decode; here the scorer would index one array by the other."""
    import pytest
    from biomedkg_amd import ops
    z = torch.randn(5, 64, device="cuda", requires_grad=True)
    rel = torch.randn(10, 64, device="cuda", requires_grad=True)
    pos, pt = torch.tensor([[0], [1]], device="cuda"), torch.tensor([2], device="cuda")
    neg, nt = torch.tensor([[1], [0]], device="cuda"), torch.full((5,), 2, device="cuda")
    with pytest.raises(ValueError):
        ops.kge_bce_loss(z, rel, pos, pt, neg, nt, ops.KGE_DISTMULT)
    with pytest.raises(ValueError):
        ops.kge_score(z, rel, neg, nt, ops.KGE_DISTMULT)
    loss, _, _ = ops.kge_bce_loss(z, rel, pos, pt, neg, nt[:1], ops.KGE_DISTMULT, reg=1e-2)
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(z.grad).all()


def test_device_rand_perm_is_a_fresh_permutation_per_call():
    """``DeviceRand.perm`` (GGD / DGI corruption, reference gcl.py:22-24,70-71): a keyed bijection of the index range, one
    launch, new on every call, torch.randperm on the CPU."""
    from biomedkg_amd import rand as R
    r = R.DeviceRand(seed=7)
    seen = []
    for n in (1, 2, 1000, 130_001):
        p = r.perm(n, DEV)
        assert p.dtype == torch.int64 and p.device.type == "cuda"
        assert torch.equal(torch.sort(p).values, torch.arange(n, device=DEV))
        q = r.perm(n, DEV)
        if n >= 1000:
            assert not torch.equal(p, q)
            assert int((p == torch.arange(n, device=DEV)).sum()) < 20        # nothing like the identity
            seen.append(p)
    again = R.DeviceRand(seed=7)                                             # same seed, same call number: same draw
    for n in (1, 1, 2, 2):
        again.perm(n, DEV)
    assert again.perm(1000, DEV).equal(seen[0])
    assert r.perm(5, "cpu").device.type == "cpu"


@pytest.mark.parametrize("n", [1, 7, 300, 70_001])
def test_ggd_bce_equals_torch(n):
    """``ops.ggd_bce`` (csrc/gcl_loss.hip) against ``binary_cross_entropy_with_logits(cat(pos, neg), cat(1, 0))`` of
    ``GGDModule.calculate_loss`` (reference gcl_module.py:229-234): value and both gradients, on the strided columns of a
    score table, large logits included; the masked form against the same loss over the leading rows."""
    import torch.nn.functional as F
    from biomedkg_amd import ops
    g = torch.Generator(device=DEV).manual_seed(n)
    s = (torch.randn(n, 2, device=DEV, generator=g) * 6.0).requires_grad_()
    loss = ops.ggd_bce(s[:, 0], s[:, 1])
    (3.0 * loss).backward()
    s2 = s.detach().clone().requires_grad_()
    ref = F.binary_cross_entropy_with_logits(torch.cat([s2[:, 0], s2[:, 1]]),
                                             torch.cat([torch.ones(n, device=DEV), torch.zeros(n, device=DEV)]))
    (3.0 * ref).backward()
    torch.testing.assert_close(loss, ref, rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(s.grad, s2.grad, rtol=1e-5, atol=1e-9)
    assert torch.equal(ops.ggd_bce(s[:, 0], s[:, 1]), loss)                      # fixed-order sums
    for live in sorted({1, max(n // 3, 1), n}):
        n_dev = torch.tensor([live], dtype=torch.int64, device=DEV)
        s3 = s.detach().clone().requires_grad_()
        lm = ops.ggd_bce(s3[:, 0], s3[:, 1], n_dev)
        lm.backward()
        s4 = s.detach().clone().requires_grad_()
        refm = F.binary_cross_entropy_with_logits(torch.cat([s4[:live, 0], s4[:live, 1]]),
                                                  torch.cat([torch.ones(live, device=DEV), torch.zeros(live, device=DEV)]))
        refm.backward()
        torch.testing.assert_close(lm, refm, rtol=2e-6, atol=1e-7)
        torch.testing.assert_close(s3.grad, s4.grad, rtol=1e-5, atol=1e-9)
        assert not bool(s3.grad[live:].any())
