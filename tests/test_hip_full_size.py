"""The hot path at BASELINE.json's full sizes (configs[1]: N = 130,000, E = 4,000,000, R = 30, d = 256).

The CPU oracle cannot finish these in seconds, so the checks are (a) size-independent properties -- sortedness /
permutation of the CSR build, checksums of the weights, linearity, view-swap symmetry and scale invariance of the
contrastive loss, membership of the sampled negatives -- and (b) for the floating-point kernels a plain-torch
fp64 restatement of the same op evaluated on the GPU in row chunks (tolerances written at each assert).
Needs an MI355X; every product call goes through the C-ABI.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
N, E, R, D = 130_000, 4_000_000, 30, 256


@pytest.fixture(scope="module")
def kg():
    from biomedkg_amd import synthetic
    g = synthetic.primekg_like(N, E, R, seed=7)
    return g["edge_index"].to(DEV), g["edge_type"].to(DEV)


def _x(seed, rows=N, d=D):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(rows, d, device=DEV, generator=g)


def test_csr_build_full_size(kg):
    from biomedkg_amd import graph as G
    ei, _ = kg
    gr = G.Graph(ei, N)
    for csr, key, other in ((gr.by_dst, ei[1], ei[0]), (gr.by_src, ei[0], ei[1])):
        rp, perm = csr.rowptr.long(), csr.perm.long()
        assert rp[0] == 0 and rp[-1] == E and bool((rp[1:] >= rp[:-1]).all())
        assert torch.equal(torch.sort(perm).values, torch.arange(E, device=DEV))            # a permutation
        k = key[perm]
        assert bool((k[1:] >= k[:-1]).all())                                                   # sorted by row
        assert bool((perm[1:] > perm[:-1])[k[1:] == k[:-1]].all())                             # stable
        assert torch.equal(csr.col.long(), other[perm])
        assert torch.equal(rp[1:] - rp[:-1], torch.bincount(key, minlength=N))                 # row lengths


def test_gcn_norm_full_size(kg):
    """PyG gcn_norm: self loops replaced by one unit loop per node, w = deg^-1/2[src] deg^-1/2[dst]."""
    from biomedkg_amd import graph as G
    ei, _ = kg
    gr = G.Graph(ei, N)
    gw = gr.gcn_weights()
    src, dst = ei
    real = src != dst
    deg = torch.bincount(dst[real], minlength=N).double() + 1.0
    dis = deg.pow(-0.5)
    w = torch.where(real, dis[src] * dis[dst], torch.zeros((), dtype=torch.float64, device=DEV))
    torch.testing.assert_close(gw.self_w.double(), dis * dis, rtol=1e-6, atol=0)
    torch.testing.assert_close(gw.w_dst.double(), w[gr.by_dst.perm.long()], rtol=1e-6, atol=0)
    torch.testing.assert_close(gw.w_src.double(), w[gr.by_src.perm.long()], rtol=1e-6, atol=0)
    # checksum of checksums: the normalised adjacency has the same total weight in both orders
    assert abs(float(gw.w_dst.double().sum() - gw.w_src.double().sum())) < 1e-6 * float(gw.w_dst.double().sum())


def test_dropout_edge_view_full_size(kg):
    """The view builder as the bench runs it (4 M edges: above the product's floor, no patching): both CSRs of a
    ``dropout_edge`` view are the kept non-loop edges in row order with PyG's gcn_norm weights of the KEPT graph, and the
    two-view propagate over them equals an fp64 restatement (row chunks) of ``A_hat_kept @ table`` with its epilogue."""
    from biomedkg_amd import graph as G, ops
    ei, _ = kg
    src, dst = ei
    gen = torch.Generator(device=DEV).manual_seed(11)
    keep = torch.rand(E, device=DEV, generator=gen) >= 0.4
    gr = G.graph_of(ei, N)
    for _ in range(4):
        gr.by_dst.split.use(); gr.by_src.split.use()            # (the graph's tables know their counts, as after a step)
    gw = gr.gcn_weights(keep)
    assert isinstance(gw.dst, G.KeptCSR) and isinstance(gw.src, G.KeptCSR)
    live = keep & (src != dst)
    m = int(live.sum())
    assert int(gw.dst.count) == m == int(gw.src.count)
    deg = torch.bincount(dst[live], minlength=N).double() + 1.0
    dis = deg.pow(-0.5)
    torch.testing.assert_close(gw.self_w.double(), dis * dis, rtol=1e-6, atol=0)
    for view, key, other in ((gw.dst, dst, src), (gw.src, src, dst)):
        rp = view.rowptr.long()
        assert rp[0] == 0 and rp[-1] == m and bool((rp[1:] >= rp[:-1]).all())
        assert torch.equal(rp[1:] - rp[:-1], torch.bincount(key[live], minlength=N))             # row lengths of the view
        rows = torch.repeat_interleave(torch.arange(N, device=DEV), rp[1:] - rp[:-1])
        col = view.col[:m].long()
        torch.testing.assert_close(view.w[:m].double(), dis[rows] * dis[col], rtol=1e-6, atol=0)
        # the same multiset of (row, col) pairs as the kept edges (checksums of both orders and of their product)
        assert int(rows.sum()) == int(key[live].sum()) and int(col.sum()) == int(other[live].sum())
        assert int((rows * 7 + col * 13).remainder(1_000_003).sum()) == int((key[live] * 7 + other[live] * 13).remainder(1_000_003).sum())
    # the two-view propagate (bf16 table, column halves by XCD, bias + ReLU epilogue) over the view
    ops.set_matmul_precision("bf16")
    try:
        table = _x(3, d=2 * D).to(torch.bfloat16)
        bias = _x(4, rows=1, d=2 * D)[0]
        c = gw.dst
        out = ops.spmm(c.rowptr, c.col, gw.w_dst, table, N, self_w=gw.self_w, bias=bias, relu=True, split=c.split, live=c.count)
        tf = table.double()
        es, ed, ew = src[live], dst[live], (dis[src] * dis[dst])[live]
        for lo in range(0, N, 26_000):
            hi = min(lo + 26_000, N)
            sel = (ed >= lo) & (ed < hi)
            ref = torch.zeros(hi - lo, 2 * D, dtype=torch.float64, device=DEV)
            ref.index_add_(0, ed[sel] - lo, tf[es[sel]] * ew[sel, None])
            ref += tf[lo:hi] * (dis[lo:hi] ** 2)[:, None] + bias.double()
            torch.testing.assert_close(out[lo:hi].double(), ref.relu(), rtol=1e-4, atol=1e-4)
    finally:
        ops.set_matmul_precision("fp32")


def test_spmm_full_size(kg):
    from biomedkg_amd import graph as G, ops
    ei, _ = kg
    gr = G.Graph(ei, N)
    gw = gr.gcn_weights()
    c = gr.by_dst
    x, y = _x(1), _x(2)
    out = ops.spmm(c.rowptr, c.col, gw.w_dst, x, N, self_w=gw.self_w, split=c.split)
    # fp64 restatement: scatter-add of the weighted source rows
    src, dst = ei
    w = gw.w_dst.double()
    ref = x.double() * gw.self_w.double()[:, None]
    perm = c.perm.long()
    for lo in range(0, E, 1_000_000):
        p = perm[lo:lo + 1_000_000]
        ref.index_add_(0, dst[p], x.double()[src[p]] * w[lo:lo + 1_000_000, None])
    # fp32 sums over rows of up to ~4e4 entries (chunk-ordered): 1e-4 of the row's scale
    torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-4)
    # row-sum checksum: a table of ones returns the row's total weight in every column
    ones = torch.ones(N, 16, device=DEV)
    rs = ops.spmm(c.rowptr, c.col, gw.w_dst, ones, N, self_w=gw.self_w, split=c.split)
    tot = torch.zeros(N, dtype=torch.float64, device=DEV).index_add_(0, dst[perm], w) + gw.self_w.double()
    torch.testing.assert_close(rs.double(), tot[:, None].expand(-1, 16), rtol=1e-5, atol=1e-6)
    # linearity
    mix = ops.spmm(c.rowptr, c.col, gw.w_dst, 0.5 * x - 2.0 * y, N, self_w=gw.self_w, split=c.split)
    oy = ops.spmm(c.rowptr, c.col, gw.w_dst, y, N, self_w=gw.self_w, split=c.split)
    torch.testing.assert_close(mix, 0.5 * out - 2.0 * oy, rtol=1e-4, atol=1e-4)
    # idempotence of the launch (no atomics on this path): bit-identical reruns
    assert torch.equal(out, ops.spmm(c.rowptr, c.col, gw.w_dst, x, N, self_w=gw.self_w, split=c.split))


def _rgcn_ref(x, weight, root, bias, ei, et):
    """fp64 RGCNConv(aggr='mean') by relation: mean over the in-edges of each (relation, target)."""
    out = x @ root + bias
    src, dst = ei
    for r in range(weight.size(0)):
        m = et == r
        s, t = src[m], dst[m]
        cnt = torch.bincount(t, minlength=x.size(0)).clamp_min(1).to(x.dtype)
        agg = torch.zeros_like(x).index_add_(0, t, x[s]) / cnt[:, None]
        out = out + agg @ weight[r]
    return out


def test_rgcn_layer_full_size(kg):
    from biomedkg_amd import graph as G, ops
    ei, et = kg
    rg = G.RelGraph(ei, et, N, R)
    torch.manual_seed(11)
    x = _x(3).requires_grad_()
    weight = (torch.randn(R, D, 64, device=DEV) / 16).requires_grad_()
    root = (torch.randn(D, 64, device=DEV) / 16).requires_grad_()
    bias = torch.randn(64, device=DEV).requires_grad_()
    dout = _x(4, d=64)
    out = ops.rgcn_conv(x, weight, root, bias, rg)
    out.backward(dout)
    xd, wd, rd, bd = (t.detach().double().requires_grad_() for t in (x, weight, root, bias))
    ref = _rgcn_ref(xd, wd, rd, bd, ei, et)
    ref.backward(dout.double())
    torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-4)
    for got, want, name in ((x.grad, xd.grad, "dx"), (weight.grad, wd.grad, "dW"), (root.grad, rd.grad, "dRoot"),
                            (bias.grad, bd.grad, "db")):
        err = float((got.double() - want).norm() / want.norm())
        assert err < 1e-5, (name, err)                      # Frobenius-relative, fp32 accumulation over up to 4e6 rows


def _kge_ref(kind, z, rel, ei, et, label, chunk=500_000):
    """fp64 scorer + BCE-with-logits in triplet chunks; gradients accumulate into z.grad / rel.grad."""
    total = ei.size(1)
    loss = 0.0
    zz = z / z.abs().sum(1, keepdim=True).clamp_min(1e-12) if kind == 1 else z
    scores = []
    for lo in range(0, total, chunk):
        h, t, r = ei[0, lo:lo + chunk], ei[1, lo:lo + chunk], et[lo:lo + chunk]
        if kind == 0:
            s = (zz[h] * rel[r] * zz[t]).sum(1)
        else:
            s = -(zz[h] + rel[r] - zz[t]).abs().sum(1)
        part = torch.nn.functional.binary_cross_entropy_with_logits(s, label[lo:lo + chunk], reduction="sum") / total
        part.backward(retain_graph=(kind == 1))
        loss += float(part.detach())
        scores.append(s.detach())
    return loss, torch.cat(scores)


@pytest.mark.parametrize("kind", [0, 1])
def test_kge_loss_full_size(kg, kind):
    """decode(pos) + decode(neg) + BCE (reference kge_module.py:110-129) at T = 2E triplets."""
    from biomedkg_amd import ops, sampling
    ei, et = kg
    torch.manual_seed(5)
    neg = sampling.negative_sampling(ei, N, E)
    nt = et[torch.randperm(E, device=DEV)]
    z = (_x(6) * (0.3 if kind == 0 else 1.0)).requires_grad_()
    rel = (torch.randn(R, D, device=DEV) * (0.3 if kind == 0 else 0.002)).requires_grad_()
    loss, pos_s, neg_s = ops.kge_bce_loss(z, rel, ei, et, neg, nt, kind)
    loss.backward()
    zd, rd = z.detach().double().requires_grad_(), rel.detach().double().requires_grad_()
    all_ei, all_et = torch.cat([ei, neg], 1), torch.cat([et, nt])
    label = torch.cat([torch.ones(E, dtype=torch.float64, device=DEV), torch.zeros(E, dtype=torch.float64, device=DEV)])
    ref_loss, ref_s = _kge_ref(kind, zd, rd, all_ei, all_et, label)
    assert abs(float(loss.detach()) - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
    torch.testing.assert_close(torch.cat([pos_s, neg_s]).double(), ref_s, rtol=1e-4, atol=1e-4 if kind == 0 else 1e-5)
    for got, want, name in ((z.grad, zd.grad, "dz"), (rel.grad, rd.grad, "drel")):
        err = float((got.double() - want).norm() / want.norm())
        assert err < (1e-5 if kind == 0 else 2e-3), (name, err)   # TransE: sign(0) ties of |.|' differ at fp32 roundoff


def test_kge_loss_full_size_in_the_sorted_frame(kg, monkeypatch):
    """The positives of a graph that stays are scored in their (relation, target) frame from the step after the R-GCN
    structure's planted order is used up (ops._SortedTriplets; 4 M triplets: above the product's floor, as shipped): the
    same loss, scores and gradients as the first call's plain form, against the fp64 restatement."""
    from biomedkg_amd import graph as G, ops, sampling
    # (the two-pass form of the positives: a DistMult training pass takes the one-pass walk by default since round 5 --
    # test_kge_loss_full_size_positives_in_one_pass below; TransE and gradient-free calls still go through the frame)
    monkeypatch.setattr(ops, "KGE_ONEPASS", False)
    ei0, et0 = kg
    ei, et = ei0.clone(), et0.clone()                          # (structures are cached on the tensors)
    G.rel_graph_of(ei, et, N, R)                               # what the encoder builds: plants the relation order
    torch.manual_seed(5)
    neg = sampling.negative_sampling(ei, N, E)
    nt = et[torch.randperm(E, device=DEV)]
    z0, rel0 = _x(6) * 0.3, torch.randn(R, D, device=DEV) * 0.3
    res = []
    for call in range(4):
        z, rel = z0.clone().requires_grad_(), rel0.clone().requires_grad_()
        loss, pos_s, neg_s = ops.kge_bce_loss(z, rel, ei, et, neg, nt, 0)
        loss.backward()
        res.append((float(loss.detach()), pos_s.clone(), z.grad.clone(), rel.grad.clone(), "sorted_triplets" in et._bkg_cache))
    assert [r[4] for r in res] == [False, False, True, True]      # two asks on the planted order, then the frame
    zd, rd = z0.double().requires_grad_(), rel0.double().requires_grad_()
    all_ei, all_et = torch.cat([ei, neg], 1), torch.cat([et, nt])
    label = torch.cat([torch.ones(E, dtype=torch.float64, device=DEV), torch.zeros(E, dtype=torch.float64, device=DEV)])
    ref_loss, ref_s = _kge_ref(0, zd, rd, all_ei, all_et, label)
    for loss, pos_s, dz, drel, _ in res:
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
        torch.testing.assert_close(pos_s.double(), ref_s[:E], rtol=1e-4, atol=1e-4)      # the caller's order
        assert float((dz.double() - zd.grad).norm() / zd.grad.norm()) < 1e-5
        assert float((drel.double() - rd.grad).norm() / rd.grad.norm()) < 1e-5


def test_kge_loss_full_size_positives_in_one_pass(kg):
    """The shipped DistMult training pass at full size (4 M positives in ONE walk over their sorted incidence lists,
    4 M fresh negatives through scorer + list walk) against the fp64 restatement: loss, the positives' scores in the
    caller's order, dz and d rel_emb; twice (the second call finds the lists cached on the graph's tensors)."""
    from biomedkg_amd import ops, sampling
    ei0, et0 = kg
    ei, et = ei0.clone(), et0.clone()
    torch.manual_seed(7)
    neg = sampling.negative_sampling(ei, N, E)
    nt = et[torch.randperm(E, device=DEV)]
    z0, rel0 = _x(8) * 0.3, torch.randn(R, D, device=DEV) * 0.3
    assert not ops.onepass_stable(ei, et)          # (first sighting of the pair; the calls below are its 2nd and 3rd)
    launches = []
    call = ops.L.call
    ops.L.call = lambda name, *a: (launches.append(name), call(name, *a))[1]
    try:
        res = []
        for _ in range(2):
            z, rel = z0.clone().requires_grad_(), rel0.clone().requires_grad_()
            loss, pos_s, neg_s = ops.kge_bce_loss(z, rel, ei, et, neg, nt, 0)
            loss.backward()
            res.append((float(loss.detach()), pos_s.clone(), neg_s.clone(), z.grad.clone(), rel.grad.clone()))
    finally:
        ops.L.call = call
    assert launches.count("bkg_kge_onepass_f32") == 2 and "onepass" in ei._bkg_cache
    zd, rd = z0.double().requires_grad_(), rel0.double().requires_grad_()
    all_ei, all_et = torch.cat([ei, neg], 1), torch.cat([et, nt])
    label = torch.cat([torch.ones(E, dtype=torch.float64, device=DEV), torch.zeros(E, dtype=torch.float64, device=DEV)])
    ref_loss, ref_s = _kge_ref(0, zd, rd, all_ei, all_et, label)
    for loss, pos_s, neg_s, dz, drel in res:
        assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss))
        torch.testing.assert_close(pos_s.double(), ref_s[:E], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(neg_s.double(), ref_s[E:], rtol=1e-4, atol=1e-4)
        assert float((dz.double() - zd.grad).norm() / zd.grad.norm()) < 1e-5
        assert float((drel.double() - rd.grad).norm() / rd.grad.norm()) < 1e-5
    assert torch.equal(res[0][3], res[1][3]) and torch.equal(res[0][4], res[1][4])      # no atomics anywhere: repeatable


def _ntxent_ref(h1, h2, tau, chunk=4096):
    """fp64 dual-branch InfoNCE with intra-view negatives in row chunks (never the [2N, 2N] matrix)."""
    n = h1.size(0)
    m = 2 * n
    z = torch.cat([h1, h2]).double().requires_grad_()
    zn = torch.nn.functional.normalize(z)
    total = 0.0
    for lo in range(0, m, chunk):
        idx = torch.arange(lo, min(lo + chunk, m), device=z.device)
        s = zn[idx] @ zn.t() / tau
        pos = s[torch.arange(idx.numel(), device=z.device), (idx + n) % m]
        s = s.index_put((torch.arange(idx.numel(), device=z.device), idx), torch.tensor(float("-inf"), device=z.device,
                                                                                         dtype=s.dtype))
        part = (torch.logsumexp(s, 1) - pos).sum() / m
        part.backward(retain_graph=True)
        total += float(part)
    return total, z.grad


def test_ntxent_full_size():
    """GRACE's loss over 2N = 260,000 rows: value and gradient vs the chunked fp64 restatement, plus the
    properties of the loss itself (view-swap symmetry, positive-scale invariance, gradient orthogonal to the row)."""
    from biomedkg_amd import ops
    h1 = _x(8).requires_grad_()
    h2 = (0.7 * h1.detach() + 0.7 * _x(9)).requires_grad_()
    loss = ops.ntxent_loss(h1, h2, 0.2)
    loss.backward()
    ref, gref = _ntxent_ref(h1.detach(), h2.detach(), 0.2)
    assert abs(float(loss) - ref) <= 1e-5 * abs(ref), (float(loss), ref)
    g = torch.cat([h1.grad, h2.grad]).double()
    err = float((g - gref).norm() / gref.norm())
    assert err < 1e-4, err
    with torch.no_grad():
        swapped = ops.ntxent_loss(h2.detach(), h1.detach(), 0.2)
        scaled = ops.ntxent_loss(3.0 * h1.detach(), 0.25 * h2.detach(), 0.2)
    assert abs(float(swapped) - float(loss)) <= 1e-6 * abs(float(loss))
    assert abs(float(scaled) - float(loss)) <= 1e-5 * abs(float(loss))
    radial = (h1.grad * h1.detach()).sum(1).abs().max() / (h1.grad.norm(dim=1) * h1.detach().norm(dim=1)).max()
    assert float(radial) < 1e-4


def test_negative_sampling_full_size(kg):
    """reference kge_module.py:93 (PyG negative_sampling): E distinct non-edges, relation ids a permutation."""
    from biomedkg_amd import sampling
    ei, _ = kg
    torch.manual_seed(12)
    neg = sampling.negative_sampling(ei, N, E)
    assert neg.shape == (2, E) and int(neg.min()) >= 0 and int(neg.max()) < N
    pos_keys = torch.unique(ei[0] * N + ei[1])
    nk = neg[0] * N + neg[1]
    assert not bool(torch.isin(nk, pos_keys).any())
    assert torch.unique(nk).numel() == E


def test_training_step_full_size_repeatable(kg):
    """Same seed, same graph -> the same loss (the forward path has no atomics); loss is finite and moves."""
    from biomedkg_amd import rand as R_, synthetic
    from biomedkg_amd.kge_module import KGEModule
    from types import SimpleNamespace
    ei, et = kg
    x = synthetic.lm_features(N, 2, 768, seed=1).to(DEV)
    losses = []
    for _ in range(2):
        torch.manual_seed(0)
        m = KGEModule(encoder_name="rgcn", decoder_name="dismult", in_dim=768, hidden_dim=256, out_dim=256,
                      num_hidden_layers=2, num_relation=R, num_heads=1, scheduler_type="cosine", learning_rate=1e-3,
                      warm_up_ratio=0.03, fuse_method="attention", neg_ratio=1, node_init_method="lm").to(DEV)
        batch = SimpleNamespace(x=x, edge_index=ei, edge_type=et)
        with R_.use(R_.DeviceRand(seed=3)):
            loss = m.training_step(batch, 0)
        loss.backward()
        assert torch.isfinite(loss)
        losses.append(float(loss))
    assert losses[0] == losses[1]


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_deterministic_mode_every_gradient_repeats_bitwise(kg, precision):
    """reference train_gcl.py:66 / train_kge.py:46 pass deterministic=True: with ops.set_deterministic(True) the split-K
    weight-gradient products sum their K chunks in a fixed order, and a full-size GGD + KGE step repeats bit for bit in
    its losses AND in every parameter gradient (without it the float atomics reorder the sums); the deterministic
    results equal the atomic ones up to that reordering."""
    from types import SimpleNamespace
    from biomedkg_amd import ops, rand as R_
    from biomedkg_amd.gcl_module import GGDModule
    from biomedkg_amd.kge_module import KGEModule
    ei, et = kg
    bk = SimpleNamespace(x=_x(21), edge_index=ei, edge_type=et)
    bg = SimpleNamespace(x=_x(22, d=768), edge_index=ei)

    def run():
        torch.manual_seed(3)
        kge = KGEModule("rgcn", "dismult", D, D, D, 2, R, 2, "cosine", 1e-3, 0.2, "none", 1, "random").to(DEV).train()
        gcl = GGDModule(in_dim=768, hidden_dim=D, out_dim=D, num_hidden_layers=2, fuse_method="none").to(DEV).train()
        out = []
        for mod, batch in ((gcl, bg), (kge, bk)):
            with R_.use(R_.DeviceRand(seed=9)):
                loss = mod.training_step(batch)
            loss.backward()
            out.append((loss.detach().clone(), [p.grad.clone() for p in mod.model.parameters()]))
        return out

    ops.set_matmul_precision(precision)
    try:
        ops.set_deterministic(True)
        a, b = run(), run()
        ops.set_deterministic(False)
        c = run()
    finally:
        ops.set_deterministic(False)
        ops.set_matmul_precision("fp32")
    for (la, ga), (lb, gb), (lc, gc) in zip(a, b, c):
        assert torch.equal(la, lb)
        assert all(torch.equal(x, y) for x, y in zip(ga, gb))
        assert torch.equal(la, lc)                                   # the forward never had atomics
        for x, y in zip(ga, gc):
            assert float((x - y).norm() / y.norm().clamp_min(1e-20)) < 1e-5


def test_kge_step_at_the_reference_default_neg_ratio_10_full_size():
    """The reference's DEFAULT KGE workload (configs/kge.yaml:14 ``neg_ratio: 10``, kge_module.py:86-102) at the bench
    size: N = 130,000, E = 4,000,000, T = 44 M triplets in one step, in the bf16 mode bench.py times -- the 40 M negatives
    are distinct, loop-free non-edges with the positives' relation multiset, every score equals a chunked fp64 DistMult
    on the same embeddings, and the loss equals the chunked fp64 BCE + L2 restatement."""
    from types import SimpleNamespace
    from biomedkg_amd import ops, rand as R, synthetic
    from biomedkg_amd.kge_module import KGEModule
    n, e, r, h = 130_000, 4_000_000, 30, 256
    prev = ops.get_matmul_precision()
    ops.set_matmul_precision("bf16")
    try:
        g = synthetic.primekg_like(n, e, r, seed=42)
        ei, et = g["edge_index"].to(DEV), g["edge_type"].to(DEV)
        e, r = ei.size(1), g["num_relations"]
        torch.manual_seed(1234)
        x = torch.randn(n, h, device=DEV)
        m = KGEModule("rgcn", "dismult", h, h, h, 2, r, 2, "cosine", 1e-3, 0.2, "none", 10, "random").to(DEV).train()
        drawn = {}
        orig = m.sample_neg_edges

        def spy(edge_index, edge_type):
            drawn["neg"] = orig(edge_index, edge_type)
            return drawn["neg"]
        m.sample_neg_edges = spy
        batch = SimpleNamespace(x=x, edge_index=ei, edge_type=et)
        with R.use(R.DeviceRand(seed=6)):
            loss = m.training_step(batch, 0)
        loss.backward()
        assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.model.parameters())
        neg_ei, neg_et = drawn["neg"]
        k = 10 * e
        assert neg_ei.shape == (2, k) and neg_et.shape == (k,)
        keys = neg_ei[0] * n + neg_ei[1]
        assert not bool((neg_ei[0] == neg_ei[1]).any())
        assert torch.unique(keys).numel() == k
        assert not bool(torch.isin(keys, ei[0] * n + ei[1]).any())
        assert torch.equal(torch.bincount(neg_et, minlength=r), 10 * torch.bincount(et, minlength=r))   # edge_type.repeat(10), permuted
        z, pos_pred, neg_pred = m._last_scores
        rel = m.model.decoder.rel_emb.detach().double()
        zz = z.detach().bfloat16().double()          # bf16 mode: the scorer reads the embedding rows as a bf16 table (ops.py
        total, acc = e + k, 0.0                      # _KgeLoss); products and sums in fp32, the L2 term on the fp32 rows
        for tri, types, label, got in ((ei, et, 1.0, pos_pred), (neg_ei, neg_et, 0.0, neg_pred)):
            for lo in range(0, tri.size(1), 2_000_000):
                hh, tt, rr = tri[0, lo:lo + 2_000_000], tri[1, lo:lo + 2_000_000], types[lo:lo + 2_000_000]
                s = (zz[hh] * rel[rr] * zz[tt]).sum(1)
                torch.testing.assert_close(got[lo:lo + 2_000_000].double(), s, rtol=1e-4, atol=1e-4)
                acc += float(torch.nn.functional.binary_cross_entropy_with_logits(s, torch.full_like(s, label), reduction="sum"))
        ref = acc / total + 1e-2 * float(z.detach().double().pow(2).mean() + rel.pow(2).mean())
        assert abs(float(loss) - ref) <= 1e-5 * max(1.0, abs(ref)), (float(loss), ref)
    finally:
        ops.set_matmul_precision(prev)
