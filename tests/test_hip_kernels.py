"""HIP kernels (through the C-ABI) vs. the CPU oracle / fp64 algebra.  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _graph(n, e, seed, self_loops=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    if self_loops and e >= 3:
        ei[:, 0] = torch.tensor([3, 3]); ei[:, 1] = torch.tensor([3, 3]); ei[:, 2] = torch.tensor([5, 5])
    return ei


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("m,n,k", [(77, 45, 33), (256, 128, 768), (1000, 16, 24), (5, 300, 130), (640, 256, 256)])
def test_gemm_f32(ta, tb, m, n, k):
    from biomedkg_amd import ops
    torch.manual_seed(0)
    a = torch.randn((k, m) if ta else (m, k), device=DEV)
    b = torch.randn((n, k) if tb else (k, n), device=DEV)
    ref = ((a.t() if ta else a).double() @ (b.t() if tb else b).double())
    out = ops.gemm(a, b, trans_a=ta, trans_b=tb)
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-4)


def test_gemm_epilogues():
    from biomedkg_amd import ops
    torch.manual_seed(1)
    m, n, k = 300, 70, 50
    a, b = torch.randn(m, k, device=DEV), torch.randn(n, k, device=DEV)
    bias, c0, gate = torch.randn(n, device=DEV), torch.randn(m, n, device=DEV), torch.randn(m, n, device=DEV)
    pre = 0.5 * (a.double() @ b.double().t()) + bias.double()
    torch.testing.assert_close(ops.gemm(a, b, trans_b=True, alpha=0.5, bias=bias, act=ops.ACT_RELU).double(),
                               pre.relu(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ops.gemm(a, b, trans_b=True, alpha=0.5, bias=bias, act=ops.ACT_ELU).double(),
                               torch.nn.functional.elu(pre), rtol=1e-5, atol=1e-5)
    out = c0.clone()
    ops.gemm(a, b, trans_b=True, alpha=0.5, bias=bias, beta=2.0, out=out)
    torch.testing.assert_close(out.double(), pre + 2.0 * c0.double(), rtol=1e-5, atol=1e-5)
    out = ops.gemm(a, b, trans_b=True, alpha=0.5, bias=bias, gate=gate, gate_scale=1.25)
    torch.testing.assert_close(out.double(), torch.where(gate > 0, pre * 1.25, 0.0), rtol=1e-5, atol=1e-5)


def test_gemm_grouped():
    from biomedkg_amd import ops
    torch.manual_seed(2)
    sizes = [0, 130, 1, 0, 257, 64]
    gptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device=DEV)
    s, k, n, g = sum(sizes), 40, 24, len(sizes)
    a, w = torch.randn(s, k, device=DEV), torch.randn(g, k, n, device=DEV)
    y = ops.gemm_grouped_rows(a, w, gptr)
    dy = torch.randn(s, n, device=DEV)
    dh = ops.gemm_grouped_rows(dy, w, gptr, trans_b=True)
    dw = ops.gemm_grouped_k(a, dy, gptr, g)
    o = 0
    for i, sz in enumerate(sizes):
        sl = slice(o, o + sz)
        torch.testing.assert_close(y[sl].double(), a[sl].double() @ w[i].double(), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(dh[sl].double(), dy[sl].double() @ w[i].double().t(), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(dw[i].double(), a[sl].double().t() @ dy[sl].double(), rtol=1e-5, atol=1e-4)
        o += sz


@pytest.mark.parametrize("d", [16, 24, 64, 128, 256, 768])
def test_spmm_and_gcn_norm(d):
    from biomedkg_amd import graph as G, ops
    from oracle import pyg_ops as P
    n, e = 200, 3000
    ei = _graph(n, e, 3)
    torch.manual_seed(3)
    x = torch.randn(n, d)
    keep = torch.rand(e) >= 0.4
    gr = G.Graph(ei.to(DEV), n)
    for kp in (None, keep):
        gw = gr.gcn_weights(None if kp is None else kp.to(DEV))
        ei_k = ei if kp is None else ei[:, kp]
        e2, w = P.gcn_norm(ei_k, n)
        ref = torch.zeros(n, d, dtype=torch.float64).index_add_(0, e2[1], x.double()[e2[0]] * w.double()[:, None])
        out = ops.spmm(gr.by_dst.rowptr, gr.by_dst.col, gw.w_dst, x.to(DEV), n, self_w=gw.self_w)
        torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-5)
        refT = torch.zeros(n, d, dtype=torch.float64).index_add_(0, e2[0], x.double()[e2[1]] * w.double()[:, None])
        outT = ops.spmm(gr.by_src.rowptr, gr.by_src.col, gw.w_src, x.to(DEV), n, self_w=gw.self_w)
        torch.testing.assert_close(outT.cpu().double(), refT, rtol=1e-5, atol=1e-5)


def test_spmm_epilogue_and_philox():
    from biomedkg_amd import graph as G, ops
    n, e, d = 300, 2000, 128
    ei = _graph(n, e, 4)
    torch.manual_seed(4)
    x, bias, add = torch.randn(n, d, device=DEV), torch.randn(d, device=DEV), torch.randn(n, d, device=DEV)
    keep = (torch.rand(n, d, device=DEV) >= 0.2)
    gr = G.Graph(ei.to(DEV), n)
    c = gr.by_dst
    base = ops.spmm(c.rowptr, c.col, None, x, n)
    out = ops.spmm(c.rowptr, c.col, None, x, n, addend=add, bias=bias, relu=True, drop=ops.Dropout(0.2, keep=keep))
    ref = (base + add + bias).relu() * keep * 1.25
    torch.testing.assert_close(out, ref, rtol=1e-6, atol=1e-6)
    # in-kernel Philox: deterministic per (seed, stream), keep rate ~ 0.8, values scaled 1.25
    o1 = ops.spmm(c.rowptr, c.col, None, x, n, drop=ops.Dropout(0.2, seed=7, stream_id=1))
    o2 = ops.spmm(c.rowptr, c.col, None, x, n, drop=ops.Dropout(0.2, seed=7, stream_id=1))
    o3 = ops.spmm(c.rowptr, c.col, None, x, n, drop=ops.Dropout(0.2, seed=8, stream_id=1))
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    nz = base != 0
    kept = (o1 != 0) & nz
    assert abs(kept.sum().item() / nz.sum().item() - 0.8) < 0.02
    torch.testing.assert_close(o1[kept], base[kept] * 1.25)


def test_gcn_conv_fwd_bwd_vs_oracle():
    from biomedkg_amd import graph as G, ops
    from oracle import pyg_ops as P
    n, e, din, dout = 150, 1200, 48, 32
    ei = _graph(n, e, 5)
    torch.manual_seed(5)
    x, w, b = torch.randn(n, din), torch.randn(dout, din) * 0.2, torch.randn(dout) * 0.1
    keep = torch.rand(n, dout) >= 0.2
    xo, wo, bo = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = torch.relu(P.gcn_conv(xo, ei, wo, bo)) * keep * 1.25
    gout = torch.randn(n, dout)
    ref.backward(gout)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    gr = G.Graph(ei.to(DEV), n)
    out = ops.gcn_conv(xd, wd, bd, gr, gr.gcn_weights(), relu=True, drop=ops.Dropout(0.2, keep=keep.to(DEV)))
    out.backward(gout.to(DEV))
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    for got, exp in ((xd, xo), (wd, wo), (bd, bo)):
        torch.testing.assert_close(got.grad.cpu(), exp.grad, rtol=1e-4, atol=1e-5)


def test_rgcn_conv_fwd_bwd_vs_oracle():
    from biomedkg_amd import graph as G, ops
    from oracle import pyg_ops as P
    n, e, r, din, dout = 120, 1500, 5, 40, 32
    ei = _graph(n, e, 6)
    torch.manual_seed(6)
    et = torch.randint(0, r, (e,))
    et[et == 2] = 3  # an empty relation
    x, w = torch.randn(n, din), torch.randn(r, din, dout) * 0.2
    root, b = torch.randn(din, dout) * 0.2, torch.randn(dout) * 0.1
    xo, wo, ro, bo = (t.clone().requires_grad_(True) for t in (x, w, root, b))
    ref = torch.relu(P.rgcn_conv(xo, ei, et, wo, ro, bo))
    gout = torch.randn(n, dout)
    ref.backward(gout)
    xd, wd, rd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, root, b))
    rg = G.RelGraph(ei.to(DEV), et.to(DEV), n, r)
    out = ops.rgcn_conv(xd, wd, rd, bd, rg, relu=True)
    out.backward(gout.to(DEV))
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    for got, exp in ((xd, xo), (wd, wo), (rd, ro), (bd, bo)):
        torch.testing.assert_close(got.grad.cpu(), exp.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("n,d", [(1, 8), (63, 16), (5000, 256), (130_001, 256), (3000, 768), (777, 2048), (500, 12)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_colsum(n, d, dtype):
    from biomedkg_amd import ops
    torch.manual_seed(n)
    x = torch.randn(n, d, device=DEV).to(dtype)
    got = ops.colsum(x)
    assert got.dtype == torch.float32
    torch.testing.assert_close(got.double(), x.double().sum(0), rtol=1e-5, atol=1e-5 * n ** 0.5)
    assert torch.equal(got, ops.colsum(x))                     # fixed summation order
    v = torch.randn(n, 2 * d, device=DEV).to(dtype)[:, :d]     # strided rows
    torch.testing.assert_close(ops.colsum(v).double(), v.double().sum(0), rtol=1e-5, atol=1e-5 * n ** 0.5)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_deterministic_split_k_products_match_the_atomic_ones(precision):
    """Per-chunk slabs + fixed-order reduce (bkg_set_deterministic_workspace) against the float-atomic split-K form:
    plain and per-relation grouped weight gradients, tile-aligned and ragged shapes, fp32 and bf16 operands."""
    from biomedkg_amd import ops
    torch.manual_seed(0)
    ops.set_matmul_precision(precision)
    try:
        cases = [(9000, 64, 48), (20000, 256, 256), (5000, 130, 70)]
        for k, m, n in cases:
            a, b = torch.randn(k, m, device=DEV), torch.randn(k, n, device=DEV)
            if precision == "bf16":
                a, b = ops.to_bf16(a) if (k * m) % 8 == 0 else a, ops.to_bf16(b) if (k * n) % 8 == 0 else b
            ref = ops.gemm(a, b, trans_a=True)
            ops.set_deterministic(True)
            d1, d2 = ops.gemm(a, b, trans_a=True), ops.gemm(a, b, trans_a=True)
            ops.set_deterministic(False)
            assert torch.equal(d1, d2)
            torch.testing.assert_close(d1, ref, rtol=1e-4, atol=1e-3)
        # grouped: 5 groups of very different sizes (one empty)
        sizes = [4000, 0, 1500, 9000, 33]
        ptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device=DEV)
        for m, n in ((128, 128), (96, 40)):
            h, dy = torch.randn(sum(sizes), m, device=DEV), torch.randn(sum(sizes), n, device=DEV)
            if precision == "bf16" and m % 8 == 0 and n % 8 == 0:
                h, dy = ops.to_bf16(h), ops.to_bf16(dy)
            ref = ops.gemm_grouped_k(h, dy, ptr, len(sizes))
            ops.set_deterministic(True)
            d1, d2 = ops.gemm_grouped_k(h, dy, ptr, len(sizes)), ops.gemm_grouped_k(h, dy, ptr, len(sizes))
            ops.set_deterministic(False)
            assert torch.equal(d1, d2) and float(d1[1].abs().max()) == 0.0
            torch.testing.assert_close(d1, ref, rtol=1e-4, atol=1e-3)
    finally:
        ops.set_deterministic(False)
        ops.set_matmul_precision("fp32")


@pytest.mark.gpu
@pytest.mark.parametrize("prec,n,d,r,hubs", [("bf16", 5000, 256, 30, 2), ("bf16", 20000, 64, 1, 0), ("fp32", 9001, 128, 60, 1),
                                             ("fp32", 4100, 32, 7, 3), ("bf16", 66000, 128, 2, 0)])
def test_kge_bwd_stream_form_matches_one_pass_form(prec, n, d, r, hubs):
    """DistMult's node-side backward (reference decoder.py:45-49 through autograd) in the stream form -- list-order
    endpoint / relation / g streams, then a weighted segment reduce -- against the one-pass form the goldens pin: hub nodes
    (lists cut into slices + fix-up), nodes without triplets, widths 32..256, one relation and many, both table types."""
    from biomedkg_amd import ops
    dev, t = "cuda", 300_001
    old_prec, old_streams = ops.get_matmul_precision(), ops.KGE_BWD_STREAMS
    launches = []
    call = ops.L.call
    try:
        ops.set_matmul_precision(prec)
        ops.L.call = lambda name, *a: (launches.append(name), call(name, *a))[1]

        def triplets(seed):
            g = torch.Generator(device=dev).manual_seed(seed)
            ei = torch.randint(0, n // 2 if seed % 2 else n, (2, t), device=dev, generator=g)
            if hubs:
                m = torch.rand(t, device=dev, generator=g) < 0.33
                ei[seed % 2][m] = torch.randint(0, hubs, (int(m.sum()),), device=dev, generator=g)
            return ei, torch.randint(0, r, (t,), device=dev, generator=g)
        (pe, pt), (ne, nt) = triplets(2), triplets(3)
        z0, rel0 = torch.randn(n, d, device=dev) * 0.3, torch.randn(r, d, device=dev) * 0.3
        got = []
        for streams in (False, True):
            ops.KGE_BWD_STREAMS = streams
            z, rel = z0.clone().requires_grad_(), rel0.clone().requires_grad_()
            loss = ops.kge_bce_loss(z, rel, pe.clone(), pt.clone(), ne.clone(), nt.clone(), ops.KGE_DISTMULT, reg=1e-2)[0]
            loss.backward()
            got.append((float(loss.detach()), z.grad.clone(), rel.grad.clone()))
        assert launches.count("bkg_kge_bwd_stream_f32") == 2 and launches.count("bkg_kge_bwd_nodes_f32") == 2
        (l0, gz0, gr0), (l1, gz1, gr1) = got
        assert l0 == l1 and torch.equal(gr0, gr1)
        assert float((gz0 - gz1).norm() / gz0.norm()) < 2e-6          # the same products, summed in another order
        rows = torch.zeros(n, dtype=torch.bool, device=dev)
        rows[pe.flatten()] = True
        rows[ne.flatten()] = True
        reg_only = 2e-2 * z0 / (n * d)                                # rows without triplets carry the L2 term alone
        assert torch.allclose(gz1[~rows], reg_only[~rows], rtol=1e-6, atol=0)
        # the upstream gradient and the L2 term as the last stream launch's tail (default) against a pass of their own,
        # under a non-unit upstream gradient and without the L2 term
        for reg, up in ((1e-2, 3.0), (0.0, 0.5)):
            tails = []
            for tail in (True, False):
                ops.KGE_BWD_TAIL = tail
                del launches[:]
                z, rel = z0.clone().requires_grad_(), rel0.clone().requires_grad_()
                loss = ops.kge_bce_loss(z, rel, pe.clone(), pt.clone(), ne.clone(), nt.clone(), ops.KGE_DISTMULT, reg=reg)[0]
                (loss * up).backward()
                tails.append((z.grad.clone(), launches.count("bkg_scale_add_rows_f32")))
            (ga, na), (gb, nb) = tails
            assert na == 0 and (nb == 1 or reg == 0.0)
            torch.testing.assert_close(ga, gb, rtol=2e-6, atol=1e-12)
    finally:
        ops.L.call = call
        ops.KGE_BWD_STREAMS = old_streams
        ops.KGE_BWD_TAIL = True
        ops.set_matmul_precision(old_prec)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,n,d,r,hubs", [("bf16", 9000, 256, 30, 3), ("fp32", 5000, 128, 7, 1), ("bf16", 40000, 64, 2, 0)])
def test_kge_positives_in_one_pass_match_the_two_pass_form(prec, n, d, r, hubs, monkeypatch):
    """The DistMult training pass of a graph's positives as ONE walk over (node, role, relation)-sorted incidence lists
    (``bkg_kge_onepass_f32``: scores, BCE terms, dz, d rel_emb records per run; reference kge_module.py:121-128 +
    decoder.py:45-49 through autograd) against scorer + node-side list walk: the same loss, scores, dz and d rel_emb --
    hub nodes (lists cut into slices + fix-up), nodes without triplets, runs longer than a batch, both table types, a
    non-unit upstream gradient, and a second backward over the kept graph."""
    from biomedkg_amd import ops
    dev, t = "cuda", 1_100_000
    old_prec = ops.get_matmul_precision()
    monkeypatch.setattr(ops, "_SORTED_MIN_TRIPLETS", 1 << 20)
    launches = []
    call = ops.L.call
    try:
        ops.set_matmul_precision(prec)
        ops.L.call = lambda name, *a: (launches.append(name), call(name, *a))[1]
        g = torch.Generator(device=dev).manual_seed(n)
        pe = torch.randint(0, n // 2, (2, t), device=dev, generator=g)          # nodes n // 2 .. n carry no positive triplet
        if hubs:
            m = torch.rand(t, device=dev, generator=g) < 0.3
            pe[0][m] = torch.randint(0, hubs, (int(m.sum()),), device=dev, generator=g)
        # relation ids skewed like a knowledge graph's: long (node, relation) runs
        pt = (torch.rand(t, device=dev, generator=g) ** 3 * r).long().clamp_(max=r - 1)
        ne = torch.randint(0, n, (2, t), device=dev, generator=g)
        nt = torch.randint(0, r, (t,), device=dev, generator=g).sort().values
        z0, rel0 = torch.randn(n, d, device=dev) * 0.3, torch.randn(r, d, device=dev) * 0.3
        got = []
        # a pair of tensors seen for the FIRST time goes the two-pass way (the lists pay from a graph's second step on)
        zf, rf = z0.clone().requires_grad_(), rel0.clone().requires_grad_()
        first = ops.kge_bce_loss(zf, rf, pe, pt, ne, nt, ops.KGE_DISTMULT, reg=1e-2)[0]
        assert launches.count("bkg_kge_onepass_f32") == 0 and launches.count("bkg_kge_fused_bce_dev_f32") == 2
        for one in (True, False):
            monkeypatch.setattr(ops, "KGE_ONEPASS", one)
            del launches[:]
            z, rel = z0.clone().requires_grad_(), rel0.clone().requires_grad_()
            loss, pp, npred = ops.kge_bce_loss(z, rel, pe, pt, ne, nt, ops.KGE_DISTMULT, reg=1e-2)
            (loss * 1.5).backward(retain_graph=True)
            gz1 = z.grad.clone()
            z.grad = rel.grad = None
            (loss * 1.5).backward()                                              # the positives' part must have stayed intact
            assert torch.equal(z.grad, gz1)
            got.append((float(loss.detach()), pp.clone(), npred.clone(), z.grad.clone(), rel.grad.clone(),
                        launches.count("bkg_kge_onepass_f32"), launches.count("bkg_kge_fused_bce_dev_f32")))
        (la, pa, na, za, ra, oa, fa), (lb, pb, nb_, zb, rb, ob, fb) = got
        assert oa == 1 and fa == 1 and ob == 0 and fb == 2
        assert abs(la - lb) <= 1e-6 * abs(lb) and abs(float(first.detach()) - lb) <= 1e-6 * abs(lb)
        tol = 2e-5 if prec == "fp32" else 1e-4
        torch.testing.assert_close(pa, pb, rtol=tol, atol=tol)
        assert torch.equal(na, nb_)
        assert float((za - zb).norm() / zb.norm()) < 5e-6 and float((ra - rb).norm() / rb.norm()) < 5e-6
        rows = torch.zeros(n, dtype=torch.bool, device=dev)
        rows[pe.flatten()] = True
        rows[ne.flatten()] = True
        if (~rows).any():
            assert torch.allclose(za[~rows], 1.5 * 2e-2 * z0[~rows] / (n * d), rtol=1e-5, atol=0)
    finally:
        ops.L.call = call
        ops.set_matmul_precision(old_prec)


@pytest.mark.gpu
def test_one_pass_walk_gives_way_when_its_lds_does_not_fit():
    """the walk's dynamic LDS is the relation table PLUS every wave's scratch rows (narrow tables have many lane
    groups: d = 64 bf16 -> 28 scratch rows).  240 relations x 64 columns pass the old ``n_rel * d * 4 <= 64 KB`` check and
    ask for 68.6 KB: ``bkg_kge_onepass_smem_bytes`` says so, ops keeps the two-pass form (same loss / gradients as with the
    walk switched off), and the C entry refuses the call with a message instead of a failed launch."""
    from biomedkg_amd import ops, _lib as L
    lib = L.lib()
    assert lib.bkg_kge_onepass_smem_bytes(64, 1, 240) == (240 + 4 * 7) * 64 * 4 > 65536 >= 240 * 64 * 4
    assert lib.bkg_kge_onepass_smem_bytes(256, 1, 30) == (30 + 4 * 3) * 256 * 4         # the bench width: 16 lanes per row
    assert lib.bkg_kge_onepass_smem_bytes(40, 1, 30) == 0 and lib.bkg_kge_onepass_batch(40, 1) == 0   # no instance: 5 pieces
    dev, n, d, r, t = "cuda", 30000, 64, 240, 1_100_000
    old_prec = ops.get_matmul_precision()
    launches, call = [], ops.L.call
    try:
        ops.set_matmul_precision("bf16")
        ops.L.call = lambda name, *a: (launches.append(name), call(name, *a))[1]
        g = torch.Generator(device=dev).manual_seed(5)
        pe = torch.randint(0, n, (2, t), device=dev, generator=g)
        pt = torch.randint(0, r, (t,), device=dev, generator=g)
        ne = torch.randint(0, n, (2, t), device=dev, generator=g)
        nt = torch.randint(0, r, (t,), device=dev, generator=g).sort().values
        z0, rel0 = torch.randn(n, d, device=dev) * 0.3, torch.randn(r, d, device=dev) * 0.3
        got = []
        for _ in range(3):                     # (the pair of tensors is "stable" from its second sighting on)
            z, rel = z0.clone().requires_grad_(), rel0.clone().requires_grad_()
            loss = ops.kge_bce_loss(z, rel, pe, pt, ne, nt, ops.KGE_DISTMULT, reg=1e-2)[0]
            loss.backward()
            got.append((float(loss.detach()), z.grad.clone(), rel.grad.clone()))
        assert launches.count("bkg_kge_onepass_f32") == 0
        assert got[0][0] == got[2][0] and torch.equal(got[0][1], got[2][1])
        # the entry point itself: an argument error with a message, not a launch failure
        zt = ops.to_bf16(z0)
        with pytest.raises(L.HipLibraryError, match="64 KB of LDS"):
            one = torch.zeros(16, dtype=torch.int32, device=dev)
            f = torch.zeros(4 * 64 * d, device=dev)
            L.call("bkg_kge_onepass_f32", L.ptr(zt), 1, zt.stride(0), n, d, L.ptr(rel0), rel0.stride(0), r, L.ptr(one),
                   L.ptr(one), L.ptr(one), L.ptr(one), 1, 1.0, 1.0, None, L.ptr(f), L.ptr(f), L.ptr(f), d,
                   L.ptr(one), L.ptr(f), L.stream())
    finally:
        ops.L.call = call
        ops.set_matmul_precision(old_prec)


def _mx_reference(zt, rel, ei, et, label, inv_t):
    """fp64 restatement of one label's DistMult training pass on the bf16 table (reference kge_module.py:121-128 +
    decoder.py:45-49 through autograd): scores, the unscaled BCE sum, dz, d rel_emb."""
    n, d = zt.shape
    z64, rel64 = zt.double(), rel.double()
    t = ei.size(1)
    s_ref = torch.empty(t, dtype=torch.float64, device=zt.device)
    dz = torch.zeros(n, d, dtype=torch.float64, device=zt.device)
    drel = torch.zeros(rel.size(0), d, dtype=torch.float64, device=zt.device)
    loss = torch.zeros((), dtype=torch.float64, device=zt.device)
    for lo in range(0, t, 1 << 18):
        h, tl, rr = ei[0, lo:lo + (1 << 18)], ei[1, lo:lo + (1 << 18)], et[lo:lo + (1 << 18)]
        zh, ztl, rw = z64[h], z64[tl], rel64[rr]
        s = (zh * rw * ztl).sum(1)
        s_ref[lo:lo + s.numel()] = s
        g = (torch.sigmoid(s) - label) * inv_t
        loss += (torch.clamp(s, min=0) - s * label + torch.log1p(torch.exp(-s.abs()))).sum()
        dz.index_add_(0, h, g[:, None] * rw * ztl)
        dz.index_add_(0, tl, g[:, None] * rw * zh)
        drel.index_add_(0, rr, g[:, None] * zh * ztl)
    return s_ref, loss, dz, drel


@pytest.mark.gpu
@pytest.mark.parametrize("n,t,r,hub", [(130, 5000, 30, False), (4001, 60_000, 32, True), (7, 300, 1, False), (2002, 1500, 5, False)])
def test_negatives_one_pass_walk_on_the_matrix_cores_against_fp64(n, t, r, hub):
    """csrc/kge_onepass_mx.hip called directly (bkg_kge_role_incidence + bkg_kge_onepass_mx_f32) on small and odd shapes: node
    counts that are not a multiple of four, nodes without any triplet (their dz rows are zeros), a hub whose list is thousands
    of entries long, self pairs, one relation and all 32, more nodes than triplets -- the BCE sum and dz against an fp64
    restatement on the same bf16 table to fp32 accuracy; d rel_emb (its terms g z_h (*) z_t are bf16 MFMA operands: 2^-9 per
    term, unbiased) to 1 % of the largest entry."""
    from biomedkg_amd import _lib as L, ops
    dev, d, label = "cuda", 256, 0.0
    g = torch.Generator(device=dev).manual_seed(n + t)
    old = ops.get_matmul_precision()
    try:
        ops.set_matmul_precision("bf16")
        z = torch.randn(n, d, device=dev, generator=g) * 0.3
        zt = ops.to_bf16(z)
        rel = (torch.randn(r, d, device=dev, generator=g) * 0.3).contiguous()
        ei = torch.randint(0, max(n - 3, 1), (2, t), device=dev, generator=g)     # (the last nodes carry no triplet)
        if hub:
            m = torch.rand(t, device=dev, generator=g) < 0.2
            ei[0][m] = 5
            ei[1][torch.rand(t, device=dev, generator=g) < 0.1] = 5               # (and self pairs of the hub)
        et = torch.randint(0, r, (t,), device=dev, generator=g)
        inv_t = 1.0 / (3 * t)
        rowptr2, packed = ops.role_incidence(ei, et, n)
        # the lists themselves: bounds are the sorted keys' counts, an entry = other endpoint | relation << 24
        cnt = (rowptr2[1:] - rowptr2[:-1]).long()
        assert int(rowptr2[0]) == 0 and int(rowptr2[-1]) == 2 * t
        assert torch.equal(cnt[0::2], torch.bincount(ei[0], minlength=n)) and torch.equal(cnt[1::2], torch.bincount(ei[1], minlength=n))
        key = torch.repeat_interleave(torch.arange(2 * n, device=dev), cnt)
        want = torch.cat([2 * ei[0] << 32 | (ei[1] | et << 24), (2 * ei[1] + 1) << 32 | (ei[0] | et << 24)]).sort().values
        assert torch.equal(((key << 32) | packed.long()).sort().values, want)
        blocks = L.lib().bkg_kge_onepass_mx_blocks(n)
        loss_part = torch.empty(4 * blocks, device=dev)
        dz = torch.full((n, d), float("nan"), device=dev)
        drel_part = torch.empty(blocks, 32, d, device=dev)
        L.call("bkg_kge_onepass_mx_f32", L.ptr(zt), zt.stride(0), n, L.ptr(rel), rel.stride(0), r, L.ptr(rowptr2), L.ptr(packed),
               None, 2 * t, label, inv_t, None, L.ptr(loss_part), L.ptr(dz), dz.stride(0), L.ptr(drel_part), blocks, L.stream())
        s_ref, loss_ref, dz_ref, drel_ref = _mx_reference(zt, rel, ei, et, label, inv_t)
        assert abs(float(loss_part.double().sum()) - float(loss_ref)) <= 5e-6 * float(loss_ref)     # (a lane's fp32 sum: the hub's 12 k terms)
        assert torch.isfinite(dz).all() and float((dz.double() - dz_ref).abs().max()) <= 1e-5 * float(dz_ref.abs().max())   # (fp32 sums; the hub's: 12 k terms)
        if n > 3:
            assert float(dz[n - 3:].abs().max()) == 0.0
        drel = drel_part.sum(0)[:r].double() * inv_t
        assert float((drel - drel_ref).abs().max()) <= 1e-2 * float(drel_ref.abs().max())
        if r < 32:
            assert float(drel_part.sum(0)[r:].abs().max()) == 0.0
        # with scores: written from the head end by the entries' positions (the generic sort keeps them)
        from biomedkg_amd.graph import _Sorter
        role = torch.cat([torch.zeros(t, dtype=torch.int64, device=dev), torch.ones(t, dtype=torch.int64, device=dev)])
        _, perm, rp2 = _Sorter.sort_rowptr(role, ei.view(-1), 2, n, 2 * n + 1, 1)
        assert torch.equal(rp2, rowptr2)
        p64 = perm.long()
        tid = torch.where(p64 < t, p64, p64 - t)
        packed2 = (torch.where(p64 < t, ei[1][tid], ei[0][tid]) | (et[tid] << 24)).to(torch.int32)
        score = torch.empty(t, device=dev)
        dz2 = torch.empty(n, d, device=dev)
        L.call("bkg_kge_onepass_mx_f32", L.ptr(zt), zt.stride(0), n, L.ptr(rel), rel.stride(0), r, L.ptr(rp2), L.ptr(packed2),
               L.ptr(perm), 2 * t, label, inv_t, L.ptr(score), L.ptr(loss_part), L.ptr(dz2), dz2.stride(0), L.ptr(drel_part),
               blocks, L.stream())
        assert float((score.double() - s_ref).abs().max()) <= 2e-6 * max(float(s_ref.abs().max()), 1.0)
        assert float((dz2 - dz).abs().max()) <= 1e-6 * float(dz.abs().max())          # (another order inside a list)
        # the entry refuses what it has no instance for
        assert not L.lib().bkg_kge_onepass_mx_ok(128, 1, r) and not L.lib().bkg_kge_onepass_mx_ok(256, 0, r)
        assert not L.lib().bkg_kge_onepass_mx_ok(256, 1, 33)
    finally:
        ops.set_matmul_precision(old)


@pytest.mark.gpu
def test_training_pass_with_both_walks_matches_the_two_pass_negatives(monkeypatch):
    """``ops.kge_bce_loss(lazy_scores=True)`` on a graph that stays: from the second step on the positives AND the fresh
    negatives go through their one-pass walks (``bkg_kge_onepass_mx_f32``, ``bkg_kge_onepass_f32``), the backward is ONE
    launch (``bkg_kge_dz_finish_f32``: both walks' sums, the upstream gradient, the L2 term).  Against the same op with the negatives
    through scorer + list walk: the same loss, dz to fp32 accuracy, d rel_emb to the bf16 operand's accuracy, the negatives'
    scores (computed on demand: ``LazyScores``) equal -- also for a non-unit upstream gradient and a second backward."""
    from biomedkg_amd import ops
    dev, n, d, r, t = "cuda", 20000, 256, 30, 1_100_000
    old = ops.get_matmul_precision()
    launches, call = [], ops.L.call
    try:
        ops.set_matmul_precision("bf16")
        ops.L.call = lambda name, *a: (launches.append(name), call(name, *a))[1]
        g = torch.Generator(device=dev).manual_seed(11)
        pe = torch.randint(0, n, (2, t), device=dev, generator=g)
        pt = (torch.rand(t, device=dev, generator=g) ** 3 * r).long().clamp_(max=r - 1)
        ne = torch.randint(0, n, (2, t), device=dev, generator=g)
        ne._bkg_exchangeable = True                     # (uniform pairs, as the library's sampler marks its own: no hubs)
        nt = torch.randint(0, r, (t,), device=dev, generator=g).sort().values
        z0, rel0 = torch.randn(n, d, device=dev) * 0.3, torch.randn(r, d, device=dev) * 0.3
        ops.onepass_stable(pe, pt)                      # (this pair of tensors has been seen: the lists pay)
        got = []
        for mx in (True, False):
            monkeypatch.setattr(ops, "KGE_ONEPASS_MX", mx)
            del launches[:]
            z, rel = z0.clone().requires_grad_(), rel0.clone().requires_grad_()
            loss, pp, npred, _ = ops.kge_bce_loss(z, rel, pe, pt, ne, nt, ops.KGE_DISTMULT, reg=1e-2, lazy_scores=True)
            (loss * 1.5).backward(retain_graph=True)
            gz1 = z.grad.clone()
            z.grad = rel.grad = None
            (loss * 1.5).backward()
            assert torch.equal(z.grad, gz1)
            n_mx, n_fused = launches.count("bkg_kge_onepass_mx_f32"), launches.count("bkg_kge_fused_bce_dev_f32")
            if isinstance(npred, ops.LazyScores):
                npred = npred()
            got.append((float(loss.detach()), pp.clone(), npred.clone(), z.grad.clone(), rel.grad.clone(), n_mx, n_fused))
        (la, pa, na, za, ra, mxa, fa), (lb, pb, nb_, zb, rb, mxb, fb) = got
        assert mxa == 1 and fa == 0 and mxb == 0 and fb == 1
        # negatives that do not carry the sampler's mark (hand-made pairs may hold hubs) keep the two-pass form
        monkeypatch.setattr(ops, "KGE_ONEPASS_MX", True)
        del launches[:]
        plain = ne.clone()
        z, rel = z0.clone().requires_grad_(), rel0.clone().requires_grad_()
        ops.kge_bce_loss(z, rel, pe, pt, plain, nt, ops.KGE_DISTMULT, reg=1e-2, lazy_scores=True)[0].backward()
        assert launches.count("bkg_kge_onepass_mx_f32") == 0 and launches.count("bkg_kge_fused_bce_dev_f32") == 1
        assert abs(la - lb) <= 1e-6 * abs(lb)
        torch.testing.assert_close(pa, pb, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(na, nb_, rtol=1e-4, atol=1e-4)
        assert float((za - zb).norm() / zb.norm()) < 5e-6
        assert float((ra - rb).norm() / rb.norm()) < 2e-3                    # (bf16 MFMA operands in the one-hot product)
    finally:
        ops.L.call = call
        ops.set_matmul_precision(old)

