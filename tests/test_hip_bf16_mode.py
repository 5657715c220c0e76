"""bf16-operand matrix-core mode (BASELINE config 3): operands rounded to bf16 (8 significant bits) while
staged, fp32 accumulate.  Tolerances follow from that rounding: products carry ~2^-9 relative error per
operand, so GEMMs are checked to 1e-2 of the output scale and step losses to 5e-3.  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def bf16_mode():
    from biomedkg_amd import ops
    ops.set_matmul_precision("bf16")
    yield
    ops.set_matmul_precision("fp32")


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("m,n,k", [(77, 45, 33), (300, 256, 768), (640, 256, 256), (256, 256, 20_000)])
def test_gemm_bf16_operands(ta, tb, m, n, k):
    from biomedkg_amd import ops
    torch.manual_seed(0)
    a = torch.randn((k, m) if ta else (m, k), device=DEV)
    b = torch.randn((n, k) if tb else (k, n), device=DEV)
    # exact reference of what the kernel computes: bf16-rounded operands, wide accumulate
    ar, br = a.bfloat16().double(), b.bfloat16().double()
    ref = (ar.t() if ta else ar) @ (br.t() if tb else br)
    out = ops.gemm(a, b, trans_a=ta, trans_b=tb)
    torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-3 * k ** 0.5)
    full = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
    assert (out.double() - full).abs().max() < 1e-2 * full.abs().max()


@pytest.mark.parametrize("d", [32, 128, 256, 768])
def test_spmm_bf16_table_matches_rounded_reference(d):
    """SpMM over a bf16 gather table == fp64 algebra on the bf16-rounded table (fp32 accumulate)."""
    from biomedkg_amd import graph as G, ops
    n, e = 2500, 120_000
    g = torch.Generator().manual_seed(9)
    ei = torch.randint(0, n, (2, e), generator=g)
    ei[1, : e // 4] = 5                                # one long row -> the split path
    x = torch.randn(n, d, generator=g)
    gr = G.Graph(ei.to(DEV), n)
    gw = gr.gcn_weights()
    xb = ops.to_bf16(x.to(DEV))
    assert xb.dtype == torch.bfloat16 and torch.equal(xb.cpu(), x.bfloat16())          # RNE, same as torch
    c = gr.by_dst
    out = ops.spmm(c.rowptr, c.col, gw.w_dst, xb, n, self_w=gw.self_w, relu=True, split=c.split)
    from oracle import pyg_ops as P
    e2, w = P.gcn_norm(ei, n)
    xr = x.bfloat16().double()
    ref = torch.zeros(n, d, dtype=torch.float64).index_add_(0, e2[1], xr[e2[0]] * w.double()[:, None]).relu()
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-4, atol=1e-4)
    gate = torch.randn(n, d, device=DEV)
    gb = ops.to_bf16(x.to(DEV), gate=gate, gate_scale=1.25)
    torch.testing.assert_close(gb.float().cpu(), torch.where(gate.cpu() > 0, x * 1.25, 0.0).bfloat16().float())


def test_grouped_gemm_bf16_operands():
    from biomedkg_amd import ops
    torch.manual_seed(1)
    sizes = [0, 130, 1, 257, 64]
    gptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device=DEV)
    s, k, n, g = sum(sizes), 40, 24, len(sizes)
    a, w, dy = torch.randn(s, k, device=DEV), torch.randn(g, k, n, device=DEV), torch.randn(s, n, device=DEV)
    y, dh, dw = ops.gemm_grouped_rows(a, w, gptr), ops.gemm_grouped_rows(dy, w, gptr, trans_b=True), ops.gemm_grouped_k(a, dy, gptr, g)
    ab, wb, dyb = a.bfloat16().double(), w.bfloat16().double(), dy.bfloat16().double()
    o = 0
    for i, sz in enumerate(sizes):
        sl = slice(o, o + sz)
        torch.testing.assert_close(y[sl].double(), ab[sl] @ wb[i], rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(dh[sl].double(), dyb[sl] @ wb[i].t(), rtol=1e-4, atol=1e-3)
        torch.testing.assert_close(dw[i].double(), ab[sl].t() @ dyb[sl], rtol=1e-4, atol=2e-3)
        o += sz


@pytest.mark.parametrize("name", ["rgcn_dismult_random", "rgcn_transe_lm_attention"])
def test_kge_step_bf16_mode_close_to_reference(golden, name):
    from biomedkg_amd import rand as R
    from biomedkg_amd.kge_module import KGEModule
    g = golden("kge_steps")[name]
    mod = KGEModule(**g["hparams"]).to(DEV)
    mod.load_state_dict(g["state"], strict=True)
    mod.train()

    class B:
        pass
    b = B()
    b.x, b.edge_index, b.edge_type = g["x"].to(DEV), g["edge_index"].to(DEV), g["edge_type"].to(DEV)
    with R.use(R.ReplayRand(g["tape"], DEV)):
        loss = mod.training_step(b)
    loss.backward()
    # four bf16-operand R-GCN layers feeding a cubic (DistMult) score: ~1e-2 relative on loss and gradients
    rel = abs(loss.item() - g["loss"].item()) / max(1.0, abs(g["loss"].item()))
    assert rel < 2e-2, (loss.item(), g["loss"].item())
    params = dict(mod.model.named_parameters())
    for k, v in g["grads"].items():
        err = (params[k].grad.cpu() - v).norm() / v.norm().clamp_min(1e-6)
        assert err < 1e-1, (k, float(err))


@pytest.mark.parametrize("m,n,k", [(1024, 32, 256), (3001, 256, 256), (2600, 160, 768), (70_000, 256, 256)])
@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("a_bf16,c_bf16", [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_xw_streaming(m, n, k, tb, a_bf16, c_bf16):
    """Tall activation x small weight (csrc/gemm_xw.hip): fp32 / bf16 A, bf16 weight image, fp32 / bf16 C."""
    from biomedkg_amd import ops
    torch.manual_seed(m + n)
    a = torch.randn(m, k, device=DEV)
    b = torch.randn((n, k) if tb else (k, n), device=DEV)
    assert ops._xw_ok(a, m, n, k)
    ar, br = a.bfloat16().double(), b.bfloat16().double()
    ref = ar @ (br.t() if tb else br)
    out = ops.gemm(a.bfloat16() if a_bf16 else a, b, trans_b=tb,
                   out_dtype=torch.bfloat16 if c_bf16 else torch.float32)
    assert out.dtype == (torch.bfloat16 if c_bf16 else torch.float32)
    if c_bf16:
        torch.testing.assert_close(out.double(), ref, rtol=2 ** -7, atol=1e-3 * k ** 0.5)
    else:
        torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-3 * k ** 0.5)


def test_gemm_xw_epilogues_and_groups():
    from biomedkg_amd import ops
    torch.manual_seed(21)
    m, n, k = 2500, 96, 256
    a, w = torch.randn(m, k, device=DEV), torch.randn(n, k, device=DEV)
    bias, c0, gate = torch.randn(n, device=DEV), torch.randn(m, n, device=DEV), torch.randn(m, n, device=DEV)
    pre = 0.5 * (a.bfloat16().double() @ w.bfloat16().double().t()) + bias.double()
    tol = dict(rtol=1e-4, atol=2e-2)
    torch.testing.assert_close(ops.gemm(a, w, trans_b=True, alpha=0.5, bias=bias, act=ops.ACT_RELU).double(), pre.relu(), **tol)
    torch.testing.assert_close(ops.gemm(a, w, trans_b=True, alpha=0.5, bias=bias, act=ops.ACT_ELU).double(),
                               torch.nn.functional.elu(pre), **tol)
    out = c0.clone()
    ops.gemm(a, w, trans_b=True, alpha=0.5, bias=bias, beta=2.0, out=out)
    torch.testing.assert_close(out.double(), pre + 2.0 * c0.double(), **tol)
    out = ops.gemm(a, w, trans_b=True, alpha=0.5, bias=bias, gate=gate, gate_scale=1.25)
    torch.testing.assert_close(out.double(), torch.where(gate > 0, pre * 1.25, 0.0), **tol)
    # the weight image follows in-place updates of the weight (optimizer steps)
    w.mul_(2.0)
    torch.testing.assert_close(ops.gemm(a, w, trans_b=True).double(), a.bfloat16().double() @ w.bfloat16().double().t(), **tol)
    # grouped rows (RGCNConv per-relation weights), empty / tiny / multi-tile groups, both weight orientations
    sizes = [0, 700, 1, 0, 1300, 64]
    gptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device=DEV)
    s, g = sum(sizes), len(sizes)
    h, wg, dy = torch.randn(s, 256, device=DEV), torch.randn(g, 256, 64, device=DEV), torch.randn(s, 64, device=DEV)
    wbig = torch.randn(g, 256, 256, device=DEV)
    y = ops.gemm_grouped_rows(h, wg, gptr)
    dh = ops.gemm_grouped_rows(h, wbig, gptr, trans_b=True, out_dtype=torch.bfloat16)
    assert dh.dtype == torch.bfloat16
    o = 0
    for i, sz in enumerate(sizes):
        sl = slice(o, o + sz)
        torch.testing.assert_close(y[sl].double(), h[sl].bfloat16().double() @ wg[i].bfloat16().double(), **tol)
        torch.testing.assert_close(dh[sl].double(), h[sl].bfloat16().double() @ wbig[i].bfloat16().double().t(),
                                   rtol=2 ** -7, atol=2e-2)
        o += sz


@pytest.mark.parametrize("m", [257, 4096, 4161, 33_000])
@pytest.mark.parametrize("c_bf16", [False, True])
def test_gemm_xw_register_weight_kernel_epilogues(m, c_bf16):
    """K = N = 256 over bf16 rows takes the 8-wave kernel (weight in registers, A tiles by LDS-DMA): every epilogue
    it accepts, partial last tiles, fewer tiles than ring slots, more blocks than tiles."""
    from biomedkg_amd import ops
    torch.manual_seed(m)
    a = (torch.randn(m, 256, device=DEV)).bfloat16()
    w, bias = torch.randn(256, 256, device=DEV) * 0.1, torch.randn(256, device=DEV)
    assert ops._xw_ok(a, m, 256, 256)
    cdt = torch.bfloat16 if c_bf16 else torch.float32
    tol = dict(rtol=2 ** -7, atol=2e-2) if c_bf16 else dict(rtol=1e-4, atol=2e-3)
    pre = a.double() @ w.bfloat16().double().t()
    torch.testing.assert_close(ops.gemm(a, w, trans_b=True, out_dtype=cdt).double(), pre, **tol)
    torch.testing.assert_close(ops.gemm(a, w.t().contiguous(), out_dtype=cdt).double(), pre, **tol)
    torch.testing.assert_close(ops.gemm(a, w, trans_b=True, alpha=0.5, bias=bias, act=ops.ACT_RELU, out_dtype=cdt).double(),
                               (0.5 * pre + bias.double()).relu(), **tol)
    torch.testing.assert_close(ops.gemm(a, w, trans_b=True, alpha=-1.5, act=ops.ACT_ELU, out_dtype=cdt).double(),
                               torch.nn.functional.elu(-1.5 * pre), **tol)
    # rows of a wider table (lda > K) into the columns of a wider result (ldc > N)
    wide = torch.randn(m, 512, device=DEV).bfloat16()
    out = torch.full((m, 512), 7.0, device=DEV)
    ops.gemm(wide[:, 256:], w, trans_b=True, out=out[:, :256])
    torch.testing.assert_close(out[:, :256].double(), wide[:, 256:].double() @ w.bfloat16().double().t(), rtol=1e-4, atol=2e-3)
    assert bool((out[:, 256:] == 7.0).all())


@pytest.mark.parametrize("c_bf16", [False, True])
@pytest.mark.parametrize("gathered", [False, True])
def test_gemm_xw_register_weight_kernel_row_groups(c_bf16, gathered):
    """The grouped form of the 8-wave kernel (RGCNConv's per-relation weights over bf16 rows, optionally gathered by
    index): empty, one-row, 63 / 64 / 65-row and multi-tile groups, more groups than blocks' ranges, both weight
    orientations -- against fp64 products of the same bf16 values."""
    from biomedkg_amd import ops
    torch.manual_seed(5 + c_bf16)
    sizes = [0, 1, 63, 0, 64, 65, 700, 0, 0, 4099, 2, 130, 31, 9000, 0]
    g, s_rows, n = len(sizes), sum(sizes), 6000
    gptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device=DEV)
    w = torch.randn(g, 256, 256, device=DEV) * 0.1
    cdt = torch.bfloat16 if c_bf16 else torch.float32
    tol = dict(rtol=2 ** -7, atol=2e-2) if c_bf16 else dict(rtol=1e-4, atol=2e-3)
    if gathered:
        table = torch.randn(n, 256, device=DEV).bfloat16()
        idx = torch.randint(0, n, (s_rows,), device=DEV, dtype=torch.int32)
        rows = table[idx.long()]
        kw = dict(a_rows=idx)
    else:
        table = rows = torch.randn(s_rows, 256, device=DEV).bfloat16()
        kw = {}
    for tb in (False, True):
        out = ops.gemm_grouped_rows(table, w, gptr, trans_b=tb, out_dtype=cdt, **kw)
        assert out.shape == (s_rows, 256) and out.dtype == cdt
        o = 0
        for i, sz in enumerate(sizes):
            wi = w[i].bfloat16().double()
            torch.testing.assert_close(out[o:o + sz].double(), rows[o:o + sz].double() @ (wi.t() if tb else wi), **tol)
            o += sz


def test_spmm_bf16_result_is_the_rounded_fp32_result():
    from biomedkg_amd import graph as G, ops
    n, e, d = 3000, 90_000, 256
    g = torch.Generator().manual_seed(4)
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    gr = G.Graph(ei, n)
    gw = gr.gcn_weights()
    x = torch.randn(n, d, device=DEV)
    bias = torch.randn(d, device=DEV)
    c = gr.by_dst
    for table in (x, ops.to_bf16(x)):
        ref = ops.spmm(c.rowptr, c.col, gw.w_dst, table, n, self_w=gw.self_w, bias=bias, relu=True, split=c.split)
        got = ops.spmm(c.rowptr, c.col, gw.w_dst, table, n, self_w=gw.self_w, bias=bias, relu=True, split=c.split,
                       out_dtype=torch.bfloat16)
        assert got.dtype == torch.bfloat16
        assert torch.equal(got, ref.bfloat16())


@pytest.mark.parametrize("n,d", [(4097, 64), (5000, 256), (70_001, 512)])
@pytest.mark.parametrize("src16", [False, True])
@pytest.mark.parametrize("gate", ["none", "fp32", "bf16"])
def test_gated_table_and_its_column_sums_in_one_pass(n, d, src16, gate):
    """The gated bf16 output-gradient table of a layer and its bias gradient from one kernel: bit-identical to the
    two-pass form (gate + cast, then column sums of the finished table)."""
    from biomedkg_amd import ops
    torch.manual_seed(n + d)
    x = torch.randn(n, d, device=DEV)
    x = x.bfloat16() if src16 else x
    g = None if gate == "none" else torch.randn(n, d, device=DEV)
    if gate == "bf16":
        g = g.bfloat16()
    table, cs = ops.to_bf16_colsum(x, gate=g, gate_scale=1.25)
    want = ops.to_bf16(x, gate=g, gate_scale=1.25)
    assert torch.equal(table, want)
    if src16 and g is None:
        assert cs is None                      # nothing to cast: the caller sums the table it already has
    else:
        assert torch.equal(cs, ops.colsum(want))
        torch.testing.assert_close(cs.double(), want.double().sum(0), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("d", [64, 256])
def test_spmm_bf16_addend_equals_its_fp32_image(d):
    """RGCNConv's root term may reach the segment sum as a bf16 table: same result as the fp32 table of those values,
    on the vector path (row splitting included) and for both result types."""
    from biomedkg_amd import graph as G, ops
    n, e = 5000, 150_000
    g = torch.Generator().manual_seed(8)
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    ei[1, : e // 3] = 7                                   # one long row: split path + fix-up
    c = G.Graph(ei, n).by_dst
    x = torch.randn(n, d, device=DEV).bfloat16()
    add16 = torch.randn(n, d, device=DEV).bfloat16()
    bias = torch.randn(d, device=DEV)
    for odt in (torch.float32, torch.bfloat16):
        a = ops.spmm(c.rowptr, c.col, None, x, n, addend=add16, bias=bias, relu=True, split=c.split, out_dtype=odt)
        b = ops.spmm(c.rowptr, c.col, None, x, n, addend=add16.float(), bias=bias, relu=True, split=c.split, out_dtype=odt)
        assert torch.equal(a, b)


@pytest.mark.parametrize("a16,b16", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("m,n,k", [(256, 256, 20_000), (256, 768, 9_001), (96, 40, 5_000)])
def test_weight_gradient_products_with_bf16_tables(a16, b16, m, n, k):
    from biomedkg_amd import ops
    torch.manual_seed(k)
    a, b = torch.randn(k, m, device=DEV), torch.randn(k, n, device=DEV)
    aa, bb = (a.bfloat16() if a16 else a), (b.bfloat16() if b16 else b)
    ref = a.bfloat16().double().t() @ b.bfloat16().double()
    out = ops.gemm(aa, bb, trans_a=True)
    torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-3 * k ** 0.5)
    sizes = [0, 1300, 1, 0, 2600, 64, k - 3965]
    gptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device=DEV)
    dw = ops.gemm_grouped_k(aa, bb, gptr, len(sizes))
    o = 0
    for i, sz in enumerate(sizes):
        sl = slice(o, o + sz)
        torch.testing.assert_close(dw[i].double(), a[sl].bfloat16().double().t() @ b[sl].bfloat16().double(),
                                   rtol=1e-4, atol=1e-3 * max(sz, 1) ** 0.5)
        o += sz


@pytest.mark.parametrize("kind", ["gcn", "rgcn"])
def test_bf16_hidden_activations_track_the_fp32_encoder(kind):
    """N >= 4096 and d % 256 == 0: the activations between conv layers are bf16 tables (ops._hidden_bf16_ok) and their
    gradients flow back in bf16; loss and every parameter gradient stay within bf16 rounding of the exact-fp32 run."""
    from biomedkg_amd import ops
    from biomedkg_amd.model.encoder import GCNEncoder, RGCN
    n, e, r, d = 6000, 90_000, 6, 256
    g = torch.Generator().manual_seed(8)
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    et = torch.randint(0, r, (e,), generator=g).to(DEV)
    x = torch.randn(n, 768, generator=g).to(DEV)
    tgt = torch.randn(n, d, generator=g).to(DEV)
    torch.manual_seed(0)
    enc = (GCNEncoder(768, d, d, 2, drop_out=False) if kind == "gcn" else RGCN(768, d, d, 2, r, drop_out=False)).to(DEV)
    args = (ei,) if kind == "gcn" else (ei, et)
    seen = []
    hooks = [m.register_forward_hook(lambda mod, inp, out: seen.append(out.dtype)) for m in enc.graph_layers]
    res = {}
    for mode in ("fp32", "bf16"):
        ops.set_matmul_precision(mode)
        seen.clear()
        enc.zero_grad(set_to_none=True)
        z = enc(x, *args)
        loss = ((z - tgt) ** 2).mean()
        loss.backward()
        res[mode] = (float(loss.detach()), [p.grad.clone() for p in enc.parameters()], list(seen))
    for h in hooks:
        h.remove()
    assert res["fp32"][2] == [torch.float32] * 4
    assert res["bf16"][2] == [torch.bfloat16] * 3 + [torch.float32]
    assert abs(res["bf16"][0] - res["fp32"][0]) <= 2e-2 * abs(res["fp32"][0])
    for gb, gf in zip(res["bf16"][1], res["fp32"][1]):
        assert gb.dtype == torch.float32
        assert float((gb - gf).norm() / gf.norm().clamp_min(1e-12)) < 5e-2


def test_masked_bf16_features():
    """GCL feature masking fused with the table cast: kept entries are bf16(x), a fraction ~p is zeroed, the mask is a
    pure function of (seed, stream), and p = 0 is the plain cast."""
    from biomedkg_amd import ops
    x = torch.randn(4096, 768, device=DEV)
    a = ops.masked_bf16(x, 0.4, 123, 1)
    assert a.dtype == torch.bfloat16 and torch.equal(a, ops.masked_bf16(x, 0.4, 123, 1))
    kept = a != 0
    assert torch.equal(a[kept], x.bfloat16()[kept])
    frac = 1.0 - float(kept.float().mean())
    assert abs(frac - 0.4) < 5e-3
    b = ops.masked_bf16(x, 0.4, 123, 2)
    assert 0.2 < float(((a != 0) & (b != 0)).float().mean()) < 0.5          # independent masks overlap on ~0.36
    assert torch.equal(ops.masked_bf16(x, 0.0, 5, 0), x.bfloat16())
    # column / row structure: no column or row is masked as a whole (element-wise mask, reference gcl.py:42,76)
    assert float(kept.float().mean(0).min()) > 0.5 and float(kept.float().mean(1).min()) > 0.5


def test_gathered_operands_equal_the_materialised_gather():
    """RGCNConv backward reads dY[s] = dOut[dst_s] through the index instead of storing it: same numbers."""
    from biomedkg_amd import ops
    torch.manual_seed(31)
    n, s_rows, g = 5000, 9000, 5
    sizes = [0, 3000, 1, 4999, 1000]
    gptr = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device=DEV)
    idx = torch.randint(0, n, (s_rows,), device=DEV, dtype=torch.int32)
    dout = torch.randn(n, 256, device=DEV).bfloat16()
    h = torch.randn(s_rows, 256, device=DEV).bfloat16()
    w = torch.randn(g, 256, 256, device=DEV)
    dy = dout[idx.long()]
    assert ops.gathered_k_ok(h, dout)
    a = ops.gemm_grouped_k(h, dout, gptr, g, b_rows=idx)
    b = ops.gemm_grouped_k(h, dy, gptr, g)
    torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-3)            # f32 atomics: order of the K splits
    c = ops.gemm_grouped_rows(dout, w, gptr, trans_b=True, a_rows=idx, out_dtype=torch.bfloat16)
    d = ops.gemm_grouped_rows(dy, w, gptr, trans_b=True, out_dtype=torch.bfloat16)
    assert torch.equal(c, d)


def test_bf16_training_follows_the_fp32_trajectory():
    """20 Adam steps of the KGE module (R-GCN + DistMult, device negatives) on a graph large enough for every bf16 fast
    path (N >= 4096, d = 256): the bf16 run's loss curve stays within 2 % of the exact-fp32 run with the same seeds."""
    from types import SimpleNamespace
    from biomedkg_amd import ops, rand as R, synthetic
    from biomedkg_amd.kge_module import KGEModule
    g = synthetic.primekg_like(6000, 120_000, 8, seed=3)
    ei, et = g["edge_index"].to(DEV), g["edge_type"].to(DEV)
    x = torch.randn(6000, 256, generator=torch.Generator().manual_seed(1)).to(DEV)
    curves = {}
    ops.set_deterministic(True)      # both runs repeatable: the comparison is between precisions, not between atomic orders
    for mode in ("fp32", "bf16"):
        ops.set_matmul_precision(mode)
        torch.manual_seed(0)
        m = KGEModule("rgcn", "dismult", 256, 256, 256, 2, 8, 1, "cosine", 1e-3, 0.1, "none", 1, "random").to(DEV)
        m.train()
        opt = torch.optim.Adam(m.model.parameters(), lr=1e-3)
        batch = SimpleNamespace(x=x, edge_index=ei, edge_type=et)
        losses = []
        with R.use(R.DeviceRand(seed=11)):
            for _ in range(20):
                opt.zero_grad(set_to_none=True)
                loss = m.training_step(batch, 0)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        curves[mode] = losses
    a, b = torch.tensor(curves["fp32"]), torch.tensor(curves["bf16"])
    assert torch.isfinite(b).all() and b[-1] < b[0]                 # it trains
    assert float(((a - b).abs() / a.abs()).max()) < 2e-2, (curves["fp32"][-3:], curves["bf16"][-3:])


def test_ggd_bf16_training_follows_the_fp32_trajectory():
    """20 Adam steps of the GGD module on a graph large enough for the side-by-side path (both views through every
    kernel, the head as a dot product per row): the bf16 run's loss curve stays within 2 % of the exact-fp32 run, which
    keeps the reference's Linear + sum (augmentation and dropout off: the comparison is between arithmetic paths)."""
    from biomedkg_amd import ops, rand as R
    from biomedkg_amd.gcl_module import GGDModule
    n, e = 6000, 100_000
    g = torch.Generator().manual_seed(12)
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    x = torch.randn(n, 768, generator=g).to(DEV)
    curves = {}
    ops.set_deterministic(True)
    try:
        for mode in ("fp32", "bf16"):
            ops.set_matmul_precision(mode)
            torch.manual_seed(0)
            m = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(DEV).train()
            m.model.encoder.drop_out = False
            m.model.p = 0.0
            opt = torch.optim.Adam(m.model.parameters(), lr=1e-3)
            losses = []
            with R.use(R.DeviceRand(seed=5)):
                if mode == "bf16":
                    assert m.model._views_ok(ops.cached_bf16(x), R.current())
                for _ in range(20):
                    opt.zero_grad(set_to_none=True)
                    loss = m.calculate_loss(x, ei)
                    loss.backward()
                    opt.step()
                    losses.append(float(loss.detach()))
            curves[mode] = losses
    finally:
        ops.set_deterministic(False)
        ops.set_matmul_precision("bf16")
    a, b = torch.tensor(curves["fp32"]), torch.tensor(curves["bf16"])
    assert torch.isfinite(b).all() and b[-1] < b[0]
    assert float(((a - b).abs() / a.abs()).max()) < 2e-2, (curves["fp32"][-3:], curves["bf16"][-3:])


def test_ggd_two_views_side_by_side_equal_two_passes():
    """GGD's positive + corrupted pass through every kernel at once (shared first-layer product, [N, 2h] gather rows,
    [2N, h] products) against the two separate encoder passes: same loss and parameter gradients (bf16 rounding)."""
    from biomedkg_amd import ops, rand as R
    from biomedkg_amd.gcl_module import GGDModule

    class TwoPass(R.DeviceRand):
        device_draws = False

    n, e = 6000, 100_000
    g = torch.Generator().manual_seed(12)
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    x = torch.randn(n, 768, generator=g).to(DEV)
    torch.manual_seed(0)
    m = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(DEV).train()
    m.model.encoder.drop_out = False
    m.model.p = 0.0                                              # no augmentation: compare the deterministic part
    res = []
    for prov in (TwoPass(seed=1), R.DeviceRand(seed=1)):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(99)                                    # same corruption permutation
        with R.use(prov):
            assert m.model._views_ok(ops.cached_bf16(x), prov) == prov.device_draws
            loss = m.calculate_loss(x, ei)
        loss.backward()
        res.append((float(loss.detach()), [p.grad.clone() for p in m.model.parameters()]))
    (la, ga), (lb, gb) = res
    assert abs(la - lb) <= 2e-3 * abs(la), (la, lb)
    for a, b in zip(ga, gb):
        assert float((a - b).norm() / a.norm().clamp_min(1e-12)) < 2e-2
    # with dropout and augmentation on, the side-by-side path trains (finite loss, gradients everywhere)
    m.model.encoder.drop_out, m.model.p = True, 1.0
    m.zero_grad(set_to_none=True)
    with R.use(R.DeviceRand(seed=3)):
        loss = m.calculate_loss(x, ei)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.model.parameters())


@pytest.mark.parametrize("n,d", [(1, 256), (5, 128), (4097, 256), (70_001, 256), (3000, 768)])
@pytest.mark.parametrize("x_bf16", [False, True])
def test_linear_rowsum_is_the_summed_linear(n, d, x_bf16):
    """GGD's head (reference gcl.py:86-90, ``mlp[-1](h).sum(1)``) as one dot product per row: value and every gradient
    against the fp64 Linear + sum on the same operand values."""
    from biomedkg_amd import ops
    torch.manual_seed(n + d)
    x = torch.randn(n, d, device=DEV)
    x = (x.bfloat16() if x_bf16 else x).requires_grad_()
    w = (torch.randn(d, d, device=DEV) * 0.1).requires_grad_()
    b = torch.randn(d, device=DEV).requires_grad_()
    assert ops.rowsum_linear_ok(x, w)
    s = ops.linear_rowsum(x, w, b)
    g = torch.randn(n, device=DEV)
    s.backward(g)
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    ref = torch.nn.functional.linear(xd, wd, bd).sum(1)
    ref.backward(g.double())
    torch.testing.assert_close(s.double(), ref.detach(), rtol=1e-5, atol=1e-4)
    tol = dict(rtol=2 ** -7, atol=1e-3) if x_bf16 else dict(rtol=1e-5, atol=1e-5)
    assert x.grad.dtype == x.dtype
    torch.testing.assert_close(x.grad.double(), xd.grad, **tol)
    torch.testing.assert_close(w.grad.double(), wd.grad, rtol=1e-4, atol=1e-4 * max(1.0, n ** 0.5))
    torch.testing.assert_close(b.grad.double(), bd.grad, rtol=1e-5, atol=1e-4)
    # run-to-run: fixed-order reductions everywhere
    x.grad = w.grad = b.grad = None
    s2 = ops.linear_rowsum(x, w, b)
    s2.backward(g)
    assert torch.equal(s2, s)


def test_pair_rows_and_its_gradient():
    from biomedkg_amd import ops
    torch.manual_seed(2)
    n, h = 5003, 256
    x = torch.randn(n, h, device=DEV).bfloat16().requires_grad_()
    perm = torch.randperm(n, device=DEV)
    out = ops.pair_rows(x, perm)
    assert torch.equal(out[:, :h], x.detach()) and torch.equal(out[:, h:], x.detach()[perm])
    g = torch.randn(n, 2 * h, device=DEV).bfloat16()
    out.backward(g)
    want = (g[:, :h].float() + torch.zeros(n, h, device=DEV).index_add_(0, perm, g[:, h:].float())).bfloat16()
    assert torch.equal(x.grad, want)


def test_dgi_two_views_side_by_side_equal_two_passes():
    from biomedkg_amd import rand as R
    from biomedkg_amd.gcl_module import DGIModule

    class TwoPass(R.DeviceRand):
        device_draws = False

    n, e = 6000, 100_000
    g = torch.Generator().manual_seed(13)
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    x = torch.randn(n, 768, generator=g).to(DEV)
    torch.manual_seed(0)
    m = DGIModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(DEV).train()
    m.model.encoder.drop_out = False
    res = []
    for prov in (TwoPass(seed=1), R.DeviceRand(seed=1)):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(99)
        with R.use(prov):
            loss = m.calculate_loss(x, ei)
        loss.backward()
        res.append((float(loss.detach()), [p.grad.clone() for p in m.model.parameters()]))
    (la, ga), (lb, gb) = res
    assert abs(la - lb) <= 2e-3 * abs(la), (la, lb)
    for a, b in zip(ga, gb):
        assert float((a - b).norm() / a.norm().clamp_min(1e-12)) < 2e-2


@pytest.mark.parametrize("d", [768, 256])
def test_projected_attention_fusion_tracks_the_exact_path(d):
    """bf16 mode evaluates AttentionFusion (reference fusion.py:17-31) in its projected form -- T = X (Wq^T Wk), the four
    scores per node as row dots, V over the combined rows (ops._attention_fusion_projected) -- two identities of the
    reference's arithmetic.  Held to the CPU oracle's AttentionFusion in float64 (the reference's literal form) and to the
    build's exact-fp32 mode; the literal bf16 form (BKG_PROJECTED_ATTENTION=0 semantics) is the yardstick for what bf16
    operands alone cost."""
    from biomedkg_amd import ops, synthetic
    from biomedkg_amd.utils.fusion import AttentionFusion
    from oracle import ref_modules as O
    n = 6000
    x = synthetic.lm_features(n, 2, d, seed=3)
    torch.manual_seed(2)
    f = AttentionFusion(d).to(DEV).eval()
    with torch.no_grad():
        for p in (f.q_proj.bias, f.k_proj.bias, f.v_proj.bias):
            p.uniform_(-0.3, 0.3)                                    # the bias terms of the identity must matter
    o = O.AttentionFusion(d).double()
    o.load_state_dict({k: v.double().cpu() for k, v in f.state_dict().items()})
    want = o(x.double()).detach()
    xd = x.to(DEV)
    ops.set_matmul_precision("fp32")                 # (this file's fixture puts every test in bf16 mode)
    exact = f(xd).double().cpu()
    assert float((exact - want).norm() / want.norm()) < 1e-5
    ops.set_matmul_precision("bf16")
    try:
        calls = []
        real = ops.L.call
        ops.L.call = lambda name, *a: (calls.append(name), real(name, *a))[1]
        try:
            got = f(xd).double().cpu()
        finally:
            ops.L.call = real
        assert "bkg_attn_scores_combine_bf16" in calls and "bkg_attn_fuse_f32" not in calls
        was = ops.PROJECTED_ATTENTION
        ops.PROJECTED_ATTENTION = False
        try:
            xd2 = xd.clone()
            literal = f(xd2).double().cpu()
        finally:
            ops.PROJECTED_ATTENTION = was
    finally:
        ops.set_matmul_precision("fp32")
    err_proj = float((got - want).norm() / want.norm())
    err_lit = float((literal - want).norm() / want.norm())
    assert err_proj < 1e-2, err_proj
    assert err_proj < 3.0 * err_lit + 1e-3, (err_proj, err_lit)     # no worse than bf16 operands make the literal form


@pytest.mark.parametrize("m", [1, 31, 32, 33, 100, 4097, 70_001])
@pytest.mark.parametrize("n,k", [(768, 768), (256, 768), (512, 512), (1024, 512)])
@pytest.mark.parametrize("c_bf16", [False, True])
def test_gemm_xw_weight_slab_kernel(m, n, k, c_bf16, monkeypatch):
    """K = 512 / 768 products of bf16 rows (csrc/gemm_xw.hip gemm_xw8k_kernel: a 256-column weight slab in registers, 32-row
    A tiles through an LDS-DMA ring): against fp64 on the bf16-rounded operands, and BIT-equal with the streaming kernel
    it replaces (same operands, same fp32 accumulation order along K: k ascending inside one accumulator) -- every row
    count around the 32-row tile, every column group, bias / ReLU / ELU / alpha epilogues, both result types."""
    from biomedkg_amd import ops
    prev = ops.get_matmul_precision()
    ops.set_matmul_precision("bf16")
    try:
        torch.manual_seed(m + n + k)
        a = torch.randn(m, k, device=DEV).bfloat16()
        w = torch.randn(n, k, device=DEV) / k ** 0.5
        bias = torch.randn(n, device=DEV)
        cdt = torch.bfloat16 if c_bf16 else torch.float32
        tol = dict(rtol=2 ** -7, atol=2e-2) if c_bf16 else dict(rtol=1e-4, atol=2e-3)
        for kw in (dict(), dict(bias=bias, act=ops.ACT_RELU), dict(bias=bias, act=2, alpha=0.5)):
            outs = {}
            for slab in ("1", "0"):
                monkeypatch.setenv("BKG_XW_SLAB", slab)
                outs[slab] = ops.gemm(a, w, trans_b=True, out_dtype=cdt, **kw)
            ref = kw.get("alpha", 1.0) * (a.double() @ w.bfloat16().double().t())
            if "bias" in kw:
                ref = ref + bias.double()
                ref = torch.relu(ref) if kw["act"] == ops.ACT_RELU else torch.where(ref > 0, ref, torch.expm1(ref))
            torch.testing.assert_close(outs["1"].double(), ref, **tol)
            if k == 768:      # (the streaming kernel sums K in 256-slabs into the same accumulator: the same order)
                assert torch.equal(outs["1"], outs["0"])
    finally:
        ops.set_matmul_precision(prev)


@pytest.mark.parametrize("n_nodes", [1, 16, 17, 130, 5000, 33_333])
@pytest.mark.parametrize("d", [512, 768])
@pytest.mark.parametrize("how", ["eval", "philox"])
def test_redaf_one_launch_on_bf16_rows_weight_slab_kernel(n_nodes, d, how, monkeypatch):
    """ReDAF as the epilogue of the weight-slab kernel (the shape the training step runs: bf16 feature rows, d = 768):
    bit-equal with the streaming kernel's epilogue -- same accumulators, same Philox masks -- which the goldens hold to
    the reference's class."""
    from biomedkg_amd import ops
    prev = ops.get_matmul_precision()
    ops.set_matmul_precision("bf16")
    try:
        g = torch.Generator(device=DEV).manual_seed(n_nodes + d)
        x = torch.randn(n_nodes, 2, d, device=DEV, generator=g).bfloat16()
        w = torch.randn(d, d, device=DEV, generator=g) / d ** 0.5
        b = torch.randn(d, device=DEV, generator=g) * 0.1
        scale = torch.rand(2, d, device=DEV, generator=g) + 0.25
        drop = ops.Dropout(0.1, keep=None, seed=99, stream_id=3) if how == "philox" else None
        outs = {}
        for slab in ("1", "0"):
            monkeypatch.setenv("BKG_XW_SLAB", slab)
            outs[slab] = ops.redaf_fusion(x, w, b, scale, drop).clone()
        assert torch.equal(outs["1"], outs["0"])
        if how == "eval":
            t = torch.relu(x.reshape(2 * n_nodes, d).double() @ w.bfloat16().double().t() + b.double()) * scale.repeat(n_nodes, 1).double()
            torch.testing.assert_close(outs["1"].double(), torch.relu(t).view(n_nodes, 2, d).mean(1), rtol=2e-4, atol=2e-4)
    finally:
        ops.set_matmul_precision(prev)


@pytest.mark.parametrize("m", [4096, 4161, 70_001, 260_000])
@pytest.mark.parametrize("scale", [1.0, 1.25])
def test_gate_and_bias_gradient_in_the_product(m, scale):
    """``ops.gemm_gated``: the input-gradient product of a 256 -> 256 layer with the gate of the layer below and that
    layer's bias gradient in its epilogue == the product, then ``to_bf16_colsum`` over its bf16 result: the table bit for
    bit (both round the product to bf16, multiply, round again), the column sums to fp32 summation order."""
    from biomedkg_amd import ops
    g = torch.Generator().manual_seed(m)
    a = torch.randn(m, 256, generator=g).to(DEV).bfloat16()
    w = (torch.randn(256, 256, generator=g) / 16).to(DEV)
    act = torch.relu(torch.randn(m, 256, generator=g)).to(DEV).bfloat16()      # ~half of it zero, like dropout(relu(.))
    assert ops.gemm_gated_ok(a, w, act)
    dx, cs = ops.gemm_gated(a, w, act, scale)
    ref, cs_ref = ops.to_bf16_colsum(ops.gemm(a, w, out_dtype=torch.bfloat16), gate=act, gate_scale=scale)
    assert dx.dtype == torch.bfloat16 and torch.equal(dx.view(torch.int16), ref.view(torch.int16))
    if cs_ref is None:          # (<= 4096 rows: the one-pass form leaves the sums to the caller)
        cs_ref = ops.colsum(ref)
    mag = ref.float().abs().sum(0)
    assert float(((cs - cs_ref).abs() / mag.clamp_min(1e-6)).max()) < 1e-5
    torch.testing.assert_close(cs.double(), ref.double().sum(0), rtol=1e-4, atol=1e-5 * float(mag.max()))


@pytest.mark.parametrize("drop_out", [False, True])
def test_ggd_gate_in_the_product_equals_the_gate_pass(drop_out, monkeypatch):
    """The GGD step with the hidden layers' backward gates inside the products above them (``ops.GateLink``) against the
    same step with a gate pass per layer: the same loss, the same gradients to fp32 summation order."""
    from biomedkg_amd import ops, rand as R
    from biomedkg_amd.gcl_module import GGDModule
    n, e = 6000, 100_000
    g = torch.Generator().manual_seed(13)
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    x = torch.randn(n, 768, generator=g).to(DEV)
    torch.manual_seed(0)
    m = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(DEV).train()
    m.model.encoder.drop_out = drop_out
    m.model.p = 0.0
    monkeypatch.setattr(ops, "GGD_FOLD_LAST", False)      # (all four conv layers as layers: three hand-overs)
    calls = {"n": 0}
    real = ops.gemm_gated

    def counted(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    monkeypatch.setattr(ops, "gemm_gated", counted)
    res = []
    for on in (True, False):
        monkeypatch.setattr(ops, "GATE_IN_PRODUCT", on)
        m.zero_grad(set_to_none=True)
        torch.manual_seed(99)
        with R.use(R.DeviceRand(seed=1)):
            loss = m.calculate_loss(x, ei)
        loss.backward()
        res.append((float(loss.detach()), {k: p.grad.clone() for k, p in m.model.named_parameters()}))
    assert calls["n"] == 3                        # the three hidden activations of the four-layer stack
    (la, ga), (lb, gb) = res
    assert la == lb
    for k in ga:       # (the weight-gradient products add their K chunks with float atomics: order, not values, differs)
        torch.testing.assert_close(ga[k], gb[k], rtol=1e-4, atol=1e-5 * float(gb[k].abs().max()) + 1e-9)


@pytest.mark.parametrize("d", [256, 512])
@pytest.mark.parametrize("with_addend", [False, True])
def test_spmm_result_through_another_layers_gate(d, with_addend):
    """``ops.spmm(out_gate=...)``: a bf16 aggregation result written through the ReLU / dropout backward gate of the layer
    below == the gate pass (``to_bf16(gate=...)``) over the ungated bf16 result, bit for bit (split long rows, short rows
    and column halves included: the graph is skewed)."""
    from biomedkg_amd import graph as G, ops
    n, e = 9000, 300_000
    g = torch.Generator().manual_seed(d)
    dst = (torch.rand(e, generator=g) ** 3 * n).long().clamp_(max=n - 1)      # a few hub rows, many short ones
    src = torch.randint(0, n, (e,), generator=g)
    gr = G.Graph(torch.stack([src, dst]).to(DEV), n)
    c = gr.by_dst
    w = torch.rand(e, generator=g).to(DEV)
    table = torch.randn(n, d, generator=g).to(DEV).bfloat16()
    addend = torch.randn(n, d, generator=g).to(DEV).bfloat16() if with_addend else None
    act = torch.relu(torch.randn(n, d, generator=g)).to(DEV).bfloat16()
    ref = ops.to_bf16(ops.spmm(c.rowptr, c.col, w, table, n, addend=addend, split=c.split, out_dtype=torch.bfloat16),
                      gate=act, gate_scale=1.25)
    out = ops.spmm(c.rowptr, c.col, w, table, n, addend=addend, split=c.split, out_dtype=torch.bfloat16, out_gate=act,
                   out_gate_scale=1.25)
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    with pytest.raises(ValueError):
        ops.spmm(c.rowptr, c.col, w, table, n, split=c.split, out_gate=act)          # an fp32 result takes no gate


@pytest.mark.parametrize("drop_out", [False, True])
def test_rgcn_gate_in_the_aggregation_equals_the_gate_pass(drop_out, monkeypatch):
    """The R-GCN + DistMult step with the hidden layers' backward gates inside the aggregation that makes their output
    gradient (``ops.GateLink``) against the same step with a gate pass per layer: same loss, gradients to fp32 order."""
    from biomedkg_amd import ops, rand as R
    from biomedkg_amd.kge_module import KGEModule
    n, e, r = 6000, 120_000, 7
    g = torch.Generator().manual_seed(14)
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    et = torch.randint(0, r, (e,), generator=g).to(DEV)
    x = torch.randn(n, 256, generator=g).to(DEV)
    torch.manual_seed(0)
    m = KGEModule(encoder_name="rgcn", decoder_name="dismult", in_dim=256, hidden_dim=256, out_dim=256, num_hidden_layers=2,
                  num_relation=r, num_heads=1, scheduler_type="cosine", learning_rate=1e-3, warm_up_ratio=0.03,
                  fuse_method="none", neg_ratio=1, node_init_method="random").to(DEV).train()
    m.model.encoder.drop_out = drop_out
    seen = {"gated": 0}
    real = ops.spmm

    def counted(*a, **k):
        seen["gated"] += k.get("out_gate") is not None
        return real(*a, **k)

    monkeypatch.setattr(ops, "spmm", counted)
    res = []
    for on in (True, False):
        monkeypatch.setattr(ops, "GATE_IN_PRODUCT", on)
        m.zero_grad(set_to_none=True)
        z_loss = None
        with R.use(R.DeviceRand(seed=5)):
            z = m(x, ei, et)
            z_loss = (z.float() ** 2).mean() + z.float().sum() * 1e-3
        z_loss.backward()
        res.append((float(z_loss.detach()), {k: p.grad.clone() for k, p in m.model.encoder.named_parameters()}))
    assert seen["gated"] == 3
    (la, ga), (lb, gb) = res
    assert la == lb
    for k in ga:
        torch.testing.assert_close(ga[k], gb[k], rtol=1e-4, atol=1e-5 * float(gb[k].abs().max()) + 1e-9)


@pytest.mark.parametrize("aug", [False, True])
def test_ggd_last_layer_folded_into_the_head_equals_the_layer(aug, monkeypatch):
    """GGD's scores are linear in the last conv layer's input (no activation behind that layer, a summed Linear as the
    head): ``A_hat (H w') + const`` with ``w' = W_L^T W_m^T 1`` (``GGD._forward_views``, one column per view through the
    aggregation) against the layer computed as a layer -- the same loss and the same gradient for EVERY parameter (the folded
    layer's and the head's included) to bf16 rounding, and both as close to the exact fp32 step."""
    from biomedkg_amd import ops, rand as R
    from biomedkg_amd.gcl_module import GGDModule
    n, e = 6000, 100_000
    g = torch.Generator().manual_seed(21)
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    x = torch.randn(n, 768, generator=g).to(DEV)
    torch.manual_seed(0)
    m = GGDModule(in_dim=768, hidden_dim=256, out_dim=256, num_hidden_layers=2, fuse_method="none").to(DEV).train()
    m.model.encoder.drop_out = aug
    m.model.p = 1.0 if aug else 0.0                     # with: feature masking + dropout_edge view + layer dropout
    narrow = {"n": 0}
    real = ops.gcn_propagate_narrow
    monkeypatch.setattr(ops, "gcn_propagate_narrow", lambda *a, **k: (narrow.__setitem__("n", narrow["n"] + 1), real(*a, **k))[1])

    def step(fold, mode="bf16"):
        ops.set_matmul_precision(mode)
        monkeypatch.setattr(ops, "GGD_FOLD_LAST", fold)
        m.zero_grad(set_to_none=True)
        torch.manual_seed(99)
        with R.use(R.DeviceRand(seed=4)):
            loss = m.calculate_loss(x, ei)
        loss.backward()
        return float(loss.detach()), {k: p.grad.clone() for k, p in m.model.named_parameters()}
    launches = []
    call = ops.L.call
    monkeypatch.setattr(ops.L, "call", lambda name, *a: (launches.append(name), call(name, *a))[1])
    (la, ga), (lb, gb) = step(True), step(False)
    assert narrow["n"] == 1
    # the head's gradient leaves through the last hidden layer's gate: one fused launch, once (the folded step)
    assert launches.count("bkg_outer_gate_colsum_bf16") == 1
    lx, gx = step(False, "fp32")
    ops.set_matmul_precision("bf16")
    # (bf16 mode against the exact step: 0.5 % with the augmentations on -- the folded form no further away than the layer form)
    assert abs(la - lb) <= 2e-3 * abs(lb) and abs(la - lx) <= max(5e-3 * abs(lx), 1.5 * abs(lb - lx)), (la, lb, lx)
    for k in ga:
        assert ga[k] is not None and torch.isfinite(ga[k]).all()
        rel_ab = float((ga[k] - gb[k]).norm() / gb[k].norm().clamp_min(1e-12))
        rel_ax = float((ga[k] - gx[k]).norm() / gx[k].norm().clamp_min(1e-12))
        rel_bx = float((gb[k] - gx[k]).norm() / gx[k].norm().clamp_min(1e-12))
        assert rel_ab < 3e-2, (k, rel_ab)
        assert rel_ax < max(3e-2, 1.5 * rel_bx), (k, rel_ax, rel_bx)        # no further from the exact step than the layer form


@pytest.mark.parametrize("n,d", [(4097, 256), (70_001, 256), (9_000, 768)])
def test_row_dot_gradient_through_the_gate_of_the_layer_below(n, d):
    """``ops.row_dot(x, w, link_in=...)``: the backward writes ``g (x) w`` through the gate of the layer whose output ``x``
    is and leaves that layer's bias gradient on the link (csrc/colsum.hip ``bkg_outer_gate_colsum_bf16``) == the outer
    product as a bf16 table, then ``to_bf16_colsum`` over it -- table and sums bit for bit (same blocking, same order)."""
    from biomedkg_amd import ops
    g = torch.Generator().manual_seed(n)
    x = torch.relu(torch.randn(n, d, generator=g)).to(DEV).bfloat16().requires_grad_()
    w = torch.randn(d, generator=g).to(DEV).requires_grad_()
    up = torch.randn(n, generator=g).to(DEV)
    link = ops.GateLink()
    link.relu, link.scale = True, 1.25
    s = ops.row_dot(x, w, link_in=link)
    torch.testing.assert_close(s.double(), x.double() @ w.double(), rtol=1e-5, atol=1e-4)
    dx, dw = torch.autograd.grad(s, (x, w), up)
    assert link.applied and link.db is not None
    plain_dx, plain_dw = torch.autograd.grad(ops.row_dot(x, w), (x, w), up)      # no link: the ungated outer product
    ref, cs = ops.to_bf16_colsum(plain_dx, gate=x.detach(), gate_scale=1.25)
    assert torch.equal(dx.view(torch.int16), ref.view(torch.int16)) and torch.equal(link.db, cs)
    assert torch.equal(dw, plain_dw)


@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_rgcn_first_layer_segment_means_of_constant_features_are_kept(mode, monkeypatch):
    """The first R-GCN layer aggregates CONSTANT node features (reference kge_module.py:77-84): its segment means are
    computed once per (graph, feature tensor, version) and reused by the following steps -- same outputs and gradients as
    recomputing them; a changed feature tensor (in-place edit or another object) is aggregated again."""
    from biomedkg_amd import ops
    from biomedkg_amd.model.encoder import RGCN
    ops.set_matmul_precision(mode)
    n, e, r = 6000, 90_000, 5
    g = torch.Generator().manual_seed(31)
    ei = torch.randint(0, n, (2, e), generator=g).to(DEV)
    et = torch.randint(0, r, (e,), generator=g).to(DEV)
    x = torch.randn(n, 256, generator=g).to(DEV)
    torch.manual_seed(0)
    enc = RGCN(256, 256, 256, 1, r, drop_out=False).to(DEV).train()
    tagged = []
    real = ops.spmm
    monkeypatch.setattr(ops, "spmm", lambda *a, **k: (tagged.append(k.get("tag")), real(*a, **k))[1])

    def step():
        enc.zero_grad(set_to_none=True)
        del tagged[:]
        z = enc(x, ei, et)
        z.float().pow(2).mean().backward()
        return z.detach().clone(), [p.grad.clone() for p in enc.parameters()], tagged.count("rgcn_segment_mean")
    z1, g1, c1 = step()
    z2, g2, c2 = step()
    assert c1 == 3 and c2 == 2                       # three layers; the second step reuses the first layer's means
    assert torch.equal(z1, z2)
    for a, b in zip(g1, g2):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6 * float(b.abs().max()) + 1e-12)
    monkeypatch.setattr(ops, "RGCN_KEEP_SEGMENT_MEANS", False)
    z3, g3, c3 = step()
    assert c3 == 3 and torch.equal(z1, z3)
    monkeypatch.setattr(ops, "RGCN_KEEP_SEGMENT_MEANS", True)
    x.mul_(2.0)                                      # another version of the tensor: aggregated again
    z4, _, c4 = step()
    assert c4 == 3 and not torch.equal(z4, z1)
    with torch.no_grad():                            # an evaluation pass neither needs nor disturbs what is kept
        enc(x, ei, et)
    assert step()[2] == 2
