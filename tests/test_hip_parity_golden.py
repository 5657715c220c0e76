"""HIP path (through the reference-shaped module API) vs. the golden vectors recorded from the
UNMODIFIED reference (tests/golden/make_goldens.py), with the recorded random draws replayed.
fp32; tolerances stated per assertion.  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOSS_TOL = dict(rtol=0.0, atol=1e-5)      # north_star: contrastive / KGE loss within 1e-5 (fp32)
VEC_TOL = dict(rtol=1e-4, atol=1e-5)


class Batch:
    pass


def _dev(t):
    return t.to(DEV)


@pytest.mark.parametrize("name", ["dismult", "transe"])
def test_decoders_vs_reference(golden, name):
    from biomedkg_amd import ops
    g = golden("decoders")[name]
    z = _dev(g["z"]).requires_grad_(True)
    rel = _dev(g["rel_emb"]).requires_grad_(True)
    kind = ops.KGE_DISTMULT if name == "dismult" else ops.KGE_TRANSE
    s = ops.kge_score(z, rel, _dev(g["edge_index"]), _dev(g["edge_type"]), kind)
    torch.testing.assert_close(s.detach().cpu(), g["score"], **VEC_TOL)
    s.backward(_dev(g["grad_out"]))
    torch.testing.assert_close(z.grad.cpu(), g["grad_z"], **VEC_TOL)
    torch.testing.assert_close(rel.grad.cpu(), g["grad_rel"], **VEC_TOL)


def test_fusion_vs_reference(golden):
    from biomedkg_amd import rand as R
    from biomedkg_amd.utils.fusion import AttentionFusion, ReDAF
    g = golden("fusion")
    a = AttentionFusion(48).to(DEV)
    a.load_state_dict(g["attention"]["state"])
    torch.testing.assert_close(a(_dev(g["attention"]["x"])).cpu(), g["attention"]["y"], **VEC_TOL)
    r = ReDAF(48).to(DEV)
    r.load_state_dict(g["redaf_eval"]["state"])
    r.eval()
    torch.testing.assert_close(r(_dev(g["redaf_eval"]["x"])).cpu(), g["redaf_eval"]["y"], **VEC_TOL)
    r.train()
    rp = R.ReplayRand(g["redaf_train"]["tape"], DEV)
    with R.use(rp):
        y = r(_dev(g["redaf_train"]["x"]))
    assert rp.exhausted()
    torch.testing.assert_close(y.cpu(), g["redaf_train"]["y"], **VEC_TOL)


GCL_CASES = ["grace_attention", "grace_none", "ggd_redaf", "ggd_none_a", "ggd_none_b", "dgi_attention", "dgi_none"]


@pytest.mark.parametrize("name", GCL_CASES)
def test_gcl_training_step_vs_reference(golden, name):
    from biomedkg_amd import gcl_module as M
    from biomedkg_amd import rand as R
    g = golden("gcl_steps")[name]
    mod = getattr(M, g["kind"])(**g["hparams"], fuse_method=g["fuse_method"]).to(DEV)
    mod.load_state_dict(g["state"], strict=True)   # the reference's own key layout
    mod.train()
    b = Batch()
    b.x, b.edge_index = _dev(g["x"]), _dev(g["edge_index"])
    rp = R.ReplayRand(g["tape"], DEV)
    with R.use(rp):
        loss = mod.training_step(b)
    assert rp.exhausted()
    torch.testing.assert_close(loss.detach().cpu(), g["loss"], **LOSS_TOL)
    loss.backward()
    params = dict(mod.model.named_parameters())
    for k, v in g["grads"].items():
        torch.testing.assert_close(params[k].grad.cpu(), v, **VEC_TOL, msg=lambda m, k=k: f"{k}: {m}")
    mod.eval()
    with torch.no_grad():
        torch.testing.assert_close(mod(b.x, b.edge_index).cpu(), g["z_eval"], **VEC_TOL)


KGE_CASES = ["rgcn_dismult_random", "rgcn_dismult_neg3", "rgcn_transe_lm_attention", "rgcn_transe_gcl_fixid"]


@pytest.mark.parametrize("name", KGE_CASES)
def test_kge_training_step_vs_reference(golden, name):
    from biomedkg_amd import rand as R
    from biomedkg_amd.kge_module import KGEModule
    g = golden("kge_steps")[name]
    mod = KGEModule(**g["hparams"]).to(DEV)
    mod.load_state_dict(g["state"], strict=True)
    mod.train()
    if g["fix_edge_id"] is not None:
        mod.fix_edge_id = g["fix_edge_id"]
    b = Batch()
    b.x, b.edge_index, b.edge_type = _dev(g["x"]), _dev(g["edge_index"]), _dev(g["edge_type"])
    rec = {}
    enc = mod.model.encode

    def encode(*a):
        rec["z"] = enc(*a)
        rec["z"].retain_grad()
        return rec["z"]

    mod.model.encode = encode
    rp = R.ReplayRand(g["tape"], DEV)
    with R.use(rp):
        loss = mod.training_step(b)        # fused score + BCE path
    assert rp.exhausted()
    _, pos_pred, neg_pred = mod._last_scores
    torch.testing.assert_close(loss.detach().cpu(), g["loss"], **LOSS_TOL)
    torch.testing.assert_close(rec["z"].detach().cpu(), g["z"], **VEC_TOL)
    torch.testing.assert_close(pos_pred.cpu(), g["pos_pred"], **VEC_TOL)
    torch.testing.assert_close(neg_pred.cpu(), g["neg_pred"], **VEC_TOL)
    loss.backward()
    torch.testing.assert_close(rec["z"].grad.cpu(), g["grad_z"], **VEC_TOL)
    # the unfused route (decode + BCE, what validation_step uses) gives the same numbers
    with R.use(R.ReplayRand(g["tape"], DEV)):
        z2, p2, n2, pred, gt = mod._score_step(b)
        loss2 = mod._loss(z2, pred, gt)
    torch.testing.assert_close(loss2.detach().cpu(), g["loss"], **LOSS_TOL)
    torch.testing.assert_close(p2.detach().cpu(), g["pos_pred"], **VEC_TOL)
    params = dict(mod.model.named_parameters())
    for k, v in g["grads"].items():
        torch.testing.assert_close(params[k].grad.cpu(), v, **VEC_TOL, msg=lambda m, k=k: f"{k}: {m}")


def test_edge_wise_precision_vs_reference(golden):
    from biomedkg_amd.utils.metrics import EdgeWisePrecision
    g = golden("edge_wise_precision")
    m = EdgeWisePrecision(class_mapping=g["mapping"])
    m.update(_dev(g["preds"][:120]), _dev(g["target"][:120]))
    m.update(_dev(g["preds"][120:]), _dev(g["target"][120:]))
    got = m.compute()
    for k, v in g["result"].items():
        assert abs(got[k] - v) < 1e-7


def test_ntxent_matches_oracle_at_c2_size():
    """N = 5000, h = 128 (BASELINE config C2): loss within 1e-5, grads within 1e-4 relative."""
    from biomedkg_amd import ops
    from oracle import pygcl_ops as G
    torch.manual_seed(0)
    n, h = 5000, 128
    h1, h2 = torch.randn(n, h), torch.randn(n, h)
    a, b = h1.clone().requires_grad_(True), h2.clone().requires_grad_(True)
    ref = G.dual_branch_l2l_intraview(a, b, 0.2)
    ref.backward()
    c, d = h1.to(DEV).requires_grad_(True), h2.to(DEV).requires_grad_(True)
    loss = ops.ntxent_loss(c, d, 0.2)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), ref.detach(), **LOSS_TOL)
    torch.testing.assert_close(c.grad.cpu(), a.grad, rtol=1e-3, atol=1e-7)
    torch.testing.assert_close(d.grad.cpu(), b.grad, rtol=1e-3, atol=1e-7)


def test_negative_sampling_properties_on_device():
    from biomedkg_amd.sampling import negative_sampling
    torch.manual_seed(1)
    n = 300
    ei = torch.randint(0, n, (2, 5000), device=DEV)
    neg = negative_sampling(ei, num_neg_samples=15000)
    assert neg.shape == (2, 15000)
    nn = int(ei.max()) + 1
    pos = set((ei[0] * nn + ei[1]).tolist())
    lin = (neg[0] * nn + neg[1]).tolist()
    assert not (set(lin) & pos) and len(set(lin)) == len(lin)
    assert not bool((neg[0] == neg[1]).any())
    full = torch.cartesian_prod(torch.arange(3), torch.arange(3)).t().contiguous().to(DEV)
    assert negative_sampling(full).shape == (2, 0)
