"""The oracle's restatement of the reference's own arithmetic vs. the golden vectors recorded
from the UNMODIFIED reference files (tests/golden/make_goldens.py).  CPU only."""
import pytest
import torch

from oracle import rand as orand
from oracle import ref_modules as R

TOL = dict(rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["dismult", "transe"])
def test_decoders(golden, name):
    g = golden("decoders")[name]
    z = g["z"].clone().requires_grad_(True)
    rel = g["rel_emb"].clone().requires_grad_(True)
    fn = R.distmult_score if name == "dismult" else R.transe_score
    s = fn(z, g["edge_index"], g["edge_type"], rel)
    torch.testing.assert_close(s, g["score"], **TOL)
    s.backward(g["grad_out"])
    torch.testing.assert_close(z.grad, g["grad_z"], **TOL)
    torch.testing.assert_close(rel.grad, g["grad_rel"], **TOL)


def test_fusion(golden):
    g = golden("fusion")
    a = R.AttentionFusion(48)
    a.load_state_dict(g["attention"]["state"])
    torch.testing.assert_close(a(g["attention"]["x"]), g["attention"]["y"], **TOL)
    r = R.ReDAF(48)
    r.load_state_dict(g["redaf_eval"]["state"])
    r.eval()
    torch.testing.assert_close(r(g["redaf_eval"]["x"]), g["redaf_eval"]["y"], **TOL)
    r.train()
    with orand.use(orand.ReplayRand(g["redaf_train"]["tape"])):
        torch.testing.assert_close(r(g["redaf_train"]["x"]), g["redaf_train"]["y"], **TOL)


def build_gcl(g):
    hp = g["hparams"]
    enc = R.GCNEncoder(hp["in_dim"], hp["hidden_dim"], hp["out_dim"], hp["num_hidden_layers"])
    h = hp["hidden_dim"]
    model = {"GRACEModule": lambda: R.GRACE(enc, h, h), "GGDModule": lambda: R.GGD(enc, h, 1, 0.5),
             "DGIModule": lambda: R.DGI(enc, h)}[g["kind"]]()
    fuser = R.create_fuser(g["fuse_method"], hp["in_dim"])
    state = g["state"]
    model.load_state_dict({k[len("model."):]: v for k, v in state.items() if k.startswith("model.")})
    if fuser is not None:
        fuser.load_state_dict({k[len("modality_transform."):]: v for k, v in state.items()
                               if k.startswith("modality_transform.")})
    loss_fn = {"GRACEModule": R.grace_loss, "GGDModule": R.ggd_loss, "DGIModule": R.dgi_loss}[g["kind"]]
    return model, fuser, loss_fn


GCL_CASES = ["grace_attention", "grace_none", "ggd_redaf", "ggd_none_a", "ggd_none_b", "dgi_attention", "dgi_none"]


@pytest.mark.parametrize("name", GCL_CASES)
def test_gcl_step(golden, name):
    g = golden("gcl_steps")[name]
    model, fuser, loss_fn = build_gcl(g)
    model.train()
    if fuser is not None:
        fuser.train()
    rp = orand.ReplayRand(g["tape"])
    with orand.use(rp):
        loss = loss_fn(model, R.fusion_fn(fuser, g["x"]), g["edge_index"])
    assert rp.pos == len(g["tape"])
    torch.testing.assert_close(loss, g["loss"], **TOL)
    loss.backward()
    for k, v in g["grads"].items():
        torch.testing.assert_close(dict(model.named_parameters())[k].grad, v, rtol=1e-4, atol=1e-6, msg=k)
    model.eval()
    if fuser is not None:
        fuser.eval()
    with torch.no_grad():
        torch.testing.assert_close(model.encoder(R.fusion_fn(fuser, g["x"]), g["edge_index"]), g["z_eval"], **TOL)


KGE_CASES = ["rgcn_dismult_random", "rgcn_dismult_neg3", "rgcn_transe_lm_attention", "rgcn_transe_gcl_fixid"]


@pytest.mark.parametrize("name", KGE_CASES)
def test_kge_step(golden, name):
    g = golden("kge_steps")[name]
    hp = g["hparams"]
    model = R.make_kge_model(hp["encoder_name"], hp["decoder_name"], hp["in_dim"], hp["hidden_dim"], hp["out_dim"],
                             hp["num_hidden_layers"], hp["num_relation"])
    model.load_state_dict({k[len("model."):]: v for k, v in g["state"].items() if k.startswith("model.")})
    fuser = R.create_fuser(hp["fuse_method"], hp["in_dim"]) if hp["node_init_method"] == "lm" else None
    if fuser is not None:
        fuser.load_state_dict({k[len("modality_transform."):]: v for k, v in g["state"].items()
                               if k.startswith("modality_transform.")})
    model.train()
    et = g["edge_type"] if g["fix_edge_id"] is None else torch.full_like(g["edge_type"], g["fix_edge_id"])
    rp = orand.ReplayRand(g["tape"])
    with orand.use(rp):
        loss, z, pos, neg = R.kge_step(model, R.fusion_fn(fuser, g["x"]), g["edge_index"], et, hp["neg_ratio"])
    assert rp.pos == len(g["tape"])
    torch.testing.assert_close(loss, g["loss"], **TOL)
    torch.testing.assert_close(z, g["z"], **TOL)
    torch.testing.assert_close(pos, g["pos_pred"], **TOL)
    torch.testing.assert_close(neg, g["neg_pred"], **TOL)
    loss.backward()
    for k, v in g["grads"].items():
        torch.testing.assert_close(dict(model.named_parameters())[k].grad, v, rtol=1e-4, atol=1e-6, msg=k)


def test_edge_wise_precision(golden):
    g = golden("edge_wise_precision")
    got = R.edge_wise_precision(g["preds"], g["target"], len(g["mapping"]))
    for c, name in g["mapping"].items():
        assert abs(got[c] - g["result"][name + "_pre"]) < 1e-7


def test_reference_checkpoint_embeddings_and_ranking(golden):
    """BASELINE configs[4] ("MRR vs reference ckpt"): tests/golden/kge_reference.ckpt was written from the reference's
    own KGEModule in Lightning's layout (make_goldens.py gen_ckpt).  The oracle, loaded from that state_dict, reproduces
    the reference module's embeddings on the DPI-shaped graph with fix_edge_id = 1, and the dense fp64 ranking checker
    its recorded filtered MRR / Hits@k."""
    import os
    from oracle import ranking
    from tests.conftest import GOLDEN
    g = golden("kge_ckpt_eval")
    ckpt = torch.load(os.path.join(GOLDEN, "kge_reference.ckpt"), weights_only=False)
    hp = ckpt["hyper_parameters"]
    assert hp == g["hparams"] and sorted(ckpt["state_dict"]) == g["state_keys"]
    assert {"state_dict", "hyper_parameters", "epoch", "global_step", "pytorch-lightning_version"} <= set(ckpt)
    model = R.make_kge_model(hp["encoder_name"], hp["decoder_name"], hp["in_dim"], hp["hidden_dim"], hp["out_dim"],
                             hp["num_hidden_layers"], hp["num_relation"])
    model.load_state_dict({k[len("model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("model.")})
    fuser = R.create_fuser(hp["fuse_method"], hp["in_dim"])
    fuser.load_state_dict({k[len("modality_transform."):]: v for k, v in ckpt["state_dict"].items()
                           if k.startswith("modality_transform.")})
    model.eval()
    et = torch.full_like(g["edge_type"], g["fix_edge_id"])
    with torch.no_grad():
        z = model.encode(R.fusion_fn(fuser, g["x"]), g["edge_index"], et)
    torch.testing.assert_close(z, g["z"], **TOL)
    got = ranking.transe_ranking(z, model.decoder.rel_emb.detach(), g["eval_edge_index"], g["eval_edge_type"],
                                 known=(g["edge_index"], et))
    for k, v in g["ranking"].items():
        assert abs(got[k] - v) <= 1e-9, (k, got[k], v)
