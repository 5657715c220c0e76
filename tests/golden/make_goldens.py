#!/usr/bin/env python3
"""Generate golden vectors from the UNMODIFIED reference files (authoring container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

Puts ``oracle/shims`` (stand-ins for torch_geometric / GCL / lightning / torchmetrics /
omegaconf, built on the oracle's restatements) and ``/root/reference`` on ``sys.path``, imports
the reference's own ``biomedkg`` package, runs it on CPU over small seeded inputs, and records
inputs + state_dict + the tape of random draws + outputs (loss, embeddings, scores, grads) into
``tests/golden/*.pt``.  Only these data files are committed; no reference source or bytecode is.

What this pins: the reference's own arithmetic (decoders, fusion modules) and its composition /
loss wiring (GRACE/GGD/DGI forward, encoder stacking, KGE training_step).  What it cannot pin:
the arithmetic inside torch_geometric / PyGCL (restated in ``oracle/pyg_ops.py`` /
``oracle/pygcl_ops.py`` -- parity unpinned, see ``oracle/__init__.py``).
"""
import importlib.util
import os
import random
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [os.path.join(ROOT, "oracle", "shims"), ROOT, REF]

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import pyg_ops, rand as orand  # noqa: E402


class Batch:
    pass


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


class patched_rng:
    """Route the torch RNG entry points the reference calls directly through the oracle provider."""

    def __enter__(self):
        self.saved = (torch.rand, torch.randperm, F.dropout)
        torch.rand = lambda *a, **k: orand.current().coin() if a == (1,) else self.saved[0](*a, **k)
        torch.randperm = lambda n, **k: orand.current().perm(n)
        F.dropout = lambda x, p=0.5, training=True, inplace=False: pyg_ops.dropout(x, p, training)
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randperm, F.dropout = self.saved
        return False


def graph(n, e, seed, with_self_loops=True):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    if with_self_loops:  # exercise add_remaining_self_loops: a couple of (duplicate) self loops
        ei[:, 0] = torch.tensor([3, 3])
        ei[:, 1] = torch.tensor([3, 3])
        ei[:, 2] = torch.tensor([5, 5])
    return ei


def lm_features(n, m, d, seed):
    """[n, m, d] features normalised across the modality axis (reference biomedkg/data/node.py:116-117)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, m, d, generator=g)
    return x / x.norm(dim=1, keepdim=True)


def grads_of(module):
    return {k: p.grad.clone() for k, p in module.named_parameters() if p.grad is not None}


def save(name, obj):
    path = os.path.join(HERE, name + ".pt")
    torch.save(obj, path)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


# --------------------------------------------------------------------------- decoders (direct)
def gen_decoders():
    from biomedkg.model.decoder import DistMult, TransE

    out = {}
    for name, cls in (("dismult", DistMult), ("transe", TransE)):
        seed_all(1)
        n, d, r, t = 37, 24, 5, 150
        dec = cls(num_relations=r, hidden_channels=d)
        z = torch.randn(n, d, requires_grad=True)
        with torch.no_grad():
            z[4].zero_()  # TransE eps path: an all-zero row
        ei = torch.randint(0, n, (2, t))
        ei[0, 0] = 4
        et = torch.randint(0, r, (t,))
        score = dec(z, ei, et)
        gout = torch.randn(t)
        score.backward(gout)
        out[name] = dict(z=z.detach().clone(), edge_index=ei, edge_type=et, rel_emb=dec.rel_emb.detach().clone(),
                         score=score.detach().clone(), grad_out=gout, grad_z=z.grad.clone(),
                         grad_rel=dec.rel_emb.grad.clone())
    save("decoders", out)


# --------------------------------------------------------------------------- fusion (direct)
def gen_fusion():
    from biomedkg.utils.fusion import AttentionFusion, ReDAF

    out = {}
    seed_all(2)
    n, d = 29, 48
    x = lm_features(n, 2, d, 3)
    att = AttentionFusion(embed_dim=d)
    out["attention"] = dict(x=x, state=att.state_dict(), y=att(x).detach().clone())
    red = ReDAF(embed_dim=d)
    with torch.no_grad():
        red.modal_weights.uniform_(0.5, 1.5)
    red.eval()
    out["redaf_eval"] = dict(x=x, state=red.state_dict(), y=red(x).detach().clone())
    red.train()
    tape = orand.TapeRand()
    saved = red.dropout.forward
    red.dropout.forward = lambda t: pyg_ops.dropout(t, 0.1, True)
    with orand.use(tape):
        y = red(x)
    red.dropout.forward = saved
    out["redaf_train"] = dict(x=x, state=red.state_dict(), y=y.detach().clone(), tape=tape.tape)
    save("fusion", out)


# --------------------------------------------------------------------------- GCL modules
def gen_gcl():
    from biomedkg.gcl_module import DGIModule, GGDModule, GRACEModule

    out = {}
    n, e, d_in, h = 48, 220, 32, 16
    cases = [
        ("grace_attention", GRACEModule, "attention", 11),
        ("grace_none", GRACEModule, "none", 12),
        ("ggd_redaf", GGDModule, "redaf", 13),
        ("ggd_none_a", GGDModule, "none", 14),
        ("ggd_none_b", GGDModule, "none", 15),
        ("dgi_attention", DGIModule, "attention", 16),
        ("dgi_none", DGIModule, "none", 17),
    ]
    coins = set()
    for name, cls, fuse, seed in cases:
        seed_all(seed)
        mod = cls(in_dim=d_in, hidden_dim=h, out_dim=h, num_hidden_layers=2, fuse_method=fuse)
        mod.train()
        if fuse == "redaf":  # route nn.Dropout through the tape as well
            drop = mod.modality_transform.dropout
            drop.forward = lambda t, drop=drop: pyg_ops.dropout(t, 0.1, drop.training)
        for p in mod.model.encoder.graph_layers:  # non-zero biases so the +bias term is exercised
            with torch.no_grad():
                p.bias.uniform_(-0.1, 0.1)
        b = Batch()
        b.x = lm_features(n, 2, d_in, seed)
        b.edge_index = graph(n, e, seed)
        tape = orand.TapeRand()
        with patched_rng(), orand.use(tape):
            loss = mod.training_step(b)
        loss.backward()
        with torch.no_grad():
            mod.eval()
            z_eval = mod(b.x, b.edge_index)  # BaseGCL.forward (gcl_module.py:55-58), dropout off
            mod.train()
        if name.startswith("ggd"):
            coins.add(bool(tape.tape[1 if fuse == "redaf" else 0][1].item() < 0.5))
        out[name] = dict(kind=cls.__name__, fuse_method=fuse, hparams=dict(in_dim=d_in, hidden_dim=h, out_dim=h,
                         num_hidden_layers=2), x=b.x, edge_index=b.edge_index,
                         state=mod.state_dict(), tape=tape.tape, loss=loss.detach().clone(),
                         grads=grads_of(mod.model), z_eval=z_eval)
    assert coins == {True, False}, f"GGD cases must cover both coin outcomes, got {coins}"
    save("gcl_steps", out)


# --------------------------------------------------------------------------- KGE module
def gen_kge():
    from biomedkg.kge_module import KGEModule

    out = {}
    n, e, r, h = 40, 180, 4, 16
    cases = [
        ("rgcn_dismult_random", "dismult", "random", "none", 32, 1, None, 21),
        ("rgcn_dismult_neg3", "dismult", "random", "none", 32, 3, None, 22),
        ("rgcn_transe_lm_attention", "transe", "lm", "attention", 32, 2, None, 23),
        ("rgcn_transe_gcl_fixid", "transe", "gcl", "none", 16, 2, 1, 24),
    ]
    for name, dec, init, fuse, d_in, neg_ratio, fix_id, seed in cases:
        seed_all(seed)
        mod = KGEModule(encoder_name="rgcn", decoder_name=dec, in_dim=d_in, hidden_dim=h, out_dim=h,
                        num_hidden_layers=2, num_relation=r, num_heads=2, scheduler_type="cosine",
                        learning_rate=1e-3, warm_up_ratio=0.2, fuse_method=fuse, neg_ratio=neg_ratio,
                        node_init_method=init)
        mod.train()
        for p in mod.model.encoder.graph_layers:
            with torch.no_grad():
                p.bias.uniform_(-0.1, 0.1)
        if fix_id is not None:
            mod.fix_edge_id = fix_id
        b = Batch()
        if init == "lm":
            b.x = lm_features(n, 2, d_in, seed)
        elif init == "gcl":
            b.x = torch.randn(n, 1, d_in)
        else:
            b.x = torch.randn(n, d_in)
        b.edge_index = graph(n, e, seed)
        b.edge_type = torch.randint(0, r, (e,))
        b.edge_type[b.edge_type == 2] = 3  # relation 2 is empty: mean over nothing -> 0
        edge_type_in = b.edge_type.clone()
        rec = {}
        enc, dec_fn = mod.model.encode, mod.model.decode

        def encode(*a):
            rec["z"] = enc(*a)
            rec["z"].retain_grad()
            return rec["z"]

        def decode(*a):
            rec.setdefault("pred", []).append(dec_fn(*a))
            return rec["pred"][-1]

        mod.model.encode, mod.model.decode = encode, decode
        tape = orand.TapeRand()
        with patched_rng(), orand.use(tape):
            loss = mod.training_step(b)
        loss.backward()
        mod.model.encode, mod.model.decode = enc, dec_fn
        out[name] = dict(hparams=dict(encoder_name="rgcn", decoder_name=dec, in_dim=d_in, hidden_dim=h, out_dim=h,
                         num_hidden_layers=2, num_relation=r, num_heads=2, scheduler_type="cosine",
                         learning_rate=1e-3, warm_up_ratio=0.2, fuse_method=fuse, neg_ratio=neg_ratio,
                         node_init_method=init), fix_edge_id=fix_id, x=b.x, edge_index=b.edge_index,
                         edge_type=edge_type_in, state={k: v for k, v in mod.state_dict().items()},
                         tape=tape.tape, loss=loss.detach().clone(), z=rec["z"].detach().clone(),
                         grad_z=rec["z"].grad.clone(), pos_pred=rec["pred"][0].detach().clone(),
                         neg_pred=rec["pred"][1].detach().clone(), grads=grads_of(mod.model))
    save("kge_steps", out)


# --------------------------------------------------------------------------- checkpoint (BASELINE configs[4])
def gen_ckpt():
    """A checkpoint as Lightning 2.2.3 lays it out, written from the reference's OWN ``KGEModule`` (rgcn + transe,
    node_init_method = lm, fuse_method = attention -- the DPI fine-tune's starting point, reference train_dpi.py:24-38):
    ``state_dict`` is the module's own (encoder, decoder, the frozen fuser, and the torchmetrics states the module
    registers once ``edge_mapping`` is set), ``hyper_parameters`` what ``save_hyperparameters`` recorded
    (kge_module.py:37).  Beside it: a DPI-shaped evaluation graph, the reference module's embeddings on it with every
    relation id forced to ``fix_edge_id = 1`` (train_dpi.py:33-35, kge_module.py:112-113) and the filtered MRR / Hits@k
    of those embeddings by the dense fp64 checker (oracle/ranking.py)."""
    from biomedkg.kge_module import KGEModule
    from oracle import ranking

    seed_all(31)
    n, e, r, h, d_in = 60, 150, 4, 16, 32
    hp = dict(encoder_name="rgcn", decoder_name="transe", in_dim=d_in, hidden_dim=h, out_dim=h, num_hidden_layers=2,
              num_relation=r, num_heads=2, scheduler_type="cosine", learning_rate=1e-3, warm_up_ratio=0.2,
              fuse_method="attention", neg_ratio=2, node_init_method="lm")
    mod = KGEModule(**hp)
    mod.edge_mapping = {0: "drug_drug", 1: "drug_protein", 2: "protein_protein", 3: "disease_protein"}
    for layer in mod.model.encoder.graph_layers:
        with torch.no_grad():
            layer.bias.uniform_(-0.1, 0.1)
    # a few optimizer steps, so that the saved weights are a trained state and not an initialisation pattern
    b = Batch()
    b.x, b.edge_index = lm_features(n, 2, d_in, 31), graph(n, e, 31, with_self_loops=False)
    b.edge_type = torch.randint(0, r, (e,))
    opt = torch.optim.Adam(mod.model.parameters(), lr=5e-3)
    mod.train()
    with patched_rng(), orand.use(orand.TapeRand()):
        for _ in range(5):
            opt.zero_grad()
            mod.training_step(b).backward()
            opt.step()
    mod.eval()
    ckpt = {"epoch": 4, "global_step": 5, "pytorch-lightning_version": "2.2.3",
            "state_dict": {k: v.detach().clone() for k, v in mod.state_dict().items()},
            "hyper_parameters": dict(vars(mod.hparams)), "loops": {}, "callbacks": {}, "optimizer_states": [],
            "lr_schedulers": []}
    torch.save(ckpt, os.path.join(HERE, "kge_reference.ckpt"))
    # DPI-shaped evaluation: bipartite drug -> protein edges made undirected (data_module.py ToUndirected), relation 1
    g = torch.Generator().manual_seed(77)
    drugs, prots = torch.arange(0, 25), torch.arange(25, n)
    src = drugs[torch.randint(0, 25, (70,), generator=g)]
    dst = prots[torch.randint(0, n - 25, (70,), generator=g)]
    pairs = torch.unique(torch.stack([src, dst]), dim=1)
    ei = torch.cat([pairs, pairs.flip(0)], dim=1)
    et = torch.randint(0, r, (ei.size(1),), generator=g)            # as loaded: mixed ids, overwritten by fix_edge_id
    x = lm_features(n, 2, d_in, 32)
    fixed = torch.full_like(et, 1)
    with torch.no_grad():
        z = mod(x, ei, fixed)
    held = pairs[:, ::3]                                            # the queries: every third drug -> protein pair
    q_et = torch.ones(held.size(1), dtype=torch.int64)
    want = ranking.transe_ranking(z, mod.model.decoder.rel_emb.detach(), held, q_et, known=(ei, fixed))
    save("kge_ckpt_eval", dict(hparams=hp, x=x, edge_index=ei, edge_type=et, fix_edge_id=1, z=z.clone(),
                               eval_edge_index=held, eval_edge_type=q_et, ranking=want,
                               state_keys=sorted(ckpt["state_dict"].keys())))


# --------------------------------------------------------------------------- metric
def gen_metric():
    from biomedkg.utils.metrics import EdgeWisePrecision

    seed_all(5)
    mapping = {0: "a_b", 1: "b_c", 2: "c_d"}
    m = EdgeWisePrecision(class_mapping=mapping)
    preds = torch.randn(200) + 0.5
    target = torch.randint(0, 2, (200,))  # class 2 never appears -> 0.0
    m.update(preds[:120], target[:120])
    m.update(preds[120:], target[120:])
    save("edge_wise_precision", dict(mapping=mapping, preds=preds, target=target, result=m.compute()))


# --------------------------------------------------------------------------- graph construction
def gen_triplets():
    """The reference's TripletBase over a small table (loaded by file path: its package __init__ pulls the
    dataset downloaders); HeteroData comes from the shim."""
    import pandas as pd
    spec = importlib.util.spec_from_file_location("ref_base", os.path.join(REF, "biomedkg/data/dataset/_base.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = [
        ("gene/protein", "TP53", "protein_protein", "gene/protein", "MDM2"),
        ("gene/protein", "MDM2", "protein_protein", "gene/protein", "TP53"),
        ("drug", "aspirin", "drug_protein", "gene/protein", "PTGS2"),
        ("gene/protein", "PTGS2", "drug_protein", "drug", "aspirin"),      # reverse direction: dropped (triples[0])
        ("drug", "ibuprofen", "drug_protein", "gene/protein", "PTGS2"),
        ("disease", "pain", "indication", "drug", "aspirin"),
        ("drug", "aspirin", "indication", "disease", "pain"),              # reverse direction: dropped
        ("disease", "fever", "indication", "drug", "ibuprofen"),
        ("drug", "aspirin", "drug_drug", "drug", "ibuprofen"),
        ("drug", "ibuprofen", "drug_drug", "drug", "aspirin"),
        ("disease", "pain", "disease_protein", "gene/protein", "TP53"),
        ("gene/protein", "BRCA1", "protein_protein", "gene/protein", "TP53"),
        ("drug", "aspirin", "drug_protein", "gene/protein", "PTGS2"),     # duplicate row: kept as a parallel edge
    ]
    df = pd.DataFrame(rows, columns=["x_type", "x_name", "relation", "y_type", "y_name"])

    class Enc:
        random_init_ratio = 0

        def __call__(self, names):
            return torch.tensor([[float(sum(map(ord, n))), float(len(n))] for n in names])

    tb = mod.TripletBase(df=df, encoder=Enc())
    save("triplets", dict(rows=rows, x=tb.data.x, edge_index=tb.data.edge_index, edge_type=tb.data.edge_type,
                          node_type=tb.data.node_type, edge_map_index=tb.edge_map_index, node_list=tb.node_list))


# --------------------------------------------------------------------------- data modules
def gen_data_modules():
    """The reference's own ``PrimeKGModule`` / ``DPIModule`` (``biomedkg/data_module.py``, unmodified) set up over the toy
    tables ``scripts/make_toy_kg.py`` writes: pandas and the reference's ``TripletBase`` / ``PrimeKG`` / ``DPI`` /
    ``RandomEncode`` are the real thing; ``HeteroData.to_homogeneous``, ``RandomLinkSplit`` and ``ToUndirected`` come from
    the shims (oracle/shims/torch_geometric: PyG restated).  Recorded: the graph, the features and the three splits under
    a fixed seed.  The tables travel with the fixture as ``tests/golden/toy_kg.csv`` / ``toy_dpi.csv`` (data written by
    the committed script, which the test re-runs and compares)."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import make_toy_kg
    work = tempfile.mkdtemp(prefix="bkg_dm_")
    cwd = os.getcwd()
    try:
        os.makedirs(os.path.join(work, "ckpt", "gcl"))          # (node.py's GCLEncode asserts on it at import time)
        kg = os.path.join(work, "data", "primekg", "kg.csv")
        dpi = os.path.join(work, "data", "dpi", "dpi_benchmark.csv")
        make_toy_kg.write_kg_csv(kg)
        make_toy_kg.write_dpi_csv(dpi)
        shutil.copy(kg, os.path.join(HERE, "toy_kg.csv"))
        shutil.copy(dpi, os.path.join(HERE, "toy_dpi.csv"))
        os.chdir(work)
        from biomedkg import data_module as ref_dm
        out = {}
        pick = lambda d: dict(x=d.x.clone(), edge_index=d.edge_index.clone(), edge_type=d.edge_type.clone())   # noqa: E731
        for name, make in (
            ("primekg", lambda: ref_dm.PrimeKGModule(data_dir="data/primekg", embed_dim=16, batch_size=8, val_ratio=0.2,
                                                     test_ratio=0.1, node_type=["gene/protein", "drug", "disease"],
                                                     node_init_method="random")),
            ("primekg_gene", lambda: ref_dm.PrimeKGModule(data_dir="data/primekg", embed_dim=8, batch_size=8, val_ratio=0.25,
                                                          test_ratio=0.25, node_type=["gene/protein"], node_init_method=None)),
            ("dpi", lambda: ref_dm.DPIModule(data_dir="data/dpi/dpi_benchmark.csv", embed_dim=16, batch_size=8, val_ratio=0.2,
                                             test_ratio=0.2, node_init_method="random")),
        ):
            seed_all(11)
            dm = make()
            dm.setup(stage="split")
            ds = dm.primekg if name.startswith("primekg") else dm.dpi
            rec = dict(data=pick(dm.data), node_type=dm.data.node_type.clone(), num_edge_types=dm.data.num_edge_types,
                       edge_map_index=dict(dm.edge_map_index), node_list=list(ds.node_list), seed=11)
            for split in ("train", "val", "test"):
                d = getattr(dm, split + "_data")
                rec[split] = dict(edge_index=d.edge_index.clone(), edge_type=d.edge_type.clone(),
                                  edge_label=d.edge_label.clone(), edge_label_index=d.edge_label_index.clone())
            out[name] = rec
        save("data_modules", out)
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("the reference is only mounted in the authoring container")
    gen_decoders()
    gen_fusion()
    gen_gcl()
    gen_kge()
    gen_ckpt()
    gen_metric()
    gen_triplets()
    gen_data_modules()
    # the reference tree must stay clean
    assert not any("__pycache__" in d for d, _, _ in os.walk(REF)), "left bytecode in /root/reference"
