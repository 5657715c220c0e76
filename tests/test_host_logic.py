"""Host-side logic that needs no GPU: synthetic graph shapes, the LightningModule stand-in
(hyper-parameters, checkpoint round trip, reference key layout), module construction."""
import os

import torch


def test_primekg_like_shape_and_skew():
    from biomedkg_amd import synthetic
    g = synthetic.primekg_like(13_000, 400_000, 30, seed=1)
    ei, et = g["edge_index"], g["edge_type"]
    assert ei.shape == (2, 400_000) and et.shape == (400_000,) and ei.dtype == torch.int64
    assert int(ei.max()) < 13_000 and int(et.max()) == g["num_relations"] - 1 == 29
    share = torch.bincount(et, minlength=30).float() / et.numel()
    assert 0.30 < share[0] < 0.45 and 0.25 < share[1] < 0.40 and share[-1] < 1e-3      # skewed relation histogram
    indeg = torch.bincount(ei[1], minlength=13_000)
    assert indeg.max() > 50 * indeg.float().median().clamp(min=1)                       # hubs
    g2 = synthetic.primekg_like(13_000, 400_000, 30, seed=1)
    assert torch.equal(g2["edge_index"], ei)                                            # seeded


def test_gcl_small_graph():
    from biomedkg_amd import synthetic
    ei = synthetic.gcl_small(500, 5_000, seed=2)
    assert ei.shape == (2, 5_000) and (ei[0] != ei[1]).all()
    assert torch.unique(ei[0] * 500 + ei[1]).numel() == 5_000
    x = synthetic.lm_features(10, 2, 16)
    torch.testing.assert_close(x.norm(dim=1), torch.ones(10, 16))


def test_module_key_layout_matches_reference(golden):
    """state_dict keys of the modules == the keys recorded from the reference's own modules."""
    from biomedkg_amd import gcl_module as M
    from biomedkg_amd.kge_module import KGEModule
    for name, g in golden("gcl_steps").items():
        mod = getattr(M, g["kind"])(**g["hparams"], fuse_method=g["fuse_method"])
        assert set(mod.state_dict()) == set(g["state"]), name
    for name, g in golden("kge_steps").items():
        assert set(KGEModule(**g["hparams"]).state_dict()) == set(g["state"]), name


def test_hparams_and_checkpoint_round_trip(tmp_path):
    from biomedkg_amd.gcl_module import GGDModule
    from biomedkg_amd.kge_module import KGEModule
    m = GGDModule(in_dim=32, hidden_dim=16, out_dim=16, num_hidden_layers=2, fuse_method="redaf")
    hp = dict(m.hparams)
    assert hp["in_dim"] == 32 and hp["fuse_method"] == "redaf" and hp["embed_dim"] == 32 and "model" not in hp
    path = os.path.join(tmp_path, "m.ckpt")
    keep = {k: v for k, v in hp.items() if not isinstance(v, torch.nn.Module)}
    torch.save({"state_dict": m.state_dict(), "hyper_parameters": keep}, path)
    m2 = GGDModule.load_from_checkpoint(path)
    for k, v in m.state_dict().items():
        assert torch.equal(m2.state_dict()[k], v)
    k = KGEModule("rgcn", "transe", 32, 16, 16, 2, 5, 2, "cosine", 1e-3, 0.2, "attention", 3, "lm")
    assert k.modality_transform is not None and k.neg_ratio == 3 and k.fix_edge_id is None
    k.fix_edge_id = 1
    k.edge_mapping = {0: "a", 1: "b"}
    assert k.fix_edge_id == 1 and hasattr(k, "edge_wise_pre_valid") and k.edge_mapping[1] == "b"
    assert torch.allclose(k.model.decoder.rel_emb.norm(dim=1), torch.ones(5))      # TransE init: row-normalised


def test_factory_names():
    from biomedkg_amd.factory import FusionFactory, KGEModelFactory
    assert FusionFactory.create_fuser("none", 8) is None
    assert type(FusionFactory.create_fuser("attention", 8)).__name__ == "AttentionFusion"
    model = KGEModelFactory.get_model("rgcn", "dismult", 8, 8, 8, 1, 3, 2)
    assert type(model.decoder).__name__ == "DistMult" and len(model.encoder.graph_layers) == 3
    assert KGEModelFactory._get_decoder("nope", 3, 8) is None     # reference returns None silently


def test_metrics_small_known_answers():
    from biomedkg_amd.utils.metrics import AUROC, AveragePrecision, F1Score
    p = torch.tensor([0.9, 0.8, 0.7, 0.6, 0.55, 0.4])
    t = torch.tensor([1, 1, 0, 1, 0, 0])
    for cls, exp in ((AUROC, 8 / 9), (AveragePrecision, (1 + 1 + 0.75) / 3), (F1Score, 2 * 3 / (2 * 3 + 2 + 0))):
        m = cls(task="binary")
        m.update(p[:3], t[:3]); m.update(p[3:], t[3:])
        assert abs(float(m.compute()) - exp) < 1e-6, cls.__name__


def test_link_split_message_passing_graphs():
    """reference data_module.py:65-69: RandomLinkSplit(num_val, num_test, neg_sampling_ratio=0) -- the modules only read
    ``edge_index`` / ``edge_type`` of each split, i.e. the MESSAGE-PASSING graphs: train = the train edges,
    val = the train edges again, test = train + val edges; relation ids travel with their edges; the three edge sets
    (train, val, test supervision) partition the input."""
    from biomedkg_amd.data_module import SyntheticKGModule
    dm = SyntheticKGModule(embed_dim=16, node_type=["gene", "drug"], val_ratio=0.2, test_ratio=0.1, num_nodes=300,
                           num_edges=4000, num_relations=5, seed=3)
    dm.setup("split")
    from biomedkg_amd import synthetic
    g = synthetic.primekg_like(300, 4000, 5, seed=3)
    ei, et = g["edge_index"], g["edge_type"]
    e = ei.size(1)
    n_val, n_test = int(0.2 * e), int(0.1 * e)
    tr, te = dm.train_data, dm.test_data
    assert tr.edge_index.size(1) == e - n_val - n_test and te.edge_index.size(1) == e - n_test
    assert dm.val_data is dm.train_data                                  # val message passing = the train graph
    key = lambda a, t: set(zip(a[0].tolist(), a[1].tolist(), t.tolist(), range(a.size(1))))  # multiset via position
    full = sorted(zip(ei[0].tolist(), ei[1].tolist(), et.tolist()))
    tr_k = sorted(zip(tr.edge_index[0].tolist(), tr.edge_index[1].tolist(), tr.edge_type.tolist()))
    te_k = sorted(zip(te.edge_index[0].tolist(), te.edge_index[1].tolist(), te.edge_type.tolist()))
    from collections import Counter
    cf, ct, cte = Counter(full), Counter(tr_k), Counter(te_k)
    assert not (ct - cf) and not (cte - cf)                              # every split edge is an input edge, with its relation
    assert not (ct - cte)                                                # test message passing contains the train graph
    assert sum((cf - cte).values()) == n_test                            # what is left out of it is the test supervision set
    assert torch.equal(te.edge_index[:, : tr.edge_index.size(1)], tr.edge_index)   # train edges first, then val
    assert dm.data.num_edge_types == 5 and len(dm.edge_map_index) == 5


def test_edge_bucket_choice_of_the_padded_mini_batch_loop():
    """biomedkg_amd/sampled.py: which captured graph a GraphSAINT batch replays under -- and when it must not."""
    from biomedkg_amd.sampled import EDGE_BUCKETS, SaintDeviceSampler, choose_bucket
    b = (1024, 2048, 4096)
    assert choose_bucket(b, 10, 900, 2700, 2690) == 4096 and choose_bucket(b, 10, 900, 1024, 1000) == 1024
    assert choose_bucket(b, 10, 900, 1025, 1000) == 2048
    assert choose_bucket(b, 10, 900, 4097, 4000) is None                 # past the ladder
    assert choose_bucket(b, 10, 1, 0, 0) is None and choose_bucket(b, 10, 5, 0, 0) is None
    # PyG's first round would not do: 1.1 k / prob reaches the population (it enumerates every cell then) ...
    assert choose_bucket(b, 10, 60, 400, 400) is None                    # 60 * 59 = 3540 cells, 4000 negatives asked for
    # ... or the batch is dense (prob <= 0.9: later rounds could be needed)
    assert choose_bucket(b, 1, 100, 1100, 1000) is None                  # 1000 / 9900 loop-free positives > 10 %
    assert choose_bucket(b, 1, 100, 900, 900) == 1024                    # 9.1 % dense, 900 negatives: margin 90 >= 6 sigma = 57
    # ... or so FEW negatives are asked for that the 10 % over-sampling margin of that one round is under six sigma
    assert choose_bucket(b, 1, 60, 320, 320) is None and choose_bucket(b, 1, 60, 30, 30) is None
    assert choose_bucket(b, 10, 60, 30, 30) == 1024 and choose_bucket(b, 10, 400, 50, 50) == 1024
    assert list(EDGE_BUCKETS) == sorted(EDGE_BUCKETS) and all(y / x <= 1.51 for x, y in zip(EDGE_BUCKETS, EDGE_BUCKETS[1:]))
    # the flat batch buffer: sizes | n_id | src | dst | type | eid, disjoint and complete
    import torch
    buf = torch.arange(3 + 7 + 4 * 5)
    v = SaintDeviceSampler.views(buf, 7, 5)
    assert [k for k in v] == ["sizes", "n_id", "src", "dst", "type", "eid"]
    assert torch.equal(torch.cat([v[k] for k in v]), buf)


def test_reference_lightning_checkpoint_loads_strictly():
    """tests/golden/kge_reference.ckpt: written from the reference's own KGEModule in Lightning's layout
    (tests/golden/make_goldens.py gen_ckpt).  Strict load, hyper-parameters restored, extra torchmetrics states (which a
    checkpoint of the real module may carry) ignored, a missing parameter refused."""
    import os
    import pytest
    from biomedkg_amd.kge_module import KGEModule
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kge_reference.ckpt")
    m = KGEModule.load_from_checkpoint(path, strict=True)
    ckpt = torch.load(path, weights_only=False)
    for k, v in ckpt["state_dict"].items():
        assert torch.equal(m.state_dict()[k], v), k
    assert m.hparams["decoder_name"] == "transe" and m.hparams["node_init_method"] == "lm" and m.neg_ratio == 2
    assert type(m.modality_transform).__name__ == "AttentionFusion"


def test_checkpoint_with_metric_states_or_missing_weights(tmp_path):
    import os
    import pytest
    from biomedkg_amd.kge_module import KGEModule
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kge_reference.ckpt")
    ckpt = torch.load(src, weights_only=False)
    extra = dict(ckpt, state_dict=dict(ckpt["state_dict"], **{"valid_metrics.BinaryAUROC.metrics.0.preds": torch.zeros(3),
                                                               "edge_wise_pre_valid.true_positive": torch.zeros(4)}))
    torch.save(extra, tmp_path / "extra.ckpt")
    KGEModule.load_from_checkpoint(str(tmp_path / "extra.ckpt"), strict=True)
    short = dict(ckpt, state_dict={k: v for k, v in ckpt["state_dict"].items() if k != "model.decoder.rel_emb"})
    torch.save(short, tmp_path / "short.ckpt")
    with pytest.raises(RuntimeError):
        KGEModule.load_from_checkpoint(str(tmp_path / "short.ckpt"), strict=True)


def test_gcl_edge_bucket_ladder():
    """biomedkg_amd/sampled_gcl.py: geometric capacities (< 25 % padding), capped by the most edges a batch can hold."""
    from biomedkg_amd.sampled_gcl import edge_buckets
    b = edge_buckets(640_000)
    assert b[-1] == 640_000 and b[0] == 4096 and list(b) == sorted(set(b))
    assert all(v % 1024 == 0 for v in b[:-1])
    assert all(hi <= 1.25 * lo + 1024 for lo, hi in zip(b, b[1:]))
    assert edge_buckets(1000) == (1000,) and edge_buckets(5000) == (4096, 5000)


def test_constant_input_table_is_remembered_only_in_a_training_pass():
    """partition.Comm remembers the gathered table of a CONSTANT layer input (the node features of the first R-GCN layer)
    on the graph structure.  Under torch.no_grad() every hidden activation looks constant too: nothing may be remembered
    then (it pinned a table per layer and evicted the features' one), while a table a training pass remembered
    is still used; different pattern keys do not evict each other."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from cpu_backend import TorchBackend as B
    from biomedkg_amd import partition as P, synthetic
    from oracle import ref_modules as R
    g = synthetic.primekg_like(120, 900, 4, seed=2)
    ei, et = g["edge_index"], g["edge_type"]
    part = P.NodePartition(ei, 120, 0, 1)
    prg = P.PartitionedRelGraph(part, ei, et, g["num_relations"], B)
    torch.manual_seed(0)
    enc = R.make_kge_model("rgcn", "dismult", 12, 8, 8, 1, g["num_relations"]).encoder.eval()      # three conv layers, no dropout
    x = torch.randn(120, 12)
    assert P.Comm.is_constant_input((False, True, True, True)) and not P.Comm.is_constant_input((True, True, True, True))
    assert not P.Comm.is_constant_input((False, False, False, False))
    with torch.no_grad():
        P.rgcn_encoder_partitioned(enc, x, prg, B)
    assert not getattr(prg, "_bkg_const_tables", None)                     # an eval pass remembers nothing
    out = P.rgcn_encoder_partitioned(enc, x, prg, B)
    tables = prg._bkg_const_tables
    assert list(tables) == ["in"] and tables["in"][0] is x                 # the features' table, and only that
    kept = tables["in"][2]
    out.sum().backward()
    with torch.no_grad():
        out2 = P.rgcn_encoder_partitioned(enc, x, prg, B)                  # the eval pass USES it and adds nothing
    assert list(prg._bkg_const_tables) == ["in"] and prg._bkg_const_tables["in"][2] is kept
    torch.testing.assert_close(out2, out.detach())
    part.comm.remember_table(prg, x, ("in", 1), kept)                      # the fused layers' key lives beside it
    assert part.comm.constant_table(prg, x) is kept and part.comm.constant_table(prg, x, ("in", 1)) is kept
    x.add_(1.0)                                                            # a new version of the tensor: stale
    assert part.comm.constant_table(prg, x) is None


def test_trainer_moves_a_repeated_batch_once():
    """A whole-graph loader hands out the SAME host batch every step; the Trainer stand-in copies it to the device once and
    reuses the copy while the object and its tensors stay unchanged (a fresh copy per step costs a full-size KGE step 170 MB
    of PCIe and every graph structure rebuilt on new tensors); another object, or a tensor written in place, is moved again."""
    from biomedkg_amd import trainer as T

    class B:
        moves = 0

        def __init__(self):
            self.x, self.edge_index = torch.zeros(4, 2), torch.zeros(2, 3, dtype=torch.long)

        def to(self, device):
            B.moves += 1
            out = B.__new__(B)
            out.x, out.edge_index = self.x.to(device), self.edge_index.to(device)
            return out

    tr = T.Trainer(max_epochs=1)
    a, b = B(), B()
    m1, m2 = tr._move(a), tr._move(a)
    assert m1 is m2 and B.moves == 1
    assert tr._move(b) is not m1 and B.moves == 2 and tr._move(a) is m1 and B.moves == 2     # (train / val batches alternate)
    a.x.add_(1)                                               # written in place: the copy is stale
    m3 = tr._move(a)
    assert m3 is not m1 and B.moves == 3 and float(m3.x[0, 0]) == 1.0
    a.x = torch.ones(4, 2)                                    # another tensor under the same name
    assert tr._move(a) is not m3 and B.moves == 4
    assert tr._move(torch.zeros(2)) is not None               # plain tensors are moved as before
