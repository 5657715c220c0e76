"""The data modules over real triplet tables (reference biomedkg/data_module.py:27-145, 148-259 and
data/dataset/_base.py:23-89): ``PrimeKGModule`` / ``DPIModule`` built here against a fixture recorded from the reference's
OWN classes (tests/golden/make_goldens.py gen_data_modules: unmodified data_module.py / TripletBase / RandomEncode over
the toy tables of scripts/make_toy_kg.py; to_homogeneous / RandomLinkSplit / ToUndirected from the PyG shims) -- same
seed, same graph, same features, same splits, bit for bit."""
import os
import pickle
import shutil
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture()
def tables(tmp_path):
    os.makedirs(tmp_path / "primekg")
    shutil.copy(os.path.join(GOLDEN, "toy_kg.csv"), tmp_path / "primekg" / "kg.csv")
    shutil.copy(os.path.join(GOLDEN, "toy_dpi.csv"), tmp_path / "dpi.csv")
    return tmp_path


def test_the_committed_tables_are_what_the_committed_script_writes(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import make_toy_kg
    make_toy_kg.write_kg_csv(str(tmp_path / "kg.csv"))
    make_toy_kg.write_dpi_csv(str(tmp_path / "dpi.csv"))
    assert (tmp_path / "kg.csv").read_bytes() == open(os.path.join(GOLDEN, "toy_kg.csv"), "rb").read()
    assert (tmp_path / "dpi.csv").read_bytes() == open(os.path.join(GOLDEN, "toy_dpi.csv"), "rb").read()


def _same(dm, ds, rec):
    d = dm.data
    assert torch.equal(d.x, rec["data"]["x"])                              # RandomEncode: the same draws in the same order
    assert torch.equal(d.edge_index, rec["data"]["edge_index"]) and torch.equal(d.edge_type, rec["data"]["edge_type"])
    assert torch.equal(d.node_type, rec["node_type"]) and d.num_edge_types == rec["num_edge_types"]
    assert d.num_nodes == rec["data"]["x"].size(0)
    assert dm.edge_map_index == rec["edge_map_index"] and ds.node_list == rec["node_list"]
    for split in ("train", "val", "test"):
        got, want = getattr(dm, split + "_data"), rec[split]
        for key in ("edge_index", "edge_type", "edge_label", "edge_label_index"):
            assert torch.equal(getattr(got, key), want[key]), (split, key)
        assert got.x is d.x and got.num_edge_types == d.num_edge_types


@pytest.mark.parametrize("case", ["primekg", "primekg_gene"])
def test_primekg_module_matches_the_reference(golden, tables, case):
    from biomedkg_amd.data_module import PrimeKGModule
    rec = golden("data_modules")[case]
    kw = (dict(embed_dim=16, val_ratio=0.2, test_ratio=0.1, node_type=["gene/protein", "drug", "disease"], node_init_method="random")
          if case == "primekg" else dict(embed_dim=8, val_ratio=0.25, test_ratio=0.25, node_type=["gene/protein"], node_init_method=None))
    torch.manual_seed(rec["seed"])
    dm = PrimeKGModule(data_dir=str(tables / "primekg"), batch_size=8, **kw)
    dm.setup(stage="split")
    _same(dm, dm.primekg, rec)
    if case == "primekg":
        # one direction per relation survives, the two extra node types are filtered out
        assert dm.data.edge_index.size(1) == 463 and dm.data.num_edge_types == 7 and dm.data.num_nodes == 119
        e = dm.data.edge_index.size(1)
        assert dm.train_data.edge_index.size(1) == e - int(0.2 * e) - int(0.1 * e)
        assert dm.test_data.edge_index.size(1) == e - int(0.1 * e)
    # stage != "split": the graph only
    dm2 = PrimeKGModule(data_dir=str(tables / "primekg"), batch_size=8, **kw)
    dm2.setup(stage="all")
    assert not hasattr(dm2, "train_data") and dm2.data.num_nodes == dm.data.num_nodes


def test_dpi_module_matches_the_reference(golden, tables):
    from biomedkg_amd.data_module import DPIModule
    rec = golden("data_modules")["dpi"]
    torch.manual_seed(rec["seed"])
    dm = DPIModule(data_dir=str(tables / "dpi.csv"), embed_dim=16, batch_size=8, val_ratio=0.2, test_ratio=0.2,
                   node_init_method="random")
    dm.setup(stage="split")
    _same(dm, dm.dpi, rec)
    ei = dm.data.edge_index
    assert torch.equal(ei.flip(0)[:, torch.argsort(ei[1] * dm.data.num_nodes + ei[0])], ei)     # undirected, coalesced
    assert int(dm.data.edge_type.max()) == 0                                                    # one relation: sums stay 0


def test_dpi_module_refuses_relation_ids_the_merge_pushed_out_of_range(tmp_path):
    """ToUndirected(reduce="add") sums ``edge_type`` over merged duplicate pairs.  On a table with TWO relations
    in which a pair is listed TWICE under relation 1 (TripletBase keeps duplicate rows) the merged id is 2 = num_edge_types:
    PyG's layers would raise an IndexError on it; the device kernels index their relation tables unchecked, so the module
    raises up front."""
    import pandas as pd
    from biomedkg_amd.data_module import DPIModule
    rows = [("drug", f"d{i}", "binds", "gene/protein", f"p{i % 5}") for i in range(12)]
    rows += [("drug", "d0", "inhibits", "gene/protein", "p1"), ("drug", "d0", "inhibits", "gene/protein", "p1")]
    path = tmp_path / "dpi2.csv"
    pd.DataFrame(rows, columns=["x_type", "x_name", "relation", "y_type", "y_name"]).to_csv(path, index=False)
    torch.manual_seed(0)
    dm = DPIModule(data_dir=str(path), embed_dim=8, batch_size=4, val_ratio=0.2, test_ratio=0.2, node_init_method="random")
    with pytest.raises(IndexError, match="num_edge_types"):
        dm.setup(stage="split")


def test_missing_tables_are_errors_that_say_where_the_file_goes(tmp_path):
    from biomedkg_amd.data_module import DPIModule, PrimeKGModule
    with pytest.raises(FileNotFoundError, match="kg.csv"):
        PrimeKGModule(data_dir=str(tmp_path), embed_dim=8, node_type=["drug"], batch_size=4, val_ratio=0.2, test_ratio=0.2).setup()
    with pytest.raises(FileNotFoundError, match="DPI"):
        DPIModule(data_dir=str(tmp_path / "nope.csv"), embed_dim=8, batch_size=4, val_ratio=0.2, test_ratio=0.2).setup()


def test_the_reference_configs_instantiate_verbatim(tables):
    """configs/data/primekg.yaml keeps the reference's ``_target_: biomedkg.data_module.PrimeKGModule``: it resolves to this
    build's class (the default command line raised ModuleNotFoundError)."""
    from biomedkg_amd import config
    from biomedkg_amd.data_module import DPIModule, PrimeKGModule
    cfg = config.compose(os.path.join(ROOT, "configs"), "kge", [f"data.data_dir={tables / 'primekg'}", "data.embed_dim=8"])
    assert cfg.data["_target_"] == "biomedkg.data_module.PrimeKGModule"
    dm = config.instantiate(cfg.data, gcl_model=cfg.gcl_model, gcl_fuse_method=cfg.gcl_fuse_method)
    assert isinstance(dm, PrimeKGModule)
    dm.setup(stage="split")
    assert dm.data.num_edge_types == 7 and dm.train_data.edge_index.size(1) > 0
    cfg = config.compose(os.path.join(ROOT, "configs"), "dpi", [f"data.data_dir={tables / 'dpi.csv'}"])
    assert isinstance(config.instantiate(cfg.data, gcl_model=cfg.gcl_model, gcl_fuse_method=cfg.gcl_fuse_method), DPIModule)


def test_pickled_node_features_lm_and_gcl(tables, monkeypatch):
    """``lm`` / ``gcl`` node features come from the reference's pickles ({name: array}); a name that is not in the
    dictionary gets a random block and counts into ``random_init_ratio``; a missing pickle is an error that names it."""
    from biomedkg_amd.data_module import PrimeKGModule
    monkeypatch.chdir(tables)
    kw = dict(data_dir=str(tables / "primekg"), embed_dim=6, node_type=["drug"], batch_size=4, val_ratio=0.2, test_ratio=0.2)
    with pytest.raises(FileNotFoundError, match="primekg_modality_lm.pickle"):
        PrimeKGModule(node_init_method="lm", **kw)
    import numpy as np
    os.makedirs(tables / "data" / "embed")
    table = {f"drug-{i:03d}": np.full((2, 6), float(i), dtype=np.float64) for i in range(1, 40, 2)}
    pickle.dump(table, open(tables / "data" / "embed" / "primekg_modality_lm.pickle", "wb"))
    dm = PrimeKGModule(node_init_method="lm", **kw)
    dm.setup(stage="split")
    x, names = dm.data.x, dm.primekg.node_list
    assert x.shape == (len(names), 2, 6) and x.dtype == torch.float32
    for i, name in enumerate(names):
        if name in table:
            assert torch.equal(x[i], torch.full((2, 6), float(int(name[-3:]))))
    assert 0 < dm.encoder.random_init_ratio < 1
    os.makedirs(tables / "data" / "gcl_embed")
    pickle.dump({n: np.ones((1, 6), dtype=np.float32) for n in names[:5]}, open(tables / "data" / "gcl_embed" / "ggd_redaf.pickle", "wb"))
    dm = PrimeKGModule(node_init_method="gcl", gcl_model="ggd", gcl_fuse_method="redaf", **kw)
    dm.setup(stage="all")
    assert dm.data.x.shape == (len(names), 1, 6) and torch.equal(dm.data.x[:5], torch.ones(5, 1, 6))


def test_star_loader_is_the_one_hop_in_neighbourhood(tables):
    from biomedkg_amd.data_module import PrimeKGModule
    dm = PrimeKGModule(data_dir=str(tables / "primekg"), embed_dim=4, node_type=["gene/protein", "drug"], batch_size=4,
                       val_ratio=0.2, test_ratio=0.2)
    dm.setup(stage="all")
    loader = dm.subgraph_dataloader()
    assert len(loader) == dm.data.num_nodes
    ei, et = dm.data.edge_index, dm.data.edge_type
    for v, b in enumerate(loader):
        into = torch.nonzero(ei[1] == v).squeeze(1)
        assert b.batch_size == 1 and int(b.n_id[0]) == v and torch.equal(b.e_id, into)
        assert torch.equal(b.n_id[b.edge_index[0]], ei[0, into]) and bool((b.edge_index[1] == 0).all())
        assert torch.equal(b.edge_type, et[into]) and torch.equal(b.x, dm.data.x[b.n_id])
        assert b.n_id.unique().numel() == b.n_id.numel()
        if v > 25:
            break


@pytest.mark.gpu
def test_entry_points_run_on_the_tables_with_the_reference_command_line(tables, monkeypatch):
    """``python train_kge.py debug=true`` / ``train_gcl.py`` / ``train_dpi.py`` with the reference's own default data
    configs (only ``data_dir`` pointed at the toy tables and the widths cut down): PrimeKGModule / DPIModule -> device
    GraphSAINT / NeighborLoader batches -> the HIP training step."""
    import train_gcl
    import train_kge
    small = ["data.embed_dim=32", "model.in_dim=32", "model.hidden_dim=16", "model.out_dim=16", "debug=true",
             f"ckpt_dir={tables / 'ckpt'}", "devices=[0]"]
    out = train_kge.main([f"data.data_dir={tables / 'primekg'}", "neg_ratio=2"] + small)
    assert out and any(k.startswith("test_") for k in out[0])
    out = train_gcl.main([f"data.data_dir={tables / 'primekg'}", "data.node_type=gene", "model.model_name=ggd"] + small)
    assert "test_loss" in out[0]
    out = train_kge.main([f"data.data_dir={tables / 'dpi.csv'}", "neg_ratio=3"] + small, config_name="dpi")
    assert out and isinstance(out[0], dict)


@pytest.mark.gpu
def test_evaluation_entry_points_load_the_checkpoints_the_training_ones_write(tables):
    """The reference's ``test_kge.py`` / ``test_dpi.py`` (evaluation entry points, not unit tests; test_kge.py:9-44): a
    checkpoint written by ``train_kge.py`` / ``train_dpi.py`` is loaded with ``KGEModule.load_from_checkpoint`` (no
    arguments: the hyper-parameters travel in the file), ``neg_ratio`` and the relation names are set, and ``trainer.test``
    over the GraphSAINT test loader returns the bootstrapped AUROC / AP / F1 and the per-relation precision."""
    import test_dpi  # noqa: F401  (importable: it only forwards to test_kge.main)
    import test_kge
    import train_kge
    small = ["data.embed_dim=32", "model.in_dim=32", "model.hidden_dim=16", "model.out_dim=16", "epochs=1", "val_every_epoch=1",
             f"ckpt_dir={tables / 'ckpt'}", "devices=[0]"]
    for name, data_dir in (("kge", tables / "primekg"), ("dpi", tables / "dpi.csv")):
        train_kge.main([f"data.data_dir={data_dir}", "neg_ratio=2"] + small, config_name=name)
        ckpts = sorted(os.path.join(d, f) for d, _, fs in os.walk(tables / "ckpt" / name) for f in fs if f.endswith(".ckpt"))
        assert ckpts
        out = test_kge.main([f"data.data_dir={data_dir}", "neg_ratio=3", f"pretrained_path={ckpts[-1]}",
                             "data.embed_dim=32", "devices=[0]"], config_name=name)
        assert out and any(k.startswith("test_") for k in out[0]), out
        assert all(v == v for v in out[0].values())                      # no NaN
    with pytest.raises(FileNotFoundError, match="pretrained_path"):
        test_kge.main([f"data.data_dir={tables / 'primekg'}", "pretrained_path=nope.ckpt", "data.embed_dim=32"])


@pytest.mark.gpu
def test_loaders_of_the_real_module_hand_out_device_batches(tables):
    from biomedkg_amd.data_module import PrimeKGModule
    torch.manual_seed(3)
    dm = PrimeKGModule(data_dir=str(tables / "primekg"), embed_dim=8, node_type=["gene/protein", "drug", "disease"],
                       batch_size=16, val_ratio=0.2, test_ratio=0.2)
    dm.setup(stage="split")
    loader = dm.train_dataloader(loader_type="neighbor")
    assert len(loader) == (dm.data.num_nodes + 15) // 16
    b = next(iter(loader))
    assert b.x.is_cuda and b.batch_size == 16 and b.edge_index.dtype == torch.int64 and b.edge_type.numel() == b.edge_index.size(1)
    tr = dm.train_data.edge_index.cuda()
    assert torch.equal(b.n_id[b.edge_index], tr[:, b.e_id])               # every batch edge is a train edge, relabelled
    saint = dm.train_dataloader(loader_type="saint")
    assert len(saint) == 1000 and len(dm.val_dataloader(loader_type="saint")) == 100
    b = next(iter(saint))
    assert torch.equal(b.n_id[b.edge_index], tr[:, b.e_id]) and torch.equal(b.edge_type, dm.train_data.edge_type.cuda()[b.e_id])


@pytest.mark.gpu
def test_three_stage_pipeline_gcl_checkpoints_feed_the_kge_stage(tables, monkeypatch):
    """The reference's pipeline B -> C on the toy tables with its own command lines: ``train_gcl.py`` once per node type
    (lm features, GGD + attention fuser) writes ``ckpt/gcl/{gene,drug,disease}/ggd_attention_lm_*/epoch=*.ckpt``; ``train_kge.py
    data.node_init_method=gcl gcl_model=ggd gcl_fuse_method=attention`` then finds no ``data/gcl_embed/ggd_attention.pickle``,
    builds it from those checkpoints (data_module.GCLEncode -> export.node_embeddings: the reference's per-node
    NeighborLoader([-1]) loop of data/node.py:196-241 in bulk) and trains on the ``[N, 1, D]`` embeddings."""
    import numpy as np
    import train_gcl
    import train_kge
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import make_toy_kg
    monkeypatch.chdir(tables)
    os.makedirs("data/primekg", exist_ok=True)
    shutil.copy(tables / "primekg" / "kg.csv", "data/primekg/kg.csv")
    os.makedirs("data/embed", exist_ok=True)
    rng = np.random.default_rng(0)
    lm = {name: rng.standard_normal((2, 64)).astype(np.float32)
          for kind in ("gene/protein", "drug", "disease") for name in make_toy_kg.node_names(kind, 40)}
    pickle.dump(lm, open("data/embed/primekg_modality_lm.pickle", "wb"))
    small = ["data.data_dir=data/primekg", "data.embed_dim=64", "model.in_dim=64", "model.hidden_dim=32", "model.out_dim=32",
             "model.model_name=ggd", "model.fuse_method=attention", "data.node_init_method=lm", "data.batch_size=16",
             "epochs=1", "val_every_epoch=1", "ckpt_dir=ckpt", "devices=[0]"]
    for nt in ("gene", "drug", "disease"):
        out = train_gcl.main([f"data.node_type={nt}"] + small)
        assert "test_loss" in out[0]
        assert [f for _, _, fs in os.walk(os.path.join("ckpt", "gcl", nt)) for f in fs if f.endswith(".ckpt")]
    out = train_kge.main(["data.data_dir=data/primekg", "data.node_init_method=gcl", "gcl_model=ggd", "gcl_fuse_method=attention",
                          "data.embed_dim=32", "model.in_dim=32", "model.hidden_dim=16", "model.out_dim=16", "neg_ratio=2",
                          "debug=true", "ckpt_dir=ckpt", "devices=[0]"])
    assert out and isinstance(out[0], dict)
    table = pickle.load(open("data/gcl_embed/ggd_attention.pickle", "rb"))
    from biomedkg_amd.data_module import PrimeKGModule
    dm = PrimeKGModule(data_dir="data/primekg", embed_dim=32, node_type=["gene/protein", "drug", "disease"], batch_size=8,
                       val_ratio=0.2, test_ratio=0.2, node_init_method="gcl", gcl_model="ggd", gcl_fuse_method="attention")
    dm.setup(stage="all")
    # (a node that has no edge inside its own type's graph is in no GCL stage: the reference gives it a random row too)
    names = set(dm.primekg.node_list)
    assert set(table) <= names and len(table) > 0.6 * len(names) and all(v.shape == (1, 32) for v in table.values())
    genes = {n for n in names if n.startswith("GENE")}
    assert len(genes & set(table)) >= len(genes) - 3                         # (nearly) every gene has a protein_protein edge here
    assert dm.data.x.shape == (dm.data.num_nodes, 1, 32) and dm.encoder.random_init_ratio < 0.5
    assert np.isfinite(np.stack(list(table.values()))).all()
