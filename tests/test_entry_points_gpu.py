"""The reference-shaped entry points run end to end on the HIP path (debug = fast_dev_run, then a
2-epoch fit with validation, checkpointing and test-from-best).  Needs an MI355X."""
import os

import pytest

pytestmark = pytest.mark.gpu


def test_train_gcl_debug_all_models():
    import train_gcl
    for model, fuse, init in (("ggd", "attention", "lm"), ("grace", "redaf", "lm"), ("dgi", "none", "random")):
        out = train_gcl.main(["devices=[0]", "data=synthetic", "data.node_type=gene", f"data.node_init_method={init}",
                              "data.num_nodes=600", "data.num_edges=6000", "data.embed_dim=64", "model.in_dim=64",
                              "model.hidden_dim=32", "model.out_dim=32", f"model.model_name={model}",
                              f"model.fuse_method={fuse}", "debug=true", "ckpt_dir=/tmp/bkg_ckpt"])
        assert "test_loss" in out[0]


def test_train_kge_two_epochs_with_checkpoint(tmp_path):
    import train_kge
    out = train_kge.main(["devices=[0]", "data=synthetic", "data.num_nodes=800", "data.num_edges=8000",
                          "data.embed_dim=32", "model.in_dim=32", "model.hidden_dim=32", "model.out_dim=32",
                          "neg_ratio=2", "epochs=2", "val_every_epoch=1", f"ckpt_dir={tmp_path}"])
    assert any(k.startswith("test_AUROC") for k in out[0])
    files = [f for _, _, fs in os.walk(tmp_path) for f in fs]
    assert "last.ckpt" in files and any(f.startswith("epoch=") for f in files)


def test_train_dpi_transe_debug():
    import train_kge
    out = train_kge.main(["devices=[0]", "data=synthetic", "data.num_nodes=500", "data.num_edges=4000",
                          "data.node_init_method=lm", "data.embed_dim=32", "model.in_dim=32", "model.hidden_dim=32",
                          "model.out_dim=32", "model.decoder_name=transe", "model.fuse_method=attention", "neg_ratio=5",
                          "debug=true", "ckpt_dir=/tmp/bkg_ckpt"], config_name="dpi")
    assert out and isinstance(out[0], dict)


def test_train_scripts_with_device_samplers():
    """The reference's own regime: NeighborLoader batches for GCL, GraphSAINT random-walk batches for KGE."""
    import train_gcl
    import train_kge
    out = train_gcl.main(["devices=[0]", "data=synthetic", "data.node_type=gene", "data.sampler=device",
                          "data.num_nodes=2000", "data.num_edges=20000", "data.embed_dim=32", "data.batch_size=128",
                          "model.in_dim=32", "model.hidden_dim=16", "model.out_dim=16", "model.model_name=ggd",
                          "epochs=1", "val_every_epoch=1", "ckpt_dir=/tmp/bkg_ckpt"])
    assert "test_loss" in out[0]
    out = train_kge.main(["devices=[0]", "data=synthetic", "data.sampler=device", "data.num_nodes=3000",
                          "data.num_edges=40000", "data.embed_dim=32", "data.batch_size=64", "data.steps_per_epoch=20",
                          "model.in_dim=32", "model.hidden_dim=16", "model.out_dim=16", "neg_ratio=1", "epochs=1",
                          "val_every_epoch=1", "ckpt_dir=/tmp/bkg_ckpt"])
    assert any(k.startswith("test_AUROC") for k in out[0])


def test_train_gcl_grace_replays_a_hipgraph(capsys):
    """configs[1]-style run through the entry point: the whole-graph loader hands out one static batch, so the Trainer
    stand-in captures the GRACE step once and replays it; GGD (host-side augmentation coin) stays eager and says so."""
    import train_gcl
    from biomedkg_amd import graphs
    made = []
    orig = graphs.GraphedStep.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)
    graphs.GraphedStep.__init__ = spy
    try:
        common = ["devices=[0]", "data=synthetic", "data.node_type=gene", "data.node_init_method=lm", "data.num_nodes=1200",
                  "data.num_edges=9000", "data.embed_dim=64", "data.steps_per_epoch=12", "model.in_dim=64", "model.hidden_dim=32",
                  "model.out_dim=32", "model.fuse_method=attention", "epochs=2", "val_every_epoch=1", "ckpt_dir=/tmp/bkg_ckpt"]
        out = train_gcl.main(common + ["model.model_name=grace"])
        assert "test_loss" in out[0] and len(made) == 1 and made[0].replays == 2 * 12
        out = train_gcl.main(common + ["model.model_name=ggd"])
        assert "test_loss" in out[0] and len(made) == 1
        assert "eager steps" in capsys.readouterr().out
    finally:
        graphs.GraphedStep.__init__ = orig
