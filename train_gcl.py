#!/usr/bin/env python3
"""GCL training entry point with the reference's flow and override grammar
(reference ``train_gcl.py``; e.g. ``python train_gcl.py devices=[0] data=synthetic data.node_type=gene
model.model_name=grace model.fuse_method=attention data.node_init_method=lm debug=true``)."""
import os
import sys
import time

from biomedkg_amd.compat import HAVE_LIGHTNING
from biomedkg_amd.config import compose, instantiate
from biomedkg_amd.gcl_module import DGIModule, GGDModule, GRACEModule

if HAVE_LIGHTNING:  # pragma: no cover
    from lightning.pytorch import Trainer, seed_everything
    from lightning.pytorch.callbacks import EarlyStopping, ModelCheckpoint
else:
    from biomedkg_amd.trainer import EarlyStopping, ModelCheckpoint, Trainer, seed_everything

MODELS = {"dgi": DGIModule, "grace": GRACEModule, "ggd": GGDModule}


def create_gcl_model(cfg):
    if cfg.model_name not in MODELS:
        raise NotImplementedError
    keys = ("in_dim", "hidden_dim", "out_dim", "num_hidden_layers", "scheduler_type", "learning_rate", "warm_up_ratio",
            "fuse_method")
    return MODELS[cfg.model_name](**{k: cfg[k] for k in keys})


def main(overrides):
    cfg = compose(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs"), "gcl", overrides)
    seed_everything(cfg.seed)
    node_type = cfg.data.node_type
    if isinstance(node_type, list) and len(node_type) > 1:
        raise ValueError("Please select only one node type")
    if isinstance(node_type, list):
        node_type = node_type[0]
    tag = f"{cfg.model.model_name}_{cfg.model.fuse_method}_{cfg.data.node_init_method}_{int(time.time())}"
    ckpt_dir = os.path.join(cfg.ckpt_dir, "gcl", node_type.replace("/", "_"), tag)
    cfg.data.node_type = ["gene/protein"] if node_type.startswith("gene") else [node_type]

    data_module = instantiate(cfg.data)
    data_module.setup(stage="split")
    model = create_gcl_model(cfg.model)

    os.makedirs(ckpt_dir, exist_ok=True)
    trainer = Trainer(accelerator="auto", log_every_n_steps=10, deterministic=True, devices=cfg.devices,
                      fast_dev_run=bool(cfg.debug), max_epochs=cfg.epochs, check_val_every_n_epoch=cfg.val_every_epoch,
                      enable_checkpointing=True, gradient_clip_val=1.0, default_root_dir=ckpt_dir,
                      callbacks=[ModelCheckpoint(dirpath=ckpt_dir, monitor="val_loss", save_top_k=1, mode="min"),
                                 EarlyStopping(monitor="val_loss", mode="min", patience=5)])
    trainer.fit(model=model, train_dataloaders=data_module.train_dataloader(loader_type="neighbor"),
                val_dataloaders=data_module.val_dataloader(loader_type="neighbor"))
    test_args = {"model": model, "dataloaders": data_module.test_dataloader()}
    if not cfg.debug:
        test_args["ckpt_path"] = "best"
    return trainer.test(**test_args)


if __name__ == "__main__":
    main(sys.argv[1:])
