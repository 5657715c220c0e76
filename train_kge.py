#!/usr/bin/env python3
"""KGE training entry point with the reference's flow and override grammar (reference ``train_kge.py``;
e.g. ``python train_kge.py devices=[0] data=synthetic neg_ratio=1 model.decoder_name=dismult debug=true``).
``train_dpi.py`` differs only in the config name and the optional checkpoint warm start."""
import os
import sys
import time

from biomedkg_amd.compat import HAVE_LIGHTNING
from biomedkg_amd.config import compose, instantiate
from biomedkg_amd.kge_module import KGEModule

if HAVE_LIGHTNING:  # pragma: no cover
    from lightning.pytorch import Trainer, seed_everything
    from lightning.pytorch.callbacks import ModelCheckpoint
else:
    from biomedkg_amd.trainer import ModelCheckpoint, Trainer, seed_everything


def build(cfg, config_name):
    data_module = instantiate(cfg.data, gcl_model=cfg.gcl_model, gcl_fuse_method=cfg.gcl_fuse_method)
    data_module.setup(stage="split")
    pretrained = cfg.get("pretrained_path") or ""
    if config_name == "dpi" and pretrained.endswith(".ckpt"):   # reference train_dpi.py:24-36
        model = KGEModule.load_from_checkpoint(pretrained)
        model.fix_edge_id = 1
        model.neg_ratio = cfg.neg_ratio
    else:
        model = KGEModule(**cfg.model, num_relation=data_module.data.num_edge_types, neg_ratio=cfg.neg_ratio,
                          node_init_method=cfg.data.node_init_method)
    model.edge_mapping = data_module.edge_map_index
    return data_module, model


def main(overrides, config_name="kge"):
    cfg = compose(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs"), config_name, overrides)
    seed_everything(cfg.seed)
    data_module, model = build(cfg, config_name)
    tag = f"{cfg.model.encoder_name}_{cfg.model.decoder_name}_{cfg.data.node_init_method}"
    if cfg.data.node_init_method == "gcl":
        tag += f"_{cfg.gcl_model}_{cfg.gcl_fuse_method}"
    ckpt_dir = os.path.join(cfg.ckpt_dir, config_name, tag + str(int(time.time())))
    os.makedirs(ckpt_dir, exist_ok=True)
    trainer = Trainer(accelerator="auto", log_every_n_steps=10, deterministic=True, devices=cfg.devices,
                      fast_dev_run=bool(cfg.debug), max_epochs=cfg.epochs, check_val_every_n_epoch=cfg.val_every_epoch,
                      enable_checkpointing=True, gradient_clip_val=1.0, default_root_dir=ckpt_dir,
                      callbacks=[ModelCheckpoint(dirpath=ckpt_dir, monitor="val_loss", save_top_k=3, mode="min",
                                                 save_last=True)])
    trainer.fit(model=model, train_dataloaders=data_module.train_dataloader(loader_type="saint"),
                val_dataloaders=data_module.val_dataloader(loader_type="saint"))
    test_args = {"model": model, "dataloaders": data_module.test_dataloader(loader_type="saint")}
    if not cfg.debug:
        test_args["ckpt_path"] = "best"
    return trainer.test(**test_args)


if __name__ == "__main__":
    main(sys.argv[1:])
