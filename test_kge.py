#!/usr/bin/env python3
"""Evaluation entry point (NOT a unit test; reference ``test_kge.py:9-44``): load a KGE checkpoint, set ``neg_ratio`` and the
relation names, run ``trainer.test`` over the GraphSAINT test loader -- AUROC / AP / F1 (bootstrapped) and the per-relation
precision, as ``KGEModule.test_step`` / ``on_test_epoch_end`` compute them on the device.
``python test_kge.py pretrained_path=ckpt/kge/.../last.ckpt neg_ratio=3 data.node_init_method=random data.data_dir=...``;
``test_dpi.py`` is the same flow on ``configs/dpi.yaml``."""
import os
import sys

from biomedkg_amd.compat import HAVE_LIGHTNING
from biomedkg_amd.config import compose, instantiate
from biomedkg_amd.kge_module import KGEModule

if HAVE_LIGHTNING:  # pragma: no cover
    from lightning.pytorch import Trainer, seed_everything
else:
    from biomedkg_amd.trainer import Trainer, seed_everything


def main(overrides, config_name="kge"):
    cfg = compose(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs"), config_name, overrides)
    seed_everything(cfg.seed)
    data_module = instantiate(cfg.data, gcl_model=cfg.gcl_model, gcl_fuse_method=cfg.gcl_fuse_method)
    data_module.setup(stage="split")
    pretrained = cfg.get("pretrained_path") or ""
    if not os.path.exists(pretrained):
        raise FileNotFoundError(f"pretrained_path={pretrained!r}: give the .ckpt a train_kge.py / train_dpi.py run wrote")
    print("=" * 20 + f"\nLoad from checkpoint: {pretrained}\n" + "=" * 20)
    model = KGEModule.load_from_checkpoint(pretrained)
    model.neg_ratio = cfg.neg_ratio
    model.edge_mapping = data_module.edge_map_index
    print("=" * 20 + f"\nNeg Ratio: {model.neg_ratio}\n" + "=" * 20)
    trainer = Trainer(accelerator="auto", log_every_n_steps=10, deterministic=True, devices=cfg.devices)
    return trainer.test(model=model, dataloaders=data_module.test_dataloader(loader_type="saint"))


if __name__ == "__main__":
    main(sys.argv[1:])
