#!/usr/bin/env python3
"""DPI fine-tune entry point (reference ``train_dpi.py``): the KGE flow on ``configs/dpi.yaml`` with an
optional ``pretrained_path=...ckpt`` warm start (``fix_edge_id = 1``, ``neg_ratio`` override)."""
import sys

from train_kge import main

if __name__ == "__main__":
    main(sys.argv[1:], config_name="dpi")
