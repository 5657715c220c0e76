#!/usr/bin/env python3
"""DPI evaluation entry point (reference ``test_dpi.py:9-46``): ``test_kge.py``'s flow on ``configs/dpi.yaml``."""
import sys

from test_kge import main

if __name__ == "__main__":
    main(sys.argv[1:], config_name="dpi")
