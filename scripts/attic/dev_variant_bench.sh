#!/bin/bash
# Dev: one short bench run, the step time and the tagged calls' launch times on one line (for scripts/dev_variants.sh)
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /tmp/vb.json 2>/dev/null
python3 scripts/dev_bench_keys.py /tmp/vb.json
