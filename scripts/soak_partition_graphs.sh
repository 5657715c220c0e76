#!/bin/bash
# Soak: the graphed node-partitioned sub-steps (captured RCCL collectives, deferred KGE update) on a one-rank RCCL group,
# N runs in a row -- every run must end with exit code 0, three captured graphs and its deferred updates counted.
#   bash scripts/soak_partition_graphs.sh [runs=30]
runs=${1:-30}; ok=0; bad=0
for i in $(seq 1 $runs); do
  BKG_PARTITION_FORCE_COLLECTIVES=1 BKG_BENCH_PARTITION_GRAPH=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29700+i)) \
    python3 bench.py --gpus 1 --mode partition --steps 8 --warmup 2 --no-extras --no-cpu-baseline --nodes 20000 --edges 300000 \
    > /tmp/spg_$i.json 2> /tmp/spg_$i.err
  rc=$?
  if [ $rc -eq 0 ] && grep -q '"captured": 3' /tmp/spg_$i.json && grep -q '"deferred_updates"' /tmp/spg_$i.json; then ok=$((ok+1)); else bad=$((bad+1)); echo "run $i rc=$rc"; tail -4 /tmp/spg_$i.err; fi
done
echo "ok=$ok bad=$bad"
