#!/bin/bash
# A/B of an environment switch on one box: scripts/ab_env.sh NAME "VAR=a" "VAR=b" [reps] -> gpurun_out/ab_NAME.txt
# (interleaved runs of bench.py --no-extras --no-cpu-baseline, 20 timed steps each; ms_per_step and the sub-step times)
name=$1; a=$2; b=$3; reps=${4:-3}
out=gpurun_out/ab_$name.txt; : > $out
for i in $(seq $reps); do
  for v in "$a" "$b"; do
    line=$(env $v python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --detail /tmp/ab_detail.json 2>/dev/null | tail -1)
    echo "$v $(echo "$line" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["ms_per_step_median"], d["ms_gcl_step"], d["ms_kge_step"])')" >> $out
  done
done
cat $out
