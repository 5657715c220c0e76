#!/bin/bash
# Regenerates the judged profile set of a build on the GPU box (run from the repo root through gpurun):
#   scripts/final_profiles.sh <tag>        e.g. r03_v21      -> gpurun_out/<tag>_*   (copy what is kept into profiles/)
# rocprofv3 gets the program itself after `--` (python3 ...), counters in passes of their own (no trace options beside --pmc).
set -u
tag=${1:-rXX}
out=gpurun_out
export TMPDIR=/tmp
mkdir -p $out
# a first bench line: its tags' grids are what the counter rows are attributed by (the line that is KEPT is taken last, below)
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --detail $out/${tag}_bench_detail.json > $out/${tag}_bench.json 2> $out/${tag}_bench.err
for prec in bf16 fp32; do
  rm -rf /tmp/ks_$prec
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$prec -o k -- python3 bench.py --precision $prec --steps 5 --warmup 2 \
      --no-cpu-baseline --no-extras --detail /tmp/ks_detail_$prec.json > $out/${tag}_bench_under_rocprof_$prec.json 2>/dev/null
  cp $(find /tmp/ks_$prec -name "*kernel_stats.csv" | head -1) $out/${tag}_${prec}_kernel_stats.csv
done
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_$name -o p -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline \
      --detail /tmp/pmc_detail_$name.json > /tmp/pmc_$name.json 2>/dev/null
done
# (the kernel trace of the bf16 --kernel-trace --stats run above gives the per-(kernel, grid) durations and the step object)
python3 scripts/pmc_summary.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv") $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv") \
    $out/${tag}_pmc_traffic.json bf16 $out/${tag}_bench_detail.json $(find /tmp/pmc_TCC_HIT_sum -name "*counter_collection.csv") \
    $(find /tmp/ks_bf16 -name "*kernel_trace.csv" | head -1) 2 5 > $out/${tag}_pmc_summary.txt 2>&1
# the bench line of this build against ITS OWN counter passes (bench.py reads the newest profiles/*pmc_traffic.json)
cp $out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $out/${tag}_bench_detail.json > $out/${tag}_bench.json 2> $out/${tag}_bench.err
# the sampled GCL regime as replayed graphs: kernel stats and the busy fraction of the trace
rm -rf /tmp/gclg
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gclg -o g -- python3 scripts/dev_gcl_graphed_loop.py fp32 100 1 > $out/${tag}_gcl_graphed.txt 2>/dev/null
python3 scripts/busy_fraction.py $(find /tmp/gclg -name "*kernel_trace.csv") >> $out/${tag}_gcl_graphed.txt 2>&1
cp $(find /tmp/gclg -name "*kernel_stats.csv" | head -1) $out/${tag}_gcl_graphed_kernel_stats.csv
# NT-Xent: kernel times and the matrix-pipe / clock counters
rm -rf /tmp/nxk /tmp/nxp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nxk -o k -- python3 scripts/prof_ntxent.py 130000 256 bf16 3 > /dev/null 2>&1
cp $(find /tmp/nxk -name "*kernel_stats.csv" | head -1) $out/${tag}_ntxent_N130000_h256_bf16_kernel_stats.csv
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d /tmp/nxp -o p -- python3 scripts/prof_ntxent.py 130000 256 bf16 3 > /dev/null 2>&1
python3 scripts/pmc_sq.py $(find /tmp/nxp -name "*counter_collection.csv") ntx2 > $out/${tag}_ntxent_pmc_sq.txt 2>&1
ls -la $out | head -40
