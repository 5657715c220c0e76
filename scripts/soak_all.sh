#!/bin/bash
# Every fuzzer / soak of scripts/ once, each under its own time limit: gpurun_out/soak_all.txt holds "name rc seconds" lines and
# the tail of any that failed.  scripts/soak_all.sh  (MI355X)
out=gpurun_out/soak_all.txt; mkdir -p gpurun_out; : > $out
run() { name=$1; shift; t0=$(date +%s); timeout 600 "$@" > /tmp/soak_$name.log 2>&1; rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s" >> $out
        if [ $rc -ne 0 ]; then tail -25 /tmp/soak_$name.log >> $out; else tail -2 /tmp/soak_$name.log >> $out; fi; }
run fuzz_kernels python3 scripts/fuzz_kernels.py
run fuzz_kge_bwd python3 scripts/fuzz_kge_bwd.py
run fuzz_partition python3 scripts/fuzz_partition.py
run fuzz_rings python3 scripts/fuzz_rings.py
run fuzz_sampled python3 scripts/fuzz_sampled.py
run fuzz_sampled_gcl python3 scripts/fuzz_sampled_gcl.py
run fuzz_steps python3 scripts/fuzz_steps.py
run fuzz_tiny python3 scripts/fuzz_tiny.py
run fuzz_views_frames python3 scripts/fuzz_views_frames.py
run fuzz_kge_mx python3 scripts/fuzz_kge_mx.py
run soak_rings python3 scripts/soak_rings.py 60
cat $out
