# Dev: kernel families of the timed steps of `bench.py --mode partition` at one rank (what every rank of an N-rank run launches)
export TMPDIR=/tmp
rm -rf /tmp/kp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kp -o k -- python3 bench.py --mode partition --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1
python3 - <<'PY'
import glob, json, sys
sys.path.insert(0, "scripts")
import pmc_summary as P
g, step = P.load_trace(glob.glob("/tmp/kp/**/*kernel_trace.csv", recursive=True)[0], 2, 5)
print(json.dumps({k: step[k] for k in ("kernel_ms_per_step", "launches_per_step", "library_share_of_kernel_time", "by_family_ms_per_step", "not_library_kernels")}, indent=0))
PY
cp $(find /tmp/kp -name "*kernel_trace.csv" | head -1) gpurun_out/partition_kernel_trace.csv 2>/dev/null
