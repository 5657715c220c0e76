export TMPDIR=/tmp
rm -rf /tmp/kf
rocprofv3 --kernel-trace --output-format csv -d /tmp/kf -o k -- python3 scripts/dev_fresh_steps.py 5
cp $(find /tmp/kf -name "*kernel_trace.csv" | head -1) gpurun_out/fresh_kernel_trace.csv
