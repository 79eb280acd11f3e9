#!/bin/bash
# rocprofv3 kernel stats of the driver's whole default bench command (headline + the six secondary workloads) on the round's
# last library: one csv whose per-kernel averages stand next to the `kernel_ms` of every entry of the bench line.
export TMPDIR=/tmp
R=${1:-r03x}
mkdir -p gpurun_out
timeout 170 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/${R}_prof -o all -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${R}_bench_stdout.txt 2> gpurun_out/${R}_bench.err
echo "rc=$?" > gpurun_out/${R}_rc.txt
tail -1 gpurun_out/${R}_bench_stdout.txt > gpurun_out/${R}_bench_line_under_rocprof.json
f=$(find gpurun_out/${R}_prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -E '^"Name"|hta::' "$f" > gpurun_out/${R}_all_kernel_stats.csv
rm -rf gpurun_out/${R}_prof
