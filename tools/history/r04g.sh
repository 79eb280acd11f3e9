#!/bin/bash
export TMPDIR=/tmp
R=${1:-r04g}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rmhmc.py -q -x -k "statistical_parity" > gpurun_out/${R}_t2.txt 2>&1; tail -3 gpurun_out/${R}_t2.txt
( time timeout 1200 python bench.py ) > gpurun_out/${R}_bench_stdout.txt 2> gpurun_out/${R}_bench_stderr.txt
tail -c 4500 gpurun_out/${R}_bench_stdout.txt | tail -1; tail -5 gpurun_out/${R}_bench_stderr.txt
cp bench_detail.json gpurun_out/${R}_bench_detail.json 2>/dev/null
HTA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/${R}_gloo2.txt 2> gpurun_out/${R}_gloo2_err.txt; tail -1 gpurun_out/${R}_gloo2.txt; tail -3 gpurun_out/${R}_gloo2_err.txt
