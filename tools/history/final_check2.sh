#!/bin/bash
# Whole -m gpu suite, then the cfg3 lines / kernel stats of the current routes; run on the GPU box via gpurun.
export TMPDIR=/tmp
R=${1:-r01j}
mkdir -p gpurun_out
timeout 240 python -m pytest tests -m gpu -x -q > gpurun_out/${R}_tests_all.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/${R}_tests_all.log
bash tools/ab_cfg3.sh ${R} 256:- 512:- 1024:- 2048:- 3072:- 3072:rmhmc_mfma4=0 4096:- 8192:-
python bench.py --workload cfg3 --chains 1024 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_cfg3_1024_bench.json 2>/dev/null
timeout 150 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/${R}_st -o x -- python bench.py --workload cfg3 --chains 1024 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp $(find gpurun_out/${R}_st -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_cfg3_1024_kernel_stats.csv; rm -rf gpurun_out/${R}_st
head -4 gpurun_out/${R}_cfg3_1024_kernel_stats.csv
