#!/bin/bash
# rocprofv3 kernel stats of the current library for cfg2 (default), cfg3 at 256 / 1024 / 4096 chains and cfg4, plus the
# HBM counters of cfg3 (separate --pmc passes).  Run on the GPU box via gpurun; summaries under gpurun_out/<tag>_*.
export TMPDIR=/tmp
R=${1:-r01e}
mkdir -p gpurun_out
stats() {   # name, bench args...
  local n=$1; shift
  timeout 150 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/${R}_${n}_stats -o $n -- python bench.py --no-cpu-baseline "$@" > gpurun_out/${R}_${n}_stats_stdout.txt 2>&1
  cp $(find gpurun_out/${R}_${n}_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_${n}_kernel_stats.csv
  rm -rf gpurun_out/${R}_${n}_stats
}
stats cfg2
stats cfg3 --workload cfg3 --steps 3 --warmup 1
stats cfg3_1024 --workload cfg3 --chains 1024 --steps 2 --warmup 1
stats cfg3_4096 --workload cfg3 --chains 4096 --steps 2 --warmup 1
stats cfg4 --workload cfg4 --steps 3 --warmup 1
python bench.py --workload cfg3 --steps 3 --warmup 1 > gpurun_out/${R}_cfg3_bench.json 2> /dev/null
python bench.py --workload cfg3 --chains 1024 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_cfg3_1024_bench.json 2> /dev/null
python bench.py --workload cfg4 --steps 3 --warmup 1 > gpurun_out/${R}_cfg4_bench.json 2> /dev/null
for CTR in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $CTR -f csv -d gpurun_out/${R}_cfg3_pmc_${CTR} -o cfg3 -- python bench.py --workload cfg3 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${R}_cfg3_pmc_${CTR}_stdout.txt 2>&1
done
python tools/pmc_summarize.py $(find gpurun_out -name "*counter_collection.csv" | sort) > gpurun_out/${R}_cfg3_pmc_hbm.txt
find gpurun_out -name "${R}_cfg3_pmc_*" -type d -exec rm -rf {} +
ls gpurun_out
