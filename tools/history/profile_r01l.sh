#!/bin/bash
# Final pass of the round: whole -m gpu suite, smoke, the bench lines (cfg2 default; cfg3 at 256 / 1024 / 2048 / 4096
# chains; cfg4), rocprofv3 kernel stats for the cfg3 routes and the matrix-pipe counters of the 1024- and 4096-chain routes.
export TMPDIR=/tmp
R=${1:-r01l}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 200 python -m pytest tests -m gpu -x -q > gpurun_out/${R}_tests_all.log 2>&1; echo "tests rc=$? ($(( $(date +%s) - t0 )) s)"; tail -1 gpurun_out/${R}_tests_all.log
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > gpurun_out/${R}_cfg2_bench.json 2> /dev/null
python bench.py --workload cfg3 --steps 3 --warmup 1 > gpurun_out/${R}_cfg3_bench.json 2> /dev/null
for C in 1024 2048 4096; do python bench.py --workload cfg3 --chains $C --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_cfg3_${C}_bench.json 2> /dev/null; done
echo "benches done ($(( $(date +%s) - t0 )) s)"
stats() { local n=$1; shift
  timeout 100 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/${R}_${n}_stats -o $n -- python bench.py --no-cpu-baseline "$@" > /dev/null 2>&1
  cp $(find gpurun_out/${R}_${n}_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_${n}_kernel_stats.csv; rm -rf gpurun_out/${R}_${n}_stats; }
stats cfg3 --workload cfg3 --steps 2 --warmup 1
stats cfg3_1024 --workload cfg3 --chains 1024 --steps 2 --warmup 1
stats cfg3_4096 --workload cfg3 --chains 4096 --steps 2 --warmup 1
echo "stats done ($(( $(date +%s) - t0 )) s)"
for C in 1024 4096; do
  timeout 100 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d gpurun_out/${R}_pmc_${C} -o c$C -- python bench.py --workload cfg3 --chains $C --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
done
python tools/pmc_summarize.py $(find gpurun_out -name "*counter_collection.csv" | sort) > gpurun_out/${R}_cfg3_pmc_mfma.txt
find gpurun_out -name "${R}_pmc_*" -type d -exec rm -rf {} +
echo "pmc done ($(( $(date +%s) - t0 )) s)"; cat gpurun_out/${R}_cfg3_pmc_mfma.txt | head -30
