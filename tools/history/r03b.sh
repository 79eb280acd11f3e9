#!/bin/bash
# round 3, second GPU run: parity of csrc/mlp3_mfma.hip + first timings and counters of the notebook model.
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_mlp3.py -x -q 2>&1 | tail -30) > $O/mlp3_tests.txt
for wl in "nbmlp" "nbmlp --chains 256" "nbmlp --chains 512" "nbmlp-full" "nbmlp-full --chains 256"; do
  timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2> /dev/null | tail -1 >> $O/nbmlp_bench_lines.txt
done
cmd="python bench.py --workload nbmlp --steps 2 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/stats -o s -- $cmd > /dev/null 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/nbmlp_kernel_stats.csv 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $O/pmc_sq -o q -- $cmd > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU -f csv -d $O/pmc_lds -o l -- $cmd > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_rd -o r -- $cmd > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_wr -o w -- $cmd > /dev/null 2>&1
python tools/pmc_summarize.py $(find $O -name "*counter_collection.csv" | sort) > $O/nbmlp_pmc.txt 2>&1
find $O -name "*.csv" -size +2M -delete
tail -5 $O/mlp3_tests.txt; cat $O/nbmlp_bench_lines.txt | cut -c 1-400
