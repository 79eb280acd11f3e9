#!/bin/bash
# Round profile of the headline workload (run on the GPU box via gpurun). Outputs under gpurun_out/.
set -x
export TMPDIR=/tmp
R=${1:-r01}
mkdir -p gpurun_out
python bench.py --sweep > gpurun_out/${R}_cfg2_bench.json 2> gpurun_out/${R}_cfg2_sweep.txt
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/${R}_cfg2_stats -o cfg2 -- python bench.py --no-cpu-baseline > gpurun_out/${R}_cfg2_stats_stdout.txt 2>&1
# PMC passes on their own (no trace domains besides kernel-trace), one counter group per run
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d gpurun_out/${R}_cfg2_pmc_fetch -o cfg2 -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${R}_cfg2_pmc_fetch_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d gpurun_out/${R}_cfg2_pmc_write -o cfg2 -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${R}_cfg2_pmc_write_stdout.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d gpurun_out/${R}_cfg2_pmc_sq -o cfg2 -- python bench.py --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/${R}_cfg2_pmc_sq_stdout.txt 2>&1
ls -R gpurun_out | head -60
