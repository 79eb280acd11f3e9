#!/bin/bash
# Round 3, final GPU pass on the round's last library (quad_variant 7 default, prepared eig block, hta_run_begin):
# the whole -m gpu suite, smoke(), the driver's bench command, and the cfg2 counter passes behind profiles/physical.json.
export TMPDIR=/tmp
R=${1:-r03w}
mkdir -p gpurun_out
t0=$(date +%s)
el() { echo $(( $(date +%s) - t0 )); }
stamp() { echo "[$(el) s] $*" >> gpurun_out/${R}_timeline.txt; }
stamp start
timeout 400 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/${R}_gpu_tests.txt 2>&1
stamp "suite rc=$?"
timeout 60 python __graft_entry__.py smoke > gpurun_out/${R}_smoke.txt 2>&1
stamp "smoke rc=$?"
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_stdout.txt 2> gpurun_out/${R}_bench.err
stamp "driver bench rc=$?"
tail -1 gpurun_out/${R}_bench_stdout.txt > gpurun_out/${R}_bench_line.json
cp bench_detail.json gpurun_out/${R}_bench_detail.json 2>/dev/null
# cfg2 counter passes (the block `one cfg2@1024 ...` of tools/physical.sh)
O=gpurun_out/${R}_phys/cfg2@1024; mkdir -p $O
cmd="python bench.py --no-cpu-baseline --no-secondary --no-api --steps 20 --warmup 3"
echo "{\"command\": \"$cmd\", \"steps\": 20, \"warmup\": 3, \"traj\": 1000, \"source\": \"tools/r03w.sh $R\"}" > $O/meta.json
timeout 100 rocprofv3 --kernel-trace --stats -f csv -d $O/stats -o s -- $cmd > $O/bench.json 2> /dev/null; stamp "stats rc=$?"
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_rd -o r -- $cmd > /dev/null 2>&1; stamp "pmc rd rc=$?"
timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_wr -o w -- $cmd > /dev/null 2>&1; stamp "pmc wr rc=$?"
timeout 100 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $O/pmc_sq -o q -- $cmd > /dev/null 2>&1; stamp "pmc sq rc=$?"
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) "gpurun_out/${R}_cfg2@1024_kernel_stats.csv" 2> /dev/null
python tools/physical.py cfg2@1024=$O > gpurun_out/${R}_physical_cfg2.json 2> gpurun_out/${R}_physical.err
python tools/pmc_summarize.py $(find $O -name "*counter_collection.csv" | sort) > gpurun_out/${R}_pmc_cfg2.txt 2>/dev/null
rm -rf gpurun_out/${R}_phys
stamp end
