#!/bin/bash
# A/B of the wave-per-task momentum kernel against the workgroup-per-task one (tests first); run on the GPU box via gpurun.
export TMPDIR=/tmp
R=${1:-r01f}
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_rmhmc.py -x -q > gpurun_out/${R}_tests_rmhmc.log 2>&1; echo "tests rc=$?" > gpurun_out/${R}_ab.txt
for C in 256 1024 4096; do
  for M in 1 0; do
    HTA_TUNING=rmhmc_momwave=$M timeout 120 python bench.py --workload cfg3 --chains $C --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_cfg3_${C}_mw${M}.json 2>> gpurun_out/${R}_ab.err
    python - gpurun_out/${R}_cfg3_${C}_mw${M}.json $C $M >> gpurun_out/${R}_ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("chains %s momwave %s: %.3e steps/s, %.2f ms/step, acc %.4f" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["acceptance_rate"]))
PY
  done
done
timeout 150 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/${R}_st -o x -- python bench.py --workload cfg3 --chains 4096 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cp $(find gpurun_out/${R}_st -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_cfg3_4096_kernel_stats.csv; rm -rf gpurun_out/${R}_st
cat gpurun_out/${R}_ab.txt
