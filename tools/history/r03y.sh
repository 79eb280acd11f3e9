#!/bin/bash
# A/B of the lean instances of the lone-wave RMHMC kernels (tuning key rmhmc_lean): bit-identity test, then bench lines.
export TMPDIR=/tmp
R=${1:-r03y}
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_rmhmc.py -m gpu -q -k "lean_instances" > gpurun_out/${R}_lean_test.txt 2>&1
echo "test rc=$?" > gpurun_out/${R}_rc.txt
for w in cfg3 cfg3@1024; do
  for v in 0 1 0 1; do
    echo "workload=$w rmhmc_lean=$v" >> gpurun_out/${R}_ab_lines.txt
    HTA_TUNING=rmhmc_lean=$v timeout 60 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-api 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({k: d[k] for k in ('value','ms_per_step')} | {'kernel': d['roofline']['kernel'], 'kernel_ms': d['roofline'].get('kernel_ms')}))" >> gpurun_out/${R}_ab_lines.txt
  done
done
