#!/bin/bash
# the momentum-row prefetch of rmhmc_uv_kernel / rmhmc_mfma4x4_kernel: cross-kernel parity tests, then the two bench lines
export TMPDIR=/tmp
R=${1:-r03p}
mkdir -p gpurun_out
timeout 48 python -m pytest tests/test_gpu_rmhmc.py -m gpu -q -x -k "uv_kernel_equals_one_chain or mfma4_kernel_equals_fused or lean_instances" > gpurun_out/${R}_tests.txt 2>&1; echo "tests rc=$?" > gpurun_out/${R}_rc.txt
for w in cfg3@1024 cfg3; do
  timeout 14 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-api 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(json.dumps({'workload': d['config']['workload'], 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'kernel': d['roofline']['kernel']}))" >> gpurun_out/${R}_lines.txt
done
