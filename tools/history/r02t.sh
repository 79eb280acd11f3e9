#!/bin/bash
# uv kernel under the 256-register cap (two workgroups per CU): 256 / 512 chains, and forced at 1024 / 768 against the four-chain kernel
export TMPDIR=/tmp
R=${1:-r02t}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rmhmc.py -m gpu -q -x -k "uv_kernel" > gpurun_out/${R}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${R}_tests.log
for cfg in "256 1" "512 1" "1024 2" "1024 1" "768 2" "768 1" "2048 2" "2048 1"; do set -- $cfg; C=$1; uv=$2
  HTA_TUNING=rmhmc_uv=$uv timeout 200 python bench.py --workload cfg3 --chains $C --traj 100 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-api > gpurun_out/${R}_c${C}_u${uv}.json 2>> gpurun_out/${R}_err.log
  python - <<P
import json
j=json.load(open("gpurun_out/${R}_c${C}_u${uv}.json")); r=j["roofline"]
print("chains=${C} uv=${uv}: %.3e steps/s, %.2f ms/step, kernel %.2f ms/step" % (j["value"], j["ms_per_step"], r["kernel_ms_per_step"]))
P
done
