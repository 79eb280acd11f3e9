#!/bin/bash
# tracked products in the one-chain kernel: parity, then A/B at 256 / 128 / 512 chains
export TMPDIR=/tmp
R=${1:-r02o}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rmhmc.py -m gpu -q -x -k "tracked or fused or mfma4" > gpurun_out/${R}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${R}_tests.log
for C in ${CHAINS:-256 128 512}; do for pr in 1 0; do
  HTA_TUNING=rmhmc_pair=$pr timeout 200 python bench.py --workload cfg3 --chains $C --traj 100 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-api > gpurun_out/${R}_c${C}_p${pr}.json 2>> gpurun_out/${R}_err.log
  python - <<P
import json
j=json.load(open("gpurun_out/${R}_c${C}_p${pr}.json")); r=j["roofline"]
print("chains=${C} pair=${pr}: %.3e steps/s, %.2f ms/step, kernel %.2f ms/step" % (j["value"], j["ms_per_step"], r["kernel_ms_per_step"]))
P
done; done
