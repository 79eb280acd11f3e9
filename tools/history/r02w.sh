#!/bin/bash
# which route between 513 and 703 chains: one-chain kernel (default), four-chain kernel (rmhmc_mfma4=2), uv kernel in two rounds (rmhmc_uv=2)
export TMPDIR=/tmp
R=${1:-r02w}
mkdir -p gpurun_out
for C in 544 576 640 700; do for t in "default rmhmc_uv=1" "fourchain rmhmc_mfma4=2" "uv rmhmc_uv=2"; do set -- $t
  HTA_TUNING=$2 timeout 200 python bench.py --workload cfg3 --chains $C --traj 100 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-api > gpurun_out/${R}_c${C}_$1.json 2>> gpurun_out/${R}_err.log
  python - <<P
import json
j=json.load(open("gpurun_out/${R}_c${C}_$1.json")); r=j["roofline"]
print("chains=${C} $1: %.3e steps/s, kernel %.2f ms/step" % (j["value"], r["kernel_ms_per_step"]))
P
done; done
