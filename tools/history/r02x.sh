#!/bin/bash
# tracked products in the 16-chain kernel: parity, then A/B at 4096 / 3072 chains (and the two-wave four-chain kernel there)
export TMPDIR=/tmp
R=${1:-r02x}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rmhmc.py tests/test_gpu_fullsize.py -m gpu -q -x -k "batched or cfg5_shapes or cfg3_bench_instances or wave_momentum" > gpurun_out/${R}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${R}_tests.log
for cfg in "4096 rmhmc_pair=1" "4096 rmhmc_pair=0" "3072 rmhmc_pair=1" "3072 rmhmc_pair=0" "3072 rmhmc_mfma4=2" "4096 rmhmc_mfma4=2" "2560 rmhmc_pair=1" "2560 rmhmc_mfma4=2"; do set -- $cfg
  HTA_TUNING=$2 timeout 200 python bench.py --workload cfg3 --chains $1 --traj 50 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-api > gpurun_out/${R}_c$1_$2.json 2>> gpurun_out/${R}_err.log
  python - <<P
import json
j=json.load(open("gpurun_out/${R}_c$1_$2.json")); r=j["roofline"]
print("chains=$1 $2: %.3e steps/s, kernel %.2f ms/step" % (j["value"], r["kernel_ms_per_step"]))
P
done
