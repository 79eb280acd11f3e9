#!/bin/bash
# four-wave mfma4 kernel: parity, then A/B at 1024 / 2048 chains
export TMPDIR=/tmp
R=${1:-r02e}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rmhmc.py tests/test_gpu_fullsize.py -m gpu -q -x -k "mfma4 or cfg5_shapes or cfg3_bench_instances" > gpurun_out/${R}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${R}_tests.log
for C in 1024 768 2048; do for wv in 4 2; do
  HTA_TUNING=rmhmc_mfma4_waves=$wv timeout 200 python bench.py --workload cfg3 --chains $C --traj 100 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_c${C}_w${wv}.json 2>> gpurun_out/${R}_err.log
  python - <<P
import json
j=json.load(open("gpurun_out/${R}_c${C}_w${wv}.json")); r=j["roofline"]
print("chains=${C} waves=${wv}: %.3e steps/s, %.2f ms/step, kernel %.2f ms/step" % (j["value"], j["ms_per_step"], r["kernel_ms_per_step"]))
P
done; done
