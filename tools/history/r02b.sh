#!/bin/bash
# metric-evaluation kernel on the matrix cores: parity tests, then A/B of the eigendecomposition route of cfg3
export TMPDIR=/tmp
R=${1:-r02b}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_rmhmc.py -m gpu -q -x -k "mfma_kernel or fused_path_equals or metric_eval or reference_fixtures or sample_rmhmc or hessian" > gpurun_out/${R}_tests.log 2>&1; echo "tests rc=$? ($(( $(date +%s) - t0 )) s)"; tail -25 gpurun_out/${R}_tests.log
for mode in 1 0; do
  HTA_RMHMC_FUSED=0 HTA_TUNING=metric_mfma=$mode timeout 200 python bench.py --workload cfg3 --traj 20 --steps 2 --warmup 1 --no-cpu-baseline --no-api > gpurun_out/${R}_jacobi_mfma${mode}.json 2> gpurun_out/${R}_err.log
  python - <<P
import json
j=json.load(open("gpurun_out/${R}_jacobi_mfma${mode}.json")); r=j["roofline"]
print("metric_mfma=${mode}: %.3e steps/s, %.1f ms/step, kernel %.1f ms/step, %d launches/step, exec %.1f TF (%.3f)" % (j["value"], j["ms_per_step"], r["kernel_ms_per_step"], r["launches_per_step"], r["achieved"], r["frac"]))
P
done
echo "done ($(( $(date +%s) - t0 )) s)"
