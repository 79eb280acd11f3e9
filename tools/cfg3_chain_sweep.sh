#!/bin/bash
# explicit-RMHMC rate of the default route at every chain count (which trajectory kernel serves it: bench.py's roofline.kernel)
export TMPDIR=/tmp
R=${1:-sweep}
mkdir -p gpurun_out
for C in 64 128 256 384 512 640 768 1024 1536 2048 3072 4096 8192; do
  T=$(( C <= 512 ? 200 : (C <= 2048 ? 100 : 50) ))
  timeout 200 python bench.py --workload cfg3 --chains $C --traj $T --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-api > gpurun_out/${R}_c${C}.json 2>> gpurun_out/${R}_err.log
  python - <<P
import json
j=json.loads([l for l in open("gpurun_out/${R}_c${C}.json").read().splitlines() if l.strip()][-1]); r=j["roofline"]
print("chains=%5d  %.3e steps/s  kernel %.2f ms per %d-trajectory call  %s" % (${C}, j["value"], r["kernel_ms_per_step"], ${T}, r["kernel"][:60]))
P
done
