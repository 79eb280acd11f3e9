#!/bin/bash
# ablation timing of rmhmc_uv_kernel (tools/uv_ablate.sh builds): what a phase's time is made of
export TMPDIR=/tmp
R=${1:-r04b}; shift
mkdir -p gpurun_out
: > gpurun_out/${R}_ablate.txt
for m in "$@"; do
  echo "== HTA_UV_ABLATE=$m" >> gpurun_out/${R}_ablate.txt
  HTA_LIB=tools/scratch/_abl/libhta_uv$m.so AB_REPS=2 timeout 300 python tools/ab_rmhmc.py 256:- 512:- 1024:rmhmc_uv_co=1 2>&1 | grep chains >> gpurun_out/${R}_ablate.txt
done
cat gpurun_out/${R}_ablate.txt
