#!/bin/bash
export TMPDIR=/tmp
R=${1:-r04c}; shift
mkdir -p gpurun_out
: > gpurun_out/${R}_ablate.txt
for m in "$@"; do
  for L in 10 20; do
  echo "== HTA_UV_ABLATE=$m L=$L" >> gpurun_out/${R}_ablate.txt
  AB_L=$L HTA_LIB=tools/scratch/_abl/libhta_uv$m.so AB_REPS=2 timeout 300 python tools/ab_rmhmc.py 256:- 1024:rmhmc_uv_co=1 2>&1 | grep chains >> gpurun_out/${R}_ablate.txt
  done
done
cat gpurun_out/${R}_ablate.txt
