#!/bin/bash
# builds the timing variant of the library (mlp3_mfma.hip with -DM3_TIMING=1, the other objects as built)
set -e
cd "$(dirname "$0")/../../hamiltorch_amd/csrc"
mkdir -p ../../tools/scratch/_abl
OBJS=$(ls build/*.o | grep -v mlp3_mfma.o)
for v in 0 1; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -fno-slp-vectorize -DM3_TIMING=1 -DM3_PREFETCH_GEMM2=$v $M3_EXTRA -x hip -c mlp3_mfma.hip -o ../../tools/scratch/_abl/mlp3_mfma_t$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/_abl/libhta_m3_timing$v.so $OBJS ../../tools/scratch/_abl/mlp3_mfma_t$v.o
done
