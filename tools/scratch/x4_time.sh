#!/bin/bash
# builds the timing variant of the library (rmhmc_fused.hip with -DHTA_RM_TIMING=1, the other objects as built)
set -e
cd "$(dirname "$0")/../../hamiltorch_amd/csrc"
mkdir -p ../../tools/scratch/_abl
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -fno-slp-vectorize -DHTA_RM_TIMING=1 -x hip -c rmhmc_fused.hip -o ../../tools/scratch/_abl/rmhmc_fused_t.o
OBJS=$(ls build/*.o | grep -v rmhmc_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/_abl/libhta_timing.so $OBJS ../../tools/scratch/_abl/rmhmc_fused_t.o
