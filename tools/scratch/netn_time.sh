#!/bin/bash
# builds the timing variant of the library (netn_hmc.hip with -DNETN_TIMING=1, the other objects as built)
set -e
cd "$(dirname "$0")/../../hamiltorch_amd/csrc"
mkdir -p ../../tools/scratch/_abl
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -fno-slp-vectorize -DNETN_TIMING=1 -x hip -c netn_hmc.hip -o ../../tools/scratch/_abl/netn_hmc_t.o
OBJS=$(ls build/*.o | grep -v netn_hmc.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/scratch/_abl/libhta_netn_timing.so $OBJS ../../tools/scratch/_abl/netn_hmc_t.o
