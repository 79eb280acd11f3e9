#!/bin/bash
# builds the cfg4 phase-timing harness: tools/scratch/_abl/mlp_plain (the library as built) and mlp_t (mlp_mfma.hip with
# -DHTA_TIMING=1: s_memtime stamps of wave 0 at the phase boundaries of a gradient)
set -e
cd "$(dirname "$0")/../.."
OUT=tools/scratch/_abl; mkdir -p $OUT
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FL -DHTA_TIMING=1 $EXTRA -Iinclude -x hip -c hamiltorch_amd/csrc/mlp_mfma.hip -o $OUT/mlp_mfma_t.o
OBJS=$(ls hamiltorch_amd/csrc/build/*.o | grep -v mlp_mfma.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libhta_mlp_t.so $OBJS $OUT/mlp_mfma_t.o
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -Iinclude -DHTA_TIMING=1 -x hip tools/scratch/mlp_ablate.cpp -o $OUT/mlp_t -L$OUT -lhta_mlp_t -Wl,-rpath,'$ORIGIN'
cp hamiltorch_amd/libhamiltorch_amd.so $OUT/libhta_plain.so
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -Iinclude -x hip tools/scratch/mlp_ablate.cpp -o $OUT/mlp_plain -L$OUT -lhta_plain -Wl,-rpath,'$ORIGIN'
