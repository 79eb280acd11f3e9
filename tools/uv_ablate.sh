#!/bin/bash
# builds the ablation variants of rmhmc_uv.hip (csrc/rmhmc_uv.hip: HTA_UV_ABLATE) into tools/scratch/_abl/libhta_uv<mask>.so -
# run on the build host, the libraries travel with the snapshot; then on the GPU box:
#   for m in 0 1 2 4 8 16 ...; do HTA_LIB=tools/scratch/_abl/libhta_uv$m.so python tools/ab_rmhmc.py 256:- 1024:rmhmc_uv_co=1; done
set -e
cd "$(dirname "$0")/../hamiltorch_amd/csrc"
OUT=../../tools/scratch/_abl
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -fno-slp-vectorize"
OBJS=$(ls build/*.o | grep -v rmhmc_uv.o)
for m in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DHTA_UV_ABLATE=$m -x hip -c rmhmc_uv.hip -o $OUT/rmhmc_uv_$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libhta_uv$m.so $OBJS $OUT/rmhmc_uv_$m.o
  rm -f $OUT/rmhmc_uv_$m.o
done
