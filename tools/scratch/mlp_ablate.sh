#!/bin/bash
# usage: tools/scratch/mlp_ablate.sh build   (here)   |   tools/scratch/mlp_ablate.sh run   (on the GPU box)
cd "$(dirname "$0")/../.."
OUT=tools/scratch/_abl
if [ "$1" = build ]; then
  mkdir -p $OUT
  for abl in ${ABLS:-0 1 2 4 8 16 5 31}; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -fno-slp-vectorize -DHTA_MLP_SINGLE -DHTA_ABL=$abl $EXTRA \
      -Iinclude -Ihamiltorch_amd/csrc -x hip hamiltorch_amd/csrc/mlp_hmc.hip hamiltorch_amd/csrc/mlp_mfma.hip hamiltorch_amd/csrc/abi.cpp -x hip tools/scratch/mlp_ablate.cpp -o $OUT/abl_${abl}$TAG &
  done
  wait
else
  for f in $(ls $OUT/abl_* | sort); do timeout 120 $f 512 $(basename $f); done
fi
