#!/bin/bash
# round 4, first GPU pass: the co-resident / four-chain instances of rmhmc_uv_kernel - parity test, then A/B lines
export TMPDIR=/tmp
R=${1:-r04a}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rmhmc.py -q -x -k "coresident_and_four_chain or lean_instances" > gpurun_out/${R}_tests.txt 2>&1
tail -3 gpurun_out/${R}_tests.txt
timeout 900 python tools/ab_rmhmc.py 256:- 256:rmhmc_uv_acc=4 256:rmhmc_uv_co=1 256:rmhmc_uv_co=1,rmhmc_uv_acc=4 \
  512:- 512:rmhmc_uv_acc=4 512:rmhmc_uv_co=1 512:rmhmc_uv_co=1,rmhmc_uv_acc=4 512:rmhmc_uv_co=1,rmhmc_uv_g=1 512:rmhmc_uv_co=1,rmhmc_uv_g=1,rmhmc_uv_acc=4 \
  768:- 768:rmhmc_uv_co=1,rmhmc_uv_acc=4 \
  1024:- 1024:rmhmc_uv=2 1024:rmhmc_uv_co=1 1024:rmhmc_uv_co=1,rmhmc_uv_acc=4 \
  1536:- 1536:rmhmc_uv_co=1,rmhmc_uv_acc=4,rmhmc_uv=2 \
  2048:- 2048:rmhmc_uv_co=1,rmhmc_uv_acc=4,rmhmc_uv=2 4096:- 4096:rmhmc_uv_co=1,rmhmc_uv_acc=4,rmhmc_uv=2 > gpurun_out/${R}_ab.txt 2>&1
cat gpurun_out/${R}_ab.txt
