#!/bin/bash
export TMPDIR=/tmp
R=${1:-r04e}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rmhmc.py -q -x -k "uvc" > gpurun_out/${R}_tests.txt 2>&1
tail -15 gpurun_out/${R}_tests.txt
timeout 600 python tools/ab_rmhmc.py 512:- 512:rmhmc_uvc=1 512:rmhmc_uvc=1,rmhmc_uv_co=1 512:rmhmc_uvc=1,rmhmc_uv_co=1,rmhmc_uv_g=1 \
   384:- 384:rmhmc_uvc=1,rmhmc_uv_co=1 384:rmhmc_uvc=1,rmhmc_uv_co=1,rmhmc_uv_g=1 \
   768:- 768:rmhmc_uvc=1,rmhmc_uv_co=1 1024:- 1024:rmhmc_uv_co=1 1024:rmhmc_uvc=1,rmhmc_uv_co=1 \
   1536:- 1536:rmhmc_uvc=1,rmhmc_uv_co=1,rmhmc_uv=2 2048:- 2048:rmhmc_uvc=1,rmhmc_uv_co=1,rmhmc_uv=2 4096:- 4096:rmhmc_uvc=1,rmhmc_uv_co=1,rmhmc_uv=2 > gpurun_out/${R}_ab.txt 2>&1
cat gpurun_out/${R}_ab.txt
