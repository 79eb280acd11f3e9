#!/bin/bash
export TMPDIR=/tmp
R=${1:-r04d}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rmhmc.py -q -x -k "uvc_kernel" > gpurun_out/${R}_tests.txt 2>&1
tail -15 gpurun_out/${R}_tests.txt
timeout 600 python tools/ab_rmhmc.py 256:- 256:rmhmc_uvc=1 256:rmhmc_uvc=1,rmhmc_uv_co=1 128:- 128:rmhmc_uvc=1 \
   512:- 512:rmhmc_uvc=1,rmhmc_uv_co=1,rmhmc_uv_g=1 1024:rmhmc_uv_co=1 > gpurun_out/${R}_ab.txt 2>&1
cat gpurun_out/${R}_ab.txt
