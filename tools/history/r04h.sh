#!/bin/bash
export TMPDIR=/tmp
R=${1:-r04h}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rmhmc.py tests/test_gpu_hmc.py -q -k "float64 or dtype1 or fp64" --durations=8 > gpurun_out/${R}_fp64.txt 2>&1
tail -25 gpurun_out/${R}_fp64.txt
