#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_mlp3.py -x -q 2>&1 | tail -8) > $O/mlp3_tests.txt
(timeout 300 python tools/scratch/m3_time.py 2>&1 | grep -v amdgpu.ids) > $O/m3_time.txt
for wl in "nbmlp" "nbmlp-full"; do
  timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2> /dev/null | tail -1 | cut -c 1-900 >> $O/nbmlp_bench_lines.txt
done
tail -3 $O/mlp3_tests.txt; cat $O/m3_time.txt; cat $O/nbmlp_bench_lines.txt | cut -c 1-330
