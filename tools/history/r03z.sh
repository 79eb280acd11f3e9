#!/bin/bash
# the lean RMHMC instances as the default: the tests that assert their route names, then the driver's bench command without
# the CPU baselines (the secondary entries on the last library)
export TMPDIR=/tmp
R=${1:-r03z2}
mkdir -p gpurun_out
timeout 60 python -m pytest tests/test_gpu_rmhmc.py -m gpu -q -k "lean_instances or cfg3_reference_fixture" > gpurun_out/${R}_tests_a.txt 2>&1; echo "a rc=$?" > gpurun_out/${R}_rc.txt
timeout 60 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "cfg3_bench and 512" > gpurun_out/${R}_tests_b.txt 2>&1; echo "b rc=$?" >> gpurun_out/${R}_rc.txt
timeout 60 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${R}_bench_stdout.txt 2> gpurun_out/${R}_bench.err; echo "bench rc=$?" >> gpurun_out/${R}_rc.txt
tail -1 gpurun_out/${R}_bench_stdout.txt > gpurun_out/${R}_bench_line_no_cpu.json
