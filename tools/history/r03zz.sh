#!/bin/bash
# round 3, final pass: the whole -m gpu suite, the driver's bench command, the rocprofv3 / PMC passes behind profiles/physical.json
cd /root/repo; mkdir -p gpurun_out/r03zz
t0=$(date +%s)
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r03zz/gpu_tests.txt 2>&1; tail -3 gpurun_out/r03zz/gpu_tests.txt
echo "tests $(( $(date +%s) - t0 )) s"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03zz/bench_stdout.txt 2> gpurun_out/r03zz/bench_stderr.txt
tail -1 gpurun_out/r03zz/bench_stdout.txt > gpurun_out/r03zz/bench_line.json; wc -c gpurun_out/r03zz/bench_line.json
cp gpurun_out/bench_detail.json gpurun_out/r03zz/bench_detail.json
echo "bench $(( $(date +%s) - t0 )) s"
python __graft_entry__.py smoke > gpurun_out/r03zz/smoke.txt 2>&1; tail -2 gpurun_out/r03zz/smoke.txt
bash tools/physical.sh r03zz > gpurun_out/r03zz/physical.log 2>&1; tail -2 gpurun_out/r03zz/physical.log | cut -c1-300
echo "all $(( $(date +%s) - t0 )) s"
