#!/bin/bash
# rocprofv3 kernel stats for the cfg3 / cfg4 workloads (run on the GPU box via gpurun)
export TMPDIR=/tmp
R=${1:-r01}
mkdir -p gpurun_out
for W in cfg3 cfg4; do
  python bench.py --workload $W --steps 3 --warmup 1 > gpurun_out/${R}_${W}_bench.json 2> /dev/null
  rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/${R}_${W}_stats -o $W -- python bench.py --workload $W --steps 3 --warmup 1 > gpurun_out/${R}_${W}_stats_stdout.txt 2>&1
done
ls gpurun_out
