#!/bin/bash
# One bounded GPU-box pass at the end of a session: the newest kernel's tests first, then smoke, the bench lines, and as
# much of the whole -m gpu suite as the time allows (overall deadline: $2 seconds, every step under `timeout`).
# Everything lands under gpurun_out/<tag>_*.
export TMPDIR=/tmp
R=${1:-r01e}
DEADLINE=${2:-400}
mkdir -p gpurun_out
t0=$(date +%s)
el() { echo $(( $(date +%s) - t0 )); }
stamp() { echo "[$(el) s] $*" >> gpurun_out/${R}_timeline.txt; }
left() { local l=$(( DEADLINE - $(el) )); [ $l -lt 1 ] && l=1; [ $l -gt $1 ] && l=$1; echo $l; }
stamp start
timeout $(left 200) python -m pytest tests/test_gpu_rmhmc.py -x -q -k "batched_mfma or momentum_overlap" > gpurun_out/${R}_tests_batch.log 2>&1; stamp "batch tests rc=$?"
timeout $(left 120) python __graft_entry__.py smoke > gpurun_out/${R}_smoke.log 2>&1; stamp "smoke rc=$?"
for C in 4096 2048; do
  timeout $(left 120) python bench.py --workload cfg3 --chains $C --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_cfg3_${C}.json 2> gpurun_out/${R}_cfg3_${C}.err; stamp "cfg3 $C rc=$?"
done
timeout $(left 200) python bench.py > gpurun_out/${R}_cfg2_bench.json 2> gpurun_out/${R}_cfg2_bench.err; stamp "bench default rc=$?"
timeout $(left 900) python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/${R}_tests_all.log 2>&1; stamp "all gpu tests rc=$?"
