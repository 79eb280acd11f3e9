#!/bin/bash
# first GPU pass of round 2: whole -m gpu suite, smoke, the default bench line (with secondaries), the physical passes
export TMPDIR=/tmp
R=${1:-r02a}
mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/${R}_tests.log 2>&1; echo "tests rc=$? ($(( $(date +%s) - t0 )) s)"; tail -30 gpurun_out/${R}_tests.log
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err; echo "bench rc=$? ($(( $(date +%s) - t0 )) s)"; head -c 3000 gpurun_out/${R}_bench.json; tail -5 gpurun_out/${R}_bench.err
bash tools/physical.sh $R
