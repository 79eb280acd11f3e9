#!/bin/bash
# the whole -m gpu suite (xdist is not used: the tests share one GPU and process-global route keys), then smoke()
export TMPDIR=/tmp
R=${1:-suite}
mkdir -p gpurun_out
timeout ${2:-3000} python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/${R}_gpu_tests.txt 2>&1
tail -25 gpurun_out/${R}_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.txt 2>&1; tail -2 gpurun_out/${R}_smoke.txt
