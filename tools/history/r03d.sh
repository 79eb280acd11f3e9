#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r03d; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $O/gpu_tests.txt
for wl in "cfg4" "cfg3 --traj 20"; do
  HTA_RMHMC_FUSED=0 timeout 300 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2> /dev/null | tail -1 | cut -c 1-1200 >> $O/lines.txt
done
tail -3 $O/gpu_tests.txt; python - <<'PY'
import json
for ln in open('gpurun_out/r03d/lines.txt'):
    j=json.loads(ln); r=j['roofline']; print(j['config']['workload'], j['config']['chains_per_gpu'], j['value'], j['ms_per_step'], r['frac'], r['kernel'], r['kernel_ms'])
PY
