#!/bin/bash
export TMPDIR=/tmp
R=${1:-r04j}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_hmc.py -q -x -k "fused_quad" > gpurun_out/${R}_tests.txt 2>&1; tail -5 gpurun_out/${R}_tests.txt
for f in 0 1 0 1; do
HTA_TUNING=quad_fused=$f timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-api --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('quad_fused=$f value %.4g ms_per_step %.5f kernel_ms %s kernel %s acc %.4f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['kernel'], j['acceptance_rate']))"
done
