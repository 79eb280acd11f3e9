#!/bin/bash
export TMPDIR=/tmp
for t in "quad_fused=0" "quad_fused=1,quad_producers=128,quad_chunks=4" "quad_fused=1,quad_producers=128,quad_chunks=6" "quad_fused=1,quad_producers=128,quad_chunks=8" "quad_fused=1,quad_producers=128,quad_chunks=12" "quad_fused=1,quad_producers=64,quad_chunks=8" "quad_fused=1,quad_producers=96,quad_chunks=8" "quad_fused=1,quad_producers=192,quad_chunks=8" "quad_fused=1,quad_producers=240,quad_chunks=8"; do
HTA_TUNING=$t timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-api --steps 50 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$t value %.4g ms_per_step %.5f kernel_ms %s' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms']))"
done
