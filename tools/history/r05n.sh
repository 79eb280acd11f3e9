#!/bin/bash
# round 5, call n: does the warm-up length move the RMHMC secondaries (clock ramp)?  same workload, three brackets, twice
for rep in 1 2; do
for sw in "10 2" "20 5" "40 10"; do set -- $sw
  python bench.py --workload cfg3@1024 --steps $1 --warmup $2 --no-cpu-baseline --no-api 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('cfg3@1024 steps $1 warmup $2: value %.4g ms_per_step %.3f kernel_ms %.3f' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms']))"
done; done
for sw in "10 2" "40 10"; do set -- $sw
  python bench.py --workload cfg3 --steps $1 --warmup $2 --no-cpu-baseline --no-api 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('cfg3@256 steps $1 warmup $2: value %.4g ms_per_step %.3f' % (j['value'], j['ms_per_step']))"
  python bench.py --workload cfg4 --steps $1 --warmup $2 --no-cpu-baseline --no-api 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('cfg4 steps $1 warmup $2: value %.4g ms_per_step %.3f' % (j['value'], j['ms_per_step']))"
done
