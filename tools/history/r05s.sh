#!/bin/bash
# round 5, call s: the two-chains-per-workgroup MLP kernel - bit identity, the MLP test files, cfg4 A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mlp.py -q -x --tb=short -k "pair" > gpurun_out/r05s_pair.txt 2>&1; tail -15 gpurun_out/r05s_pair.txt | cut -c1-250
for v in 0 1 0 1; do
  HTA_TUNING=mlp_pair=$v timeout 300 python bench.py --workload cfg4 --steps 20 --warmup 5 --no-cpu-baseline --no-api 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('mlp_pair $v cfg4 value %.4g ms_per_step %.3f kernel_ms %.3f acc %.3f route %s' % (j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j.get('acceptance_rate') or -1, j.get('roofline',{}).get('kernel')))"
done
for C in 1024 2048; do for v in 0 1; do
  HTA_TUNING=mlp_pair=$v timeout 300 python bench.py --workload cfg4 --chains $C --steps 10 --warmup 3 --no-cpu-baseline --no-api 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('mlp_pair $v cfg4 C=$C value %.4g ms_per_step %.3f' % (j['value'], j['ms_per_step']))"
done; done
