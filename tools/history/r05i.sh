#!/bin/bash
# round 5, call i: what the per-trajectory reset of the warm eigenbases costs on the generic explicit-RMHMC path
for v in 1 0 1 0; do
  HAMILTORCH_AMD_WARM_RESET=$v timeout 300 python bench.py --workload funnel-rmhmc --steps 3 --warmup 1 --no-cpu-baseline --no-api 2> /dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('warm_reset $v funnel-rmhmc value %.4g ms_per_step %.2f acc %.3f' % (j['value'], j['ms_per_step'], j.get('acceptance_rate') or -1))"
done
