#!/bin/bash
# round 5, call g: chain groups of the callback path (parity test, funnel rates at 1 / 2 / 4 / 8 groups), the cfg2 saturating sweep
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r05g
timeout 600 python -m pytest tests/test_gpu_routes.py -q -x --tb=short -k "chain_groups or generic or graph or callback" > ${O}_tests.txt 2>&1; tail -15 ${O}_tests.txt | cut -c1-250
for G in 1 2 4 8; do
  HAMILTORCH_AMD_GROUPS=$G timeout 300 python bench.py --workload funnel-hmc --steps 6 --warmup 2 --no-cpu-baseline --no-api 2> /dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('groups $G funnel-hmc value %.4g ms_per_step %.2f acc %.3f' % (j['value'], j['ms_per_step'], j.get('acceptance_rate') or -1))"
done
timeout 600 python bench.py --sweep --no-secondary --no-cpu-baseline --no-api --steps 20 --warmup 5 > ${O}_sweep_stdout.txt 2> ${O}_sweep.txt; grep sweep ${O}_sweep.txt; cp bench_detail.json ${O}_sweep_detail.json
