#!/bin/bash
# round 3, step h: split momentum draw + serial default.  tests of the RMHMC routes, then the three explicit-RMHMC bench lines
cd /root/repo; mkdir -p gpurun_out/r03h
timeout 1500 python -m pytest tests/test_gpu_rmhmc.py tests/test_gpu_fullsize.py tests/test_gpu_routes.py -x -q -m gpu > gpurun_out/r03h/tests.txt 2>&1
tail -5 gpurun_out/r03h/tests.txt
for w in cfg3@1024 cfg3 cfg5; do
  timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-secondary 2>&1 | tail -1 > gpurun_out/r03h/bench_$w.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/r03h/bench_$w.json").read())
print("$w", d["value"], d["ms_per_step"], d.get("config",{}).get("kernel"), d.get("roofline",{}).get("frac"))
PY
done
for ov in 0 1; do for sp in 0 1; do
  HTA_TUNING="rmhmc_overlap=$ov,rmhmc_momsplit=$sp" timeout 300 python bench.py --workload cfg3@1024 --steps 10 --warmup 3 --no-secondary 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap $ov split $sp', d['value'])"
done; done
