#!/bin/bash
# round 3, step n: which trajectory kernel wins at which chain count now that the momentum draws cost next to nothing
cd /root/repo; mkdir -p gpurun_out/r03n
run() { # chains tuning
  HTA_TUNING="$2" timeout 200 python bench.py --workload cfg3@1024 --chains $1 --traj 100 --no-cpu-baseline --no-api --no-secondary --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C=%5d %-40s %.4e steps/s  %8.3f ms  %s' % ($1, '$2' or 'default', d['value'], d['ms_per_step'], d['roofline']['kernel']))"
}
for C in 256 512 768 1024 1536 2048 3072 4096 8192; do
  run $C ""
  run $C "rmhmc_uv=2"
  run $C "rmhmc_uv=0,rmhmc_mfma4=2"
  run $C "rmhmc_uv=0,rmhmc_mfma4=2,rmhmc_mfma4_waves=2"
  run $C "rmhmc_uv=0,rmhmc_mfma4=0,rmhmc_batch=2"
done > gpurun_out/r03n/route_sweep.txt 2>&1
cat gpurun_out/r03n/route_sweep.txt
