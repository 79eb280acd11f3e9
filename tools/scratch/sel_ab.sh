export TMPDIR=/tmp
for sel in 1 0; do for ch in 256 1024; do
  r=$(HTA_TUNING_DEFAULTS=metric_select=$sel HTA_RMHMC_FUSED=0 timeout 600 python bench.py --workload cfg3 --chains $ch --traj 5 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-api 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])")
  echo "select=$sel chains=$ch: $r"
done; done
