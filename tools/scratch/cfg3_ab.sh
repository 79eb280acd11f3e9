export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rmhmc.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -2
for w in cfg3@1024 cfg3; do for k in 1 2; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-api 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w', j['value'], j['ms_per_step'])"
done; done
