# tests/test_gpu_rmhmc.py with the parity-partner modes of the matrix-core metric kernel as process defaults (profiles/r06r, r06x)
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/modes.txt
for m in "metric_bx3=1" "metric_bx3=0" "metric_resident=0" "metric_sqrtdraw=0" "metric_second=0" "metric_resident=0,metric_bx3=0,metric_sqrtdraw=0"; do
  r=$(HTA_TUNING_DEFAULTS=$m timeout 600 python -m pytest tests/test_gpu_rmhmc.py -x -q -m gpu 2>&1 | tail -1)
  printf "== %-52s %s\n" "$m" "$r" >> gpurun_out/modes.txt
done
cat gpurun_out/modes.txt
