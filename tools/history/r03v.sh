#!/bin/bash
# Round 3, last GPU pass (one call, ~10 GPU-minutes left): every step under `timeout`, most informative first.
#   1. cfg2 A/B of the quad-kernel variants (tuning key quad_variant 0 / 3 / 7) through bench.py
#   2. the whole -m gpu suite with the variant as the process default (HTA_TUNING_DEFAULTS=quad_variant=7): every other kernel
#      runs exactly as before, the Gaussian-HMC tests run on the new instance, the bit-identity test compares it with the old
#   3. the driver's bench command (default keys)
#   4. RCCL world-1 collectives, rocprofv3 kernel stats of cfg2 under both instances
export TMPDIR=/tmp
R=${1:-r03v}
mkdir -p gpurun_out
t0=$(date +%s)
el() { echo $(( $(date +%s) - t0 )); }
stamp() { echo "[$(el) s] $*" >> gpurun_out/${R}_timeline.txt; }
stamp start
timeout 240 python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/${R}_device.txt 2>&1
stamp "torch import rc=$?"
# (variant, prepared eig block): (0,0) = the library before this session
for vp in 0:0 0:1 3:1 7:1 7:0; do
  v=${vp%%:*}; pz=${vp##*:}
  echo "quad_variant=$v HTA_BENCH_PREPARE=$pz" >> gpurun_out/${R}_ab_lines.txt
  HTA_BENCH_PREPARE=$pz HTA_TUNING=quad_variant=$v timeout 90 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2> gpurun_out/${R}_ab_v${v}_p${pz}.err | tail -1 >> gpurun_out/${R}_ab_lines.txt
  stamp "ab v=$v prepare=$pz rc=$?"
done
HTA_TUNING_DEFAULTS=quad_variant=7 timeout 420 python -m pytest tests -m gpu -q --durations=10 > gpurun_out/${R}_gpu_tests_variant7.txt 2>&1
stamp "suite under quad_variant=7 rc=$?"
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_stdout.txt 2> gpurun_out/${R}_bench.err
stamp "driver bench rc=$?"
tail -1 gpurun_out/${R}_bench_stdout.txt > gpurun_out/${R}_bench_line.json
cp bench_detail.json gpurun_out/${R}_bench_detail.json 2>/dev/null
timeout 90 python tools/rccl_world1.py > gpurun_out/${R}_rccl_world1.json 2> gpurun_out/${R}_rccl_world1.err
stamp "rccl world-1 rc=$?"
for v in 7 0; do
  HTA_TUNING=quad_variant=$v timeout 120 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/${R}_prof_v$v -o cfg2 -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-api > gpurun_out/${R}_prof_v$v.log 2>&1
  stamp "rocprof v=$v rc=$?"
  f=$(find gpurun_out/${R}_prof_v$v -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -8 "$f" > gpurun_out/${R}_cfg2_v${v}_kernel_stats.csv
  rm -rf gpurun_out/${R}_prof_v$v
done
stamp end
