#!/bin/bash
# A/B runs of bench.py --workload cfg3: every argument after the tag is "<chains>:<HTA_TUNING string or ->"; run on the GPU
# box via gpurun, one line per combination in gpurun_out/<tag>_ab.txt.
export TMPDIR=/tmp
R=$1; shift
mkdir -p gpurun_out
: > gpurun_out/${R}_ab.txt
for combo in "$@"; do
  C=${combo%%:*}; TUN=${combo#*:}; [ "$TUN" = "-" ] && TUN=""
  HTA_TUNING=$TUN timeout 120 python bench.py --workload cfg3 --chains $C --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${R}_tmp.json 2>> gpurun_out/${R}_ab.err
  python - gpurun_out/${R}_tmp.json $C "$TUN" >> gpurun_out/${R}_ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("chains %5s %-34s %.3e steps/s, %.2f ms/step, acc %.4f" % (sys.argv[2], sys.argv[3] or "(default)", d["value"], d["ms_per_step"], d["acceptance_rate"]))
PY
done
rm -f gpurun_out/${R}_tmp.json
cat gpurun_out/${R}_ab.txt
