#!/bin/bash
# round 6: instruction / stall counters of the fused RMHMC callback kernel (funnel, 1024 chains) - separate PMC passes, kernel trace only.
# usage (on the GPU box): tools/jit_rmhmc_pmc.sh <outdir> [chains]
out=$1; chains=${2:-1024}
cd /tmp && export TMPDIR=/tmp
cmd="python $GRAFT_REPO_ROOT/bench.py --workload funnel-rmhmc --chains $chains --no-cpu-baseline --no-secondary --steps 2 --warmup 1"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -f csv -d $out/$tag -o p -- $cmd > /dev/null 2>&1
done
python - "$out" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in tot.items():
    if "hta_cb" in k or "metric" in k:
        print(k, {a: b for a, b in sorted(d.items())})
PY
