#!/bin/bash
# (round 5: the step counts of the workloads are the bench line's own - 10 timed + 2 warm-up for the RMHMC / cfg4 secondaries - so that the
#  average kernel duration of a pass is the duration the line's HIP events see, not that of the first three calls after a cold start)
# tools/physical.sh <tag>: the rocprofv3 passes behind profiles/physical.json (bench.py's `roofline.physical`) and the per-round
# kernel-stat summaries.  One workload per block; per workload: --kernel-trace --stats, then three SEPARATE --pmc passes
# (kernel-trace only beside them).  Run on the GPU box (gpurun); writes gpurun_out/<tag>_*.
export TMPDIR=/tmp
export HTA_BENCH_ESS_EXTRA=0     # (no untimed ESS continuation steps under the profiler: launches per step must be the timed region's)
R=${1:-r03}
O=gpurun_out/${R}_phys
mkdir -p $O
t0=$(date +%s)
# ONLY="cfg3@256 cfg3jacobi@256" tools/physical.sh <tag>: just these workloads (the others' kernels did not change); their entries
# replace the ones in profiles/physical.json, the rest of that file is carried over (each entry names the pass it came from).
KEYS=""
one() {  # key steps warmup traj env -- bench args
  local key=$1 steps=$2 warm=$3 traj=$4 envs=$5; shift 5
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $key "; then return; fi
  KEYS="$KEYS $key=$O/$key"
  local d=$O/$key; mkdir -p $d
  local cmd="python bench.py --no-cpu-baseline --no-secondary --no-api --steps $steps --warmup $warm $*"
  echo "{\"command\": \"$envs $cmd\", \"steps\": $steps, \"warmup\": $warm, \"traj\": $traj, \"source\": \"tools/physical.sh $R\"}" > $d/meta.json
  env $envs timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $d/stats -o s -- $cmd > $d/bench.json 2> /dev/null
  env $envs timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $d/pmc_rd -o r -- $cmd > /dev/null 2>&1
  env $envs timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $d/pmc_wr -o w -- $cmd > /dev/null 2>&1
  env $envs timeout 120 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $d/pmc_sq -o q -- $cmd > /dev/null 2>&1
  case $key in funnel*)      # the compiled-callback kernels are VALU code: instruction count and VALU-active share of the waves' cycles
    env $envs timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES -f csv -d $d/pmc_valu -o v -- $cmd > /dev/null 2>&1;;
  esac
  cp $(find $d/stats -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_${key}_kernel_stats.csv 2> /dev/null
  echo "$key done ($(( $(date +%s) - t0 )) s)"
}
one cfg2@1024 20 3 1000 X=1
one cfg3@1024 20 5 100 X=1 --workload cfg3@1024
one cfg3@256 10 2 400 X=1 --workload cfg3
one cfg3jacobi@256 3 1 20 HTA_RMHMC_FUSED=0 --workload cfg3 --traj 20
one cfg4@512 10 2 20 X=1 --workload cfg4
one nbmlp@1024 6 2 1 X=1 --workload nbmlp
one nbmlp-full@1024 6 2 1 X=1 --workload nbmlp-full
one funnel-hmc@1024 10 2 200 X=1 --workload funnel-hmc
one funnel-rmhmc@1024 5 1 2 X=1 --workload funnel-rmhmc
python tools/physical.py $KEYS > gpurun_out/${R}_physical_new.json
python - <<P
import json
new = json.load(open("gpurun_out/${R}_physical_new.json"))
old = json.load(open("profiles/physical.json")) if "$ONLY" else {}
old.update(new)
json.dump(old, open("gpurun_out/${R}_physical.json", "w"), indent=1)
P
python tools/pmc_summarize.py $(find $O -name "*counter_collection.csv" | sort) > gpurun_out/${R}_pmc_all.txt
# keep the merge small: raw traces stay on the box
find $O -name "*.csv" -size +2M -delete
echo "physical done ($(( $(date +%s) - t0 )) s)"; head -c 1500 gpurun_out/${R}_physical.json
