#!/bin/bash
# where the cycles of the four-chain RMHMC kernels go: SQ wave-cycle split and instruction mix at 1024 chains
export TMPDIR=/tmp
R=${1:-r02h}
mkdir -p gpurun_out
for wv in 4 2; do
  HTA_TUNING=rmhmc_mfma4_waves=$wv timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -f csv -d gpurun_out/${R}_pmc_w$wv -o p -- python bench.py --workload cfg3 --chains 1024 --traj 100 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  HTA_TUNING=rmhmc_mfma4_waves=$wv timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 -f csv -d gpurun_out/${R}_pmc2_w$wv -o p -- python bench.py --workload cfg3 --chains 1024 --traj 100 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== waves=$wv"; python tools/pmc_summarize.py $(find gpurun_out/${R}_pmc_w$wv gpurun_out/${R}_pmc2_w$wv -name "*counter_collection.csv" | sort) | grep mfma4
done
find gpurun_out -path "*${R}_pmc*" -name "*.csv" -size +1M -delete
