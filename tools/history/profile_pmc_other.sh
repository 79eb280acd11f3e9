#!/bin/bash
# PMC passes (one counter group per run, kernel-trace only) for the cfg3 / cfg4 workloads; run on the GPU box via gpurun.
export TMPDIR=/tmp
R=${1:-r01}
mkdir -p gpurun_out
rocprofv3 -L 2>/dev/null | grep -o -E "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*|SQ_LDS[A-Z_0-9]*|SQ_INSTS_LDS|SQ_INSTS_VALU[A-Z_0-9]*" | sort -u > gpurun_out/${R}_pmc_counter_names.txt
for W in cfg3 cfg4; do
  for G in "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $G | cut -d' ' -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $G -f csv -d gpurun_out/${R}_${W}_pmc_${tag} -o $W -- python bench.py --workload $W --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/${R}_${W}_pmc_${tag}_stdout.txt 2>&1
  done
done
python tools/pmc_summarize.py $(find gpurun_out -name "*counter_collection.csv" | sort) > gpurun_out/${R}_pmc_other_summary.txt; find gpurun_out -name "*_pmc_*" -type d -exec rm -rf {} +; cat gpurun_out/${R}_pmc_other_summary.txt
