#!/bin/bash
# round 3, step j: Philox with v_mad_u64_u32 - micro-benchmark, then the bench line against the r03i numbers
cd /root/repo; mkdir -p gpurun_out/r03j
./tools/scratch/philox_rate.bin > gpurun_out/r03j/philox_rate.txt 2>&1; cat gpurun_out/r03j/philox_rate.txt
timeout 900 python -m pytest tests/test_gpu_hmc.py tests/test_gpu_rmhmc.py -x -q -m gpu -k "philox or gibbs or sample_rmhmc_vs_oracle or cfg2 or vs_oracle" > gpurun_out/r03j/tests.txt 2>&1; tail -3 gpurun_out/r03j/tests.txt
timeout 600 python bench.py --no-cpu-baseline --no-api > gpurun_out/r03j/bench_stdout.txt 2> gpurun_out/r03j/bench_stderr.txt
tail -1 gpurun_out/r03j/bench_stdout.txt > gpurun_out/r03j/bench_line.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03j/bench_line.json").read())
print("cfg2", d["value"], d["ms_per_step"])
for e in d.get("secondary", []):
    print(e.get("key"), e.get("value"), e.get("ms_per_step"), "frac", e.get("frac"), e.get("error"))
PY
