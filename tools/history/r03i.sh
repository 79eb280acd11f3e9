#!/bin/bash
# round 3, step i: phase attribution of rmhmc_mfma4x4_kernel after the split momentum draw; the default bench line with API timings
cd /root/repo; mkdir -p gpurun_out/r03i
timeout 300 python tools/scratch/x4_time.py 1024 8 > gpurun_out/r03i/x4_time.txt 2>&1
cat gpurun_out/r03i/x4_time.txt | tail -24
timeout 600 python bench.py > gpurun_out/r03i/bench_stdout.txt 2> gpurun_out/r03i/bench_stderr.txt
tail -1 gpurun_out/r03i/bench_stdout.txt > gpurun_out/r03i/bench_line.json
wc -c gpurun_out/r03i/bench_line.json
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03i/bench_line.json").read())
print("cfg2", d["value"], d["ms_per_step"], "api", d.get("api_ms_per_step"))
for e in d.get("secondary", []):
    print(e.get("key"), e.get("value"), e.get("ms_per_step"), "api_ms", e.get("api_ms"), "frac", e.get("frac"), e.get("error"))
full=json.loads(open("gpurun_out/bench_detail.json").read())
for r in full.get("secondary", []):
    print(r.get("key"), r.get("api_ms_per_step"), r.get("api_sync_ms"), r.get("api_route"), r.get("api_error"))
PY
