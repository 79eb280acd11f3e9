#!/bin/bash
# the driver's round-end sequence on the box: whole GPU suite, smoke, the default bench line (with its detail record)
export TMPDIR=/tmp
R=${1:-r04i}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=10 > gpurun_out/${R}_gpu_tests.txt 2>&1; tail -4 gpurun_out/${R}_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.txt 2>&1; tail -1 gpurun_out/${R}_smoke.txt
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/${R}_bench_stdout.txt 2> gpurun_out/${R}_bench_stderr.txt
tail -1 gpurun_out/${R}_bench_stdout.txt > gpurun_out/${R}_bench_line.json; cp bench_detail.json gpurun_out/${R}_bench_detail.json
tail -4 gpurun_out/${R}_bench_stderr.txt; python - <<P
import json
j=json.load(open("gpurun_out/${R}_bench_line.json"))
print("value %.4g ms %.4f frac %.3f cpu %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j.get("cpu_baseline",{}).get("value")))
for s in j.get("secondary", []):
    print(s.get("key"), s.get("value"), s.get("frac"), s.get("mfma_busy"), s.get("kernel"), s.get("cpu"), s.get("extras"), s.get("error"))
print(len(json.dumps(j)))
P
