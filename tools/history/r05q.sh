#!/bin/bash
# round 5, call q: run-to-run spread of the driver's command - three complete default runs back to back on one box
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> /dev/null | tail -1 > gpurun_out/r05q_bench_line_$i.json
done
python - <<'P'
import json
rows = [json.load(open("gpurun_out/r05q_bench_line_%d.json" % i)) for i in (1, 2, 3)]
print("cfg2 value", ["%.4g" % r["value"] for r in rows], "frac", [r["roofline"]["frac"] for r in rows])
for k in range(len(rows[0]["secondary"])):
    print(rows[0]["secondary"][k]["key"], ["%.4g" % r["secondary"][k]["value"] for r in rows], "frac", [r["secondary"][k]["frac"] for r in rows],
          "ess_x_cpu", [r["secondary"][k].get("ess_per_sec_vs_cpu") for r in rows])
P
