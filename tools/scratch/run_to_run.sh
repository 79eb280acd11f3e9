export TMPDIR=/tmp
mkdir -p gpurun_out
for k in 1 2 3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> /dev/null | tail -1 > gpurun_out/${TAG:-r06u}_bench_line_$k.json
  python - <<P
import json
j=json.load(open("gpurun_out/${TAG:-r06u}_bench_line_$k.json"))
sec={s["key"]:s for s in j["secondary"]}
print("run $k: cfg2 %.4g (frac %.3f) | cfg3@1024 %.4g | cfg3 %.4g | cfg3-eig %.4g (frac %.3f, busy %s, 1024: %s) | cfg4 %.4g | nbmlp %.4g / %.4g | funnel-hmc %.4g | funnel-rmhmc %.4g" % (
  j["value"], j["roofline"]["frac"], sec["cfg3@1024"]["value"], sec["cfg3"]["value"], sec["cfg3-eig"]["value"], sec["cfg3-eig"]["frac"], sec["cfg3-eig"].get("mfma_busy"), sec["cfg3-eig"]["extras"]["value_1024"],
  sec["cfg4"]["value"], sec["nbmlp"]["value"], sec["nbmlp-full"]["value"], sec["funnel-hmc"]["value"], sec["funnel-rmhmc"]["value"]))
P
done
