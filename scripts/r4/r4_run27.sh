#!/bin/bash
# round-4 GPU call 27: the final code -- the default bench line, then the headline trace that reproduces its roofline.frac
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 700 python bench.py > gpurun_out/r4/bench_final.json 2> gpurun_out/r4/bench_final.err || { tail -20 gpurun_out/r4/bench_final.err; echo "bench failed: stop"; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/bench_final.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["unit"], d["ms_per_step"], "frac", d["roofline"]["frac"])
for k in ("encode", "encode_vit_b16", "encode_efficientnet_v2_s"):
    print(k, d[k]["value"], d[k]["ms_per_step"], d[k]["roofline"]["frac"])
PY
bash scripts/trace_headline.sh r04 2>&1 | tail -5
