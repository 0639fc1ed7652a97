#!/bin/bash
# round-4 GPU calls 34 and 44: the final code (persistent attention, transposed normalisation stores) -- full GPU
# suite, smoke, the default bench line and the headline trace that reproduces its roofline.frac
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t46.log 2>&1 || { tail -40 gpurun_out/r4/t46.log; echo "GPU suite failed: stop"; exit 1; }
tail -3 gpurun_out/r4/t46.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4/smoke46.log 2>&1 || { tail -20 gpurun_out/r4/smoke46.log; echo "smoke failed: stop"; exit 1; }
tail -2 gpurun_out/r4/smoke46.log
timeout -k 10 400 python bench.py > gpurun_out/r4/bench_final4.json 2> gpurun_out/r4/bench_final4.err || { tail -20 gpurun_out/r4/bench_final4.err; echo "bench failed: stop"; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/bench_final4.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["unit"], d["ms_per_step"], "frac", d["roofline"]["frac"])
for k in ("encode", "encode_vit_b16", "encode_efficientnet_v2_s"):
    print(k, d[k]["value"], d[k]["ms_per_step"], d[k]["roofline"]["frac"], d[k]["roofline"].get("traffic_over_algorithmic"))
PY
bash scripts/trace_headline.sh r04 2>&1 | tail -3
