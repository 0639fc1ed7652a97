#!/bin/bash
# round-4 GPU call 35: the N > 1 path of bench.py on the final code, rehearsed with gloo on one device (ranks share the GPU;
# the default workload at a quarter of the bank so that four shards + workspaces fit beside each other)
mkdir -p gpurun_out/r4
ulimit -c 0
export ISC_BENCH_BACKEND=gloo
timeout -k 10 500 python bench.py --gpus 4 --steps 5 --warmup 2 --no-cpu-baseline --bank-rows 4000000 > gpurun_out/r4/bench_gloo4.json 2> gpurun_out/r4/bench_gloo4.err; echo "rc=$?"
tail -3 gpurun_out/r4/bench_gloo4.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4/bench_gloo4.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "n_gpus", "ms_per_step", "scaling")}, d["selfcheck"], d["config"]["parallelism"])
for k in ("encode", "encode_vit_b16", "encode_efficientnet_v2_s"):
    if k in d: print(k, d[k]["value"])
PY
