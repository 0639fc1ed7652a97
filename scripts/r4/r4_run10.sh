#!/bin/bash
# round-4 GPU call 10: L2 touch-ahead of the single-query-tile SPLIT form -- correctness first, then the A/B
mkdir -p gpurun_out/r4
ulimit -c 0
export ISC_ALLOW_ABLATION=1
ISC_LIB=$PWD/imagescry_amd/libimagescry_hip_ta2.so timeout -k 10 300 python -m pytest tests/test_gpu_search.py -x -q -k "oracle or golden or full_size_properties_1m or ties" > gpurun_out/r4/t10.log 2>&1 || { tail -20 gpurun_out/r4/t10.log; echo "touch-ahead build failed its tests: stop"; exit 1; }
tail -3 gpurun_out/r4/t10.log
bash scripts/ab.sh search -r 3 -a ta0:ta0 -a ta1:ta1 -a ta2:ta2 -a ta4:ta4 -- 10000000x256 10000000x192 1250000x256 2>&1 | tee gpurun_out/r4/ab_touch.log
