#!/bin/bash
# round-4 GPU call 38: a smaller sample level on a 1.25 M-row shard (one main level still suffices there): does the shorter
# sample launch buy more than the weaker threshold costs?  + the max-pool shape tests on the generic kernel
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 200 python -m pytest tests/test_gpu_encoder.py -x -q -k "maxpool" > gpurun_out/r4/t38.log 2>&1 || { tail -20 gpurun_out/r4/t38.log; exit 1; }
tail -1 gpurun_out/r4/t38.log
bash scripts/ab.sh search -r 3 -a s256:ablation -a s128:ablation:ISC_SAMPLE_TILES=128 -a s64:ablation:ISC_SAMPLE_TILES=64 -a s40:ablation:ISC_SAMPLE_TILES=40 -- 1250000x1 1250000x16 1250000x64 2>&1 | tee gpurun_out/r4/ab_sample_tiles.log
