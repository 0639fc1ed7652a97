#!/bin/bash
# round-4 GPU call 4: the 128-query tile (tests + A/B against the 64- and 256-query tiles), headline trace, ResNet layers
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_search.py -x -q > gpurun_out/r4/t4.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/t4.log; tail -8 gpurun_out/r4/t4.log
bash scripts/ab.sh search -r 2 -a t128:ablation -a t64:ablation:ISC_FORCE_TILE=64 -a t256:ablation:ISC_FORCE_TILE=256 -- 10000000x128 10000000x100 10000000x65 1250000x128 2>&1 | tee gpurun_out/r4/ab_t128.log
bash scripts/trace_headline.sh r04 2>&1 | tail -5
bash scripts/trace_encode_layers.sh > gpurun_out/r4/resnet_layers.txt 2>&1; tail -30 gpurun_out/r4/resnet_layers.txt
