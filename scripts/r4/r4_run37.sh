#!/bin/bash
# round-4 GPU call 37: the branch-free 3 x 3 / 2 max-pool -- parity, then the ResNet-50 step
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_bench_shapes.py -x -q > gpurun_out/r4/t37.log 2>&1 || { tail -30 gpurun_out/r4/t37.log; echo "tests failed: stop"; exit 1; }
tail -2 gpurun_out/r4/t37.log
bash scripts/trace_encode_layers.sh > gpurun_out/r4/resnet_layers_maxpool.txt 2>&1; grep "total conv\|k_maxpool" gpurun_out/r4/resnet_layers_maxpool.txt
python scripts/quick_encode_bench.py resnet50 512 2>&1 | grep -v amdgpu.ids
