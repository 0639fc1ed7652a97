#!/bin/bash
# round-4 GPU call 23: halo-tile convolution, second version (residual requested up front, pipelined fragment reads)
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_efficientnet.py -x -q > gpurun_out/r4/t23.log 2>&1 || { tail -30 gpurun_out/r4/t23.log; echo "halo convolution failed its tests: stop"; exit 1; }
tail -2 gpurun_out/r4/t23.log
python scripts/fuzz_kernels.py 40 12 conv > gpurun_out/r4/fuzz_conv_halo2.log 2>&1; tail -5 gpurun_out/r4/fuzz_conv_halo2.log
bash scripts/ab.sh encode -r 2 -a halo:ablation -a packedk:ablation:ISC_CONV_NO_HALO=1 -- effnet_s 512 2>&1 | tee gpurun_out/r4/ab_halo_effnet.log
bash scripts/ab.sh encode -r 2 -a halo:ablation -a packedk:ablation:ISC_CONV_NO_HALO=1 -- resnet50 512 2>&1 | tee gpurun_out/r4/ab_halo_resnet.log
bash scripts/trace_effnet_layers.sh > gpurun_out/r4/effnet_layers.txt 2>&1; head -4 gpurun_out/r4/effnet_layers.txt; grep "total conv" gpurun_out/r4/effnet_layers.txt
bash scripts/trace_encode_layers.sh > gpurun_out/r4/resnet_layers2.txt 2>&1; head -2 gpurun_out/r4/resnet_layers2.txt; grep "total conv" gpurun_out/r4/resnet_layers2.txt
