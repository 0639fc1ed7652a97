#!/bin/bash
# round-4 GPU call 20: the halo-tile convolution (conv_halo.hip) -- parity first, then same-device A/B of the encoders
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_efficientnet.py -x -q > gpurun_out/r4/t20.log 2>&1 || { tail -30 gpurun_out/r4/t20.log; echo "halo convolution failed its tests: stop"; exit 1; }
tail -3 gpurun_out/r4/t20.log
python scripts/fuzz_kernels.py 60 11 conv > gpurun_out/r4/fuzz_conv_halo.log 2>&1; tail -6 gpurun_out/r4/fuzz_conv_halo.log
bash scripts/ab.sh encode -r 2 -a halo:ablation -a packedk:ablation:ISC_CONV_NO_HALO=1 -- effnet_s 512 2>&1 | tee gpurun_out/r4/ab_halo_effnet.log
bash scripts/ab.sh encode -r 2 -a halo:ablation -a packedk:ablation:ISC_CONV_NO_HALO=1 -- resnet50 512 2>&1 | tee gpurun_out/r4/ab_halo_resnet.log
