#!/bin/bash
# round-4 GPU call 24: halo test cases; short-K residual layers as half tiles at three workgroups per CU (A/B)
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_efficientnet.py tests/test_gpu_bench_shapes.py -x -q > gpurun_out/r4/t24.log 2>&1 || { tail -30 gpurun_out/r4/t24.log; echo "tests failed: stop"; exit 1; }
tail -2 gpurun_out/r4/t24.log
bash scripts/ab.sh encode -r 2 -a base:ablation -a h2:ablation:ISC_CONV_HALVES_KSTEPS=2 -a h4:ablation:ISC_CONV_HALVES_KSTEPS=4 -a h8:ablation:ISC_CONV_HALVES_KSTEPS=8 -a h8any:ablation:ISC_CONV_HALVES_KSTEPS=8,ISC_CONV_HALVES_ANY=1 -- resnet50 512 2>&1 | tee gpurun_out/r4/ab_halves_resnet.log
bash scripts/ab.sh encode -r 2 -a base:ablation -a h4:ablation:ISC_CONV_HALVES_KSTEPS=4 -a h8:ablation:ISC_CONV_HALVES_KSTEPS=8 -a h8any:ablation:ISC_CONV_HALVES_KSTEPS=8,ISC_CONV_HALVES_ANY=1 -a h16any:ablation:ISC_CONV_HALVES_KSTEPS=16,ISC_CONV_HALVES_ANY=1 -- effnet_s 512 2>&1 | tee gpurun_out/r4/ab_halves_effnet.log
