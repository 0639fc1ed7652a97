#!/bin/bash
# round-4 GPU call 29: does the streaming GEMM follow its quantisation model?  (qkv: 16 tiles per workgroup with XCD groups of 3
# feature blocks, 15 with every feature block on its own) ; half tiles up to 16 K steps on ResNet-50
mkdir -p gpurun_out/r4
ulimit -c 0
bash scripts/ab.sh gemm -r 3 -a g3:ablation -a g1:ablation:ISC_GEMM_GROUP=1 2>&1 | tee gpurun_out/r4/ab_gemm_group.log
bash scripts/ab.sh encode -r 2 -a h8:ablation -a h16:ablation:ISC_CONV_HALVES_KSTEPS=16 -a h0:ablation:ISC_CONV_HALVES_KSTEPS=0 -- resnet50 512 2>&1 | tee gpurun_out/r4/ab_halves16_resnet.log
