#!/bin/bash
# round-4 GPU call 28: per-layer traces and HBM traffic of the two convolutional encoders on the final code
mkdir -p gpurun_out/r4
ulimit -c 0
bash scripts/trace_encode_layers.sh > gpurun_out/r4/resnet_layers_final.txt 2>&1; grep "total conv" gpurun_out/r4/resnet_layers_final.txt
bash scripts/trace_effnet_layers.sh > gpurun_out/r4/effnet_layers_final.txt 2>&1; grep "total conv" gpurun_out/r4/effnet_layers_final.txt
bash scripts/pmc_encoder.sh r04 resnet50 && bash scripts/pmc_encoder.sh r04 efficientnet_s
