#!/bin/bash
# round-4 GPU call 8: randomised differential runs on the changed kernels; ResNet-50 traffic counters
mkdir -p gpurun_out/r4
python scripts/fuzz_search.py 150 41 > gpurun_out/r4/fuzz_search.log 2>&1; tail -6 gpurun_out/r4/fuzz_search.log
python scripts/fuzz_search.py 150 42 deep > gpurun_out/r4/fuzz_search_deep.log 2>&1; tail -6 gpurun_out/r4/fuzz_search_deep.log
python scripts/fuzz_kernels.py 60 7 attention,conv_dual,conv,gemm > gpurun_out/r4/fuzz_kernels.log 2>&1; tail -8 gpurun_out/r4/fuzz_kernels.log
bash scripts/pmc_encoder.sh r04 resnet50 2>&1 | tail -2
