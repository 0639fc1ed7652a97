#!/bin/bash
# round-4 GPU call 31: randomised differential runs on the final code (all kernel families, search plain + deep)
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 420 python scripts/fuzz_kernels.py 25 31 > gpurun_out/r4/fuzz31_kernels.log 2>&1; tail -16 gpurun_out/r4/fuzz31_kernels.log
timeout -k 10 200 python scripts/fuzz_search.py 120 31 > gpurun_out/r4/fuzz31_search.log 2>&1; tail -4 gpurun_out/r4/fuzz31_search.log
timeout -k 10 200 python scripts/fuzz_search.py 100 32 deep > gpurun_out/r4/fuzz31_search_deep.log 2>&1; tail -4 gpurun_out/r4/fuzz31_search_deep.log
