#!/bin/bash
# round-4 GPU call 36: a longer randomised differential soak on the final code (every kernel family, search plain + deep)
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 520 python scripts/fuzz_kernels.py 35 36 > gpurun_out/r4/fuzz36_kernels.log 2>&1; tail -16 gpurun_out/r4/fuzz36_kernels.log
timeout -k 10 330 python scripts/fuzz_search.py 300 36 > gpurun_out/r4/fuzz36_search.log 2>&1; tail -2 gpurun_out/r4/fuzz36_search.log
timeout -k 10 230 python scripts/fuzz_search.py 200 37 deep > gpurun_out/r4/fuzz36_search_deep.log 2>&1; tail -2 gpurun_out/r4/fuzz36_search_deep.log
