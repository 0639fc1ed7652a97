#!/bin/bash
# round-4 GPU call 45: the NCHW normalisation as four pixels per lane (one load, one store) -- parity, then A/B
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 400 python -m pytest tests/test_gpu_preprocess.py tests/test_gpu_vit.py -x -q > gpurun_out/r4/t45.log 2>&1 || { tail -30 gpurun_out/r4/t45.log; echo "tests failed: stop"; exit 1; }
tail -2 gpurun_out/r4/t45.log
timeout -k 10 100 python scripts/fuzz_kernels.py 40 45 normalize > gpurun_out/r4/fuzz45.log 2>&1; tail -2 gpurun_out/r4/fuzz45.log
export ISC_ALLOW_ABLATION=1 ISC_LIB=$PWD/imagescry_amd/libimagescry_hip_ablation.so
for r in 1 2; do
  python scripts/quick_normalize_bench.py 2>&1 | grep "NCHW" | sed "s/^/[vec4 r$r] /"
  ISC_NORMALIZE_VEC16=1 python scripts/quick_normalize_bench.py 2>&1 | grep "NCHW" | sed "s/^/[vec16 r$r] /"
done | tee gpurun_out/r4/ab_normalize_nchw.log
