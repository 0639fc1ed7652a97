#!/bin/bash
# round-4 GPU call 40 / 41: persistent attention variants (dedicated staging waves; two query blocks per wave) -- parity first, then A/B
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 400 python -m pytest tests/test_gpu_vit.py tests/test_gpu_bench_shapes.py -x -q > gpurun_out/r4/t40.log 2>&1 || { tail -30 gpurun_out/r4/t40.log; echo "tests failed: stop"; exit 1; }
tail -2 gpurun_out/r4/t40.log
timeout -k 10 150 python scripts/fuzz_kernels.py 60 40 attention > gpurun_out/r4/fuzz40.log 2>&1; tail -3 gpurun_out/r4/fuzz40.log
bash scripts/ab.sh attention -r 3 -a persistent:ablation -a compute_only:ablation:ISC_ATT_ABL=4 -a oneshot:ablation:ISC_ATT_ONE_SHOT=1 2>&1 | tee gpurun_out/r4/ab_att_stagers.log
bash scripts/ab.sh encode -r 2 -a persistent:ablation -a oneshot:ablation:ISC_ATT_ONE_SHOT=1 -- vit_b16 512 2>&1 | grep "ms/step" | tee gpurun_out/r4/ab_att_stagers_vit.log
