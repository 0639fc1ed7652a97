#!/bin/bash
# round-4 GPU call 33: the persistent sixteen-wave attention kernel -- parity first, then A/B against one workgroup per head
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 400 python -m pytest tests/test_gpu_vit.py -x -q > gpurun_out/r4/t33.log 2>&1 || { tail -30 gpurun_out/r4/t33.log; echo "attention tests failed: stop"; exit 1; }
tail -2 gpurun_out/r4/t33.log
timeout -k 10 200 python scripts/fuzz_kernels.py 60 33 attention > gpurun_out/r4/fuzz33.log 2>&1; tail -3 gpurun_out/r4/fuzz33.log
bash scripts/ab.sh attention -r 3 -a persistent:ablation -a oneshot:ablation:ISC_ATT_ONE_SHOT=1 2>&1 | tee gpurun_out/r4/ab_att_persistent.log
bash scripts/ab.sh encode -r 2 -a persistent:ablation -a oneshot:ablation:ISC_ATT_ONE_SHOT=1 -- vit_b16 512 2>&1 | tee gpurun_out/r4/ab_att_persistent_vit.log
