#!/bin/bash
# round-4 GPU call 3: attention hazard fix + dual conv + GEMM spread-epilogue A/B
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_vit.py tests/test_gpu_encoder.py tests/test_gpu_bench_shapes.py -x -q > gpurun_out/r4/t3.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/t3.log; tail -15 gpurun_out/r4/t3.log
python scripts/quick_attention_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/att2.log
python scripts/quick_encode_bench.py resnet50 512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/enc_resnet.log
VARIANTS="gs80 gs22 gs32 gs21 gs42 gs23 gs40 gs60 gs51" bash scripts/ab_gemm_spread.sh > gpurun_out/r4/ab_gemm_spread.log 2>&1
grep -E "==|qkv|fc1|passed|failed" gpurun_out/r4/ab_gemm_spread.log | tail -80
