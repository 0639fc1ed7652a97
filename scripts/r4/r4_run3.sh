#!/bin/bash
# round-4 GPU call 3: attention hazard fix + dual conv + GEMM spread-epilogue A/B
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_vit.py tests/test_gpu_encoder.py tests/test_gpu_bench_shapes.py -x -q > gpurun_out/r4/t3.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/t3.log; tail -15 gpurun_out/r4/t3.log
python scripts/quick_attention_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/att2.log
python scripts/quick_encode_bench.py resnet50 512 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/enc_resnet.log
# (variant libraries: python -m imagescry_amd.build --variant=gs<NI><NH> -DISC_GS_NI=<NI> -DISC_GS_NH=<NH> on the tree with
#  scripts/experiments/gemm_spread_epilogue.patch applied; every variant first passed tests/test_gpu_vit.py::test_gemm_f16)
bash scripts/ab.sh gemm -r 3 -a gs80:gs80 -a gs22:gs22 -a gs32:gs32 -a gs21:gs21 -a gs42:gs42 -a gs23:gs23 -a gs40:gs40 -a gs60:gs60 -a gs51:gs51 > gpurun_out/r4/ab_gemm_spread.log 2>&1
grep -E "==|qkv|fc1|passed|failed" gpurun_out/r4/ab_gemm_spread.log | tail -80
