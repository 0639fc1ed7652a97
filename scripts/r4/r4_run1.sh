#!/bin/bash
# round-4 GPU call 1: the new tests, the vendor GEMM reference, the full bench line
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_search.py -k "query_dtype or golden or nan_and_inf or full_size_properties_1m" tests/test_gpu_pipeline.py tests/test_gpu_dist.py tests/test_gpu_encoder.py tests/test_gpu_decomposition.py -x -q --durations=15 > gpurun_out/r4/t1.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/t1.log
tail -30 gpurun_out/r4/t1.log
GEMM_VENDOR_REF=1 python scripts/quick_gemm_bench.py > gpurun_out/r4/gemm_ref.log 2>&1
cat gpurun_out/r4/gemm_ref.log
python scripts/quick_attention_bench.py > gpurun_out/r4/att0.log 2>&1; cat gpurun_out/r4/att0.log
python bench.py > gpurun_out/r4/bench1.json 2> gpurun_out/r4/bench1.err
echo "bench rc=$?"; tail -c 600 gpurun_out/r4/bench1.json
