#!/bin/bash
# round-4 GPU call 2: the whole GPU suite on the ABI-4 / attention changes, attention timing
mkdir -p gpurun_out/r4
python scripts/quick_attention_bench.py > gpurun_out/r4/att1.log 2>&1; cat gpurun_out/r4/att1.log
python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/r4/t2.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/t2.log
tail -45 gpurun_out/r4/t2.log
