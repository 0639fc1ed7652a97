#!/bin/bash
# round-4 GPU call 7: the whole GPU suite, then the full bench line
mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -x -q --durations=20 > gpurun_out/r4/t7.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/t7.log; tail -40 gpurun_out/r4/t7.log
python bench.py > gpurun_out/r4/bench7.json 2> gpurun_out/r4/bench7.err; echo "bench rc=$?"; tail -c 300 gpurun_out/r4/bench7.json
