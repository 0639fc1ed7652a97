#!/bin/bash
# round-4 final GPU call: smoke, the whole GPU suite, the full bench line, the headline trace
mkdir -p gpurun_out/r4
python __graft_entry__.py smoke > gpurun_out/r4/smoke.log 2>&1; tail -2 gpurun_out/r4/smoke.log
python -m pytest tests -m gpu -x -q > gpurun_out/r4/t_final.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/t_final.log; tail -4 gpurun_out/r4/t_final.log
python bench.py > gpurun_out/r4/bench_final.json 2> gpurun_out/r4/bench_final.err; echo "bench rc=$?"
bash scripts/trace_headline.sh r04 2>&1 | tail -2
