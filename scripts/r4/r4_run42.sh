#!/bin/bash
# round-4 GPU call 42: the tree as committed -- full GPU suite and smoke
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t42.log 2>&1 || { tail -40 gpurun_out/r4/t42.log; echo "GPU suite failed"; exit 1; }
tail -2 gpurun_out/r4/t42.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
