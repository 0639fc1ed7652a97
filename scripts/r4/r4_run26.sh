#!/bin/bash
# round-4 GPU call 26: the final code -- full GPU suite, smoke, fuzz of the convolution family (half tiles for K <= 256 are new)
mkdir -p gpurun_out/r4
ulimit -c 0
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r4/t26.log 2>&1 || { tail -40 gpurun_out/r4/t26.log; echo "GPU suite failed: stop"; exit 1; }
tail -3 gpurun_out/r4/t26.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4/smoke26.log 2>&1 || { tail -20 gpurun_out/r4/smoke26.log; echo "smoke failed: stop"; exit 1; }
tail -2 gpurun_out/r4/smoke26.log
timeout -k 10 300 python scripts/fuzz_kernels.py 40 26 conv,conv_dual > gpurun_out/r4/fuzz26.log 2>&1; tail -6 gpurun_out/r4/fuzz26.log
