#!/bin/bash
# round-4 GPU call 30: the streaming GEMM's main loop (no stores) beside the search loop on GEMM-like shapes, one device
mkdir -p gpurun_out/r4
ulimit -c 0
bash scripts/ab.sh gemm -r 2 -a d0:ablation -a nostore:ablation:ISC_GEMM_DEBUG=1 -a nostore_l2hot:ablation:ISC_GEMM_DEBUG=3 -a nostore_nosplit:ablation:ISC_GEMM_DEBUG=1,ISC_GEMM_NO_SPLIT=1 2>&1 | tee gpurun_out/r4/ab_gemm_mainloop.log
bash scripts/ab.sh search -r 2 -a base:ablation -a thrinf:ablation:ISC_THR_INF=1 -- 1008640x768 1008640x1024 2>&1 | tee gpurun_out/r4/ab_search_as_gemm.log
