#!/bin/bash
# Where the waves of the headline search kernel spend their cycles (SQ counters, one pass, kernel trace only).
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/pmc_stalls_${TAG}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/sq" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sweep > "$OUT/bench.json" 2> "$OUT/err.txt"
echo "pmc_stalls exit $?"
