#!/bin/bash
# Profiles the bench on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats          -> per-kernel time
#   2. rocprofv3 --pmc FETCH_SIZE                -> HBM read bytes   (separate pass: TCC slots)
#   3. rocprofv3 --pmc WRITE_SIZE                -> HBM write bytes
#   4. rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -> matrix-core utilisation
# Counter passes use --kernel-trace only (no sys/hip/hsa tracing), as the pool requires.
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/prof_${TAG}
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_stats.json" 2> "$OUT/stats.err" &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err" &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err" &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python3 bench.py $ARGS > "$OUT/bench_mfma.json" 2> "$OUT/mfma.err"
echo "profile exit $?"
find "$OUT" -name '*.csv' | head -30
