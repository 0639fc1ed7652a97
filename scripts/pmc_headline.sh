#!/bin/bash
# HBM traffic of the headline search step alone (bench.py defaults, no sweep / secondary / CPU baseline):
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (kernel trace only, as the pool requires).
# Run through gpurun from the repo root; then `python scripts/summarize_headline_traffic.py <tag>` on the host.
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/pmc_headline_${TAG}
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-sweep"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err" &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err"
echo "pmc_headline exit $?"
