#!/bin/bash
# HBM traffic of the ResNet-50 encode step (bench.py --workload encode): FETCH_SIZE / WRITE_SIZE in separate passes.
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/pmc_encode_${TAG}
ARGS="--workload encode --steps 3 --warmup 1 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench.py $ARGS > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err" &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 bench.py $ARGS > "$OUT/bench_write.json" 2> "$OUT/write.err"
echo "pmc_encode exit $?"
