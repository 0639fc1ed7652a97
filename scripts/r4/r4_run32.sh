#!/bin/bash
# round-4 GPU call 32: rocprofv3 --kernel-trace --stats of the default bench line (no sweep, no CPU baseline) on the final code
mkdir -p gpurun_out/r4
ulimit -c 0
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_r04_stats
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sweep > "$OUT/bench_stats.json" 2> "$OUT/stats.err"
echo "profile exit $?"
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1); echo "$f"; head -25 "$f" | cut -c1-200
