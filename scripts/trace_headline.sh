#!/bin/bash
# roofline.frac reproducible from profiles/: ONE rocprofv3 kernel trace of the headline command, summarised BY LAUNCH INDEX
# so that warm-up, the bracketed loop and the self-check search are dropped -> profiles/<tag>_headline_trace.json, beside
# the JSON line OF THAT SAME RUN (profiles/<tag>_headline_trace_bench.json).
#   gpurun -- bash scripts/trace_headline.sh r04
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/trace_headline_${TAG}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --no-sweep --no-secondary --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err" &&
python3 scripts/summarize_headline_trace.py "$TAG" "$OUT"
