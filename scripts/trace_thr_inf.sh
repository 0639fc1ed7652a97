#!/bin/bash
# What the filter epilogue costs the Q = 1024 search (ablation build, wrong results in the ablated runs): kernel timeline
#   1. as shipped;  2. filter launches return before their tail (ISC_FILTER_ABL=2): scan + survivor stores only;
#   3. every non-sample level filters against +inf (ISC_THR_INF): no scan, no survivors, an empty tail.
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_ablation.so ISC_ALLOW_ABLATION=1
echo "== real thresholds"; bash scripts/trace_search.sh 10000000x1024
echo "== real thresholds, no tail"; ISC_FILTER_ABL=2 bash scripts/trace_search.sh 10000000x1024
echo "== thresholds +inf"; ISC_THR_INF=1 bash scripts/trace_search.sh 10000000x1024
