#!/bin/bash
# A/B: non-temporal vs default cache policy on the bank stream of single-query-tile launches (ablation build, same box)
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_ablation.so ISC_ALLOW_ABLATION=1
for round in 1 2 3; do
  echo "== nt"; python3 scripts/quick_search_bench.py "$@" 2>&1 | grep -v amdgpu.ids
  echo "== default policy"; ISC_NO_NT=1 python3 scripts/quick_search_bench.py "$@" 2>&1 | grep -v amdgpu.ids
done
