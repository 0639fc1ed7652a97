#!/bin/bash
# A/B of headline-kernel variants at Q = 1024 (ablation build, same box, interleaved)
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_ablation.so ISC_ALLOW_ABLATION=1
for round in 1 2 3; do
  echo "== base"; python3 scripts/quick_search_bench.py 10000000x1024 2>&1 | grep -v amdgpu.ids
  echo "== no static prio"; ISC_NO_STATIC_PRIO=1 python3 scripts/quick_search_bench.py 10000000x1024 2>&1 | grep -v amdgpu.ids
  echo "== split"; ISC_FORCE_SPLIT=1 python3 scripts/quick_search_bench.py 10000000x1024 2>&1 | grep -v amdgpu.ids
done
