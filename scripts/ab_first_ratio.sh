#!/bin/bash
# A/B of the intermediate first level (make_plan: first_ratio) at Q = 1024 and Q = 256 (ablation build, same box, interleaved)
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_ablation.so ISC_ALLOW_ABLATION=1
for round in 1 2; do
  for r in 1048576 8 4 16; do
    echo "== first ratio $r (applies to one query tile only)"; ISC_FIRST_RATIO=$r python3 scripts/quick_search_bench.py 10000000x1024 10000000x256 2>&1 | grep -v amdgpu.ids
  done
done
