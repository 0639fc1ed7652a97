#!/bin/bash
# A/B: launches per level at Q = 1024 (partner realignment at kernel boundaries), ablation build, same box, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_ablation.so ISC_ALLOW_ABLATION=1
for round in 1 2; do
 for seg in 100000 256 128 64 32; do
  echo "== ISC_SEG_TILES=$seg"; ISC_SEG_TILES=$seg python3 scripts/quick_search_bench.py 10000000x1024 2>&1 | grep -v amdgpu.ids
 done
done
