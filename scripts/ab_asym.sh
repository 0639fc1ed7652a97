#!/bin/bash
# A/B of the asymmetric row split of the 256-query filter loop (wm = 0 waves own MBLO of a tile's 16 row blocks):
# variant libraries built with  python -m imagescry_amd.build --variant=<name> -DISC_ASYM_MBLO=<n>  (sym = 8).
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_ALLOW_ABLATION=1
for round in 1 2; do
  for v in ${VARIANTS:-sym a7 a6 a5}; do
    ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_$v.so python3 scripts/quick_search_bench.py ${SHAPES:-10000000x1024} 2>&1 | grep "N=" | sed "s/^/[$v r$round] /"
  done
done
