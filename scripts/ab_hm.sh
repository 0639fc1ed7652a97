#!/bin/bash
# A/B of the half-major form of the 256-query loop (ISC_DEBUG_MODE 46 of ablation / variant builds) against the form as
# built (12): variant libraries from  python -m imagescry_amd.build --variant=<name> -D...  (hm0 = defaults).
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_ALLOW_ABLATION=1
for round in 1 2; do
  for v in ${VARIANTS:-hm0 hm_s8 hm_d0 hm_d2 hm_b2}; do
    for m in ${MODES:-46}; do
      ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_$v.so ISC_DEBUG_MODE=$m python3 scripts/quick_search_bench.py ${SHAPES:-10000000x1024} 2>&1 | grep "N=" | sed "s/^/[$v r$round] /"
    done
  done
  ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_hm0.so ISC_DEBUG_MODE=12 python3 scripts/quick_search_bench.py ${SHAPES:-10000000x1024} 2>&1 | grep "N=" | sed "s/^/[base r$round] /"
done
