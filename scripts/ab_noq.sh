#!/bin/bash
# What do the query operand's LDS bytes cost?  Ablation modes of the 256-query loop on one box, interleaved rounds:
# 12 = production form, 41 = no query staging / fragment reads (garbage B registers), 17 = no fragment reads at all,
# 2 = no staging after the prologue, 7 = MFMAs alone.
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_ALLOW_ABLATION=1
export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_ablation.so
for round in 1 2; do
  for m in ${MODES:-12 41 17 2 7}; do
    ISC_DEBUG_MODE=$m python3 scripts/quick_search_bench.py ${SHAPES:-10000000x1024} 2>&1 | grep "N=" | sed "s/^/[r$round] /"
  done
done
