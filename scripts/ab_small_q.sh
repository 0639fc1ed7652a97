#!/bin/bash
# A/B of the HBM-bound search shapes: this tree against the round-1 tree (git worktree _r1/), same box, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
for round in 1 2; do
  for tree in . _r1; do
    echo "== tree $tree round $round"
    python3 $tree/scripts/quick_search_bench.py 10000000x1 10000000x64 1250000x1 1250000x16 1250000x64 2>&1 | grep -v amdgpu.ids
  done
done
