#!/bin/bash
# phase spread of the streaming GEMM's workgroups (ablation build): ISC_GEMM_PHASE = s_sleep(127) units per phase group
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_ablation.so ISC_ALLOW_ABLATION=1
for round in 1 2; do
for cfg in "-1 4" "1 4" "2 4" "3 4" "2 8" "1 8" "4 2"; do
  set -- $cfg
  echo "== round $round ISC_GEMM_PHASE=$1 groups=$2"; ISC_GEMM_PHASE=$1 ISC_GEMM_PHASE_GROUPS=$2 python3 scripts/quick_gemm_bench.py 2>&1 | grep -E "proj|fc2"
done
done
