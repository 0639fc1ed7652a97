#!/bin/bash
# ablations of the streaming GEMM (ablation build): where the time goes
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_ablation.so ISC_ALLOW_ABLATION=1
for dbg in 0 1 2 3 4; do
  echo "== ISC_GEMM_DEBUG=$dbg"; ISC_GEMM_DEBUG=$dbg python3 scripts/quick_gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
