#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_ablation.so ISC_ALLOW_ABLATION=1
for seg in 0 8 4 2; do
  echo "== ISC_GEMM_SEG=$seg"; ISC_GEMM_SEG=$seg python3 scripts/quick_gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
