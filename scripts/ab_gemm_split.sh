#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_ablation.so ISC_ALLOW_ABLATION=1
for round in 1 2; do
  echo "== split (waves 0-3 issue the LDS-DMA, stores bunched)"; python3 scripts/quick_gemm_bench.py 2>&1 | grep -v amdgpu.ids
  echo "== no split (every wave issues, deferred epilogue)"; ISC_GEMM_NO_SPLIT=1 python3 scripts/quick_gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
