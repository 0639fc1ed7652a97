#!/bin/bash
# A/B of the streaming GEMM's spread epilogue (gemm_stream.hip: NI row blocks written at the tile boundary, NH held in
# registers, the rest written in place during the next tile's first K step).  Variant libraries:
#   python -m imagescry_amd.build --variant=gs<NI><NH> -DISC_GS_NI=<NI> -DISC_GS_NH=<NH>      (gs80 = the bunched epilogue)
# Interleaved rounds on one device; correctness of every variant first (tests/test_gpu_vit.py::test_gemm_f16, packed).
cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_ALLOW_ABLATION=1
VARIANTS=${VARIANTS:-"gs80 gs32 gs21 gs42 gs23 gs40 gs60 gs51"}
for v in $VARIANTS; do
  export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_$v.so
  echo "== $v correctness"; python3 -m pytest tests/test_gpu_vit.py -q -x -k "test_gemm_f16 and packed and not tile" 2>&1 | tail -2
done
for round in 1 2 3; do
  for v in $VARIANTS; do
    export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_$v.so
    echo "== round $round $v"; python3 scripts/quick_gemm_bench.py 2>&1 | grep -v amdgpu.ids
  done
done
