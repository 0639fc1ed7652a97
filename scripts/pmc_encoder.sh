#!/bin/bash
# HBM (L2 <-> fabric) traffic of the k_conv_f32 launches of one encoder's predict_step (512 x 224 x 224, three warm steps):
# FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes (kernel trace only, as the pool requires).
#   bash scripts/pmc_encoder.sh <tag> resnet50|efficientnet_s     then on the host:
#   python scripts/summarize_encoder_traffic.py <tag> resnet50|efficientnet_s
set -o pipefail
TAG=${1:-r03}
WHICH=${2:-resnet50}
OUT=gpurun_out/pmc_${WHICH}_${TAG}
SCRIPT=scripts/trace_encode.py
[ "$WHICH" = "efficientnet_s" ] && SCRIPT=scripts/trace_effnet.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 $SCRIPT > "$OUT/fetch.log" 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 $SCRIPT > "$OUT/write.log" 2>&1
echo "pmc_encoder $WHICH exit $?"
