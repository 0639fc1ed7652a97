#!/bin/bash
# HBM/fabric traffic and matrix-pipe occupancy of the streaming GEMM on the ViT-B shapes (separate PMC passes, kernel trace only)
set -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/pmc_gemm_${TAG}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 scripts/quick_gemm_bench.py > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 scripts/quick_gemm_bench.py > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 scripts/quick_gemm_bench.py > "$OUT/sq.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
def load(sub):
    f = glob.glob(f"{out}/{sub}/*/*counter_collection.csv")
    rows = list(csv.DictReader(open(f[0]))) if f else []
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "k_gemm_f16" in r["Kernel_Name"]:
            key = (r["Kernel_Name"][:60], r["Grid_Size"])
            d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d
for sub in ("fetch", "write", "sq"):
    for key, cs in load(sub).items():
        print(sub, key, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
