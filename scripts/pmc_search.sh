#!/bin/bash
# PMC pass over the search-only bench: SQ wait / LDS counters for k_dots_filter (one pass, 8 SQ slots).
set -o pipefail
TAG=${1:-x}
shift
COUNTERS=${@:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES}
OUT=gpurun_out/pmc_${TAG}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$OUT"
rocprofv3 --kernel-trace --pmc $COUNTERS --output-format csv -d "$OUT" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/bench.json" 2> "$OUT/err.log"
echo "exit $?"; tail -2 "$OUT/err.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not f:
    print("no counter file"); sys.exit(0)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "k_dots_filter" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k:36s} max={max(v):.4g} n={len(v)}")
PY
