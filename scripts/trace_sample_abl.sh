#!/bin/bash
# where the level-0 ("sample") kernel spends its time: kernel trace of the ablation build with the epilogue / tail cut off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export ISC_LIB=$GRAFT_REPO_ROOT/imagescry_amd/libimagescry_hip_ablation.so ISC_ALLOW_ABLATION=1
for abl in 0 1 2; do
  rm -rf gpurun_out/prof_abl
  ISC_SAMPLE_ABL=$abl rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_abl -- python3 scripts/quick_search_bench.py "$@" > gpurun_out/prof_abl.log 2>&1
  python3 - $abl <<'PY'
import csv, glob, sys, statistics
f = glob.glob("gpurun_out/prof_abl/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
d = {}
for r in rows:
    n = r["Kernel_Name"]
    if "k_dots_filter" in n and "Lb1E" in n:
        d.setdefault(r["Grid_Size_X"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"ISC_SAMPLE_ABL={sys.argv[1]}:", {g: round(statistics.median(v), 1) for g, v in d.items()})
PY
done
