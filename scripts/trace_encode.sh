#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_enc
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_enc -- python3 scripts/trace_encode.py > gpurun_out/prof_enc.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_enc/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "(anonymous namespace)" in r["Kernel_Name"] or "_GLOBAL__N" in r["Kernel_Name"]]
n = len(rows) // 3
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
tot = 0
for r in last:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{name[:34]:34s} grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):7d} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} dur {d:8.1f} us")
print("busy", tot, "span", (int(last[-1]["End_Timestamp"]) - t0) / 1e3)
PY
