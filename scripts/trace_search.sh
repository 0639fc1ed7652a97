#!/bin/bash
# kernel-by-kernel durations of one search call (rocprofv3 kernel trace of scripts/quick_search_bench.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_trace -- python3 scripts/quick_search_bench.py "$@" > gpurun_out/prof_trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_trace/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
names = ["k_dots_filter", "k_select", "k_final2", "k_final", "k_prep", "k_exact"]
seq = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Grid_Size_X"]) for r in rows if any(n in r["Kernel_Name"] for n in names)]
def short(n):
    for k in names:
        if k in n: return k + ("<64>" if "Li64E" in n else "") + (" sample" if "Lb1E" in n and "k_dots_filter" in n else "")
# split into search calls at k_init
calls, cur = [], []
for e in seq:
    if "k_prep" in e[0] and cur:
        calls.append(cur); cur = []
    cur.append(e)
calls.append(cur)
shown = set()
for c in calls[::-1]:
    key = (len(c), c[-1][3], "Li64E" in c[1][0] if len(c) > 2 else 0, c[2][3] if len(c) > 2 else 0)
    if key in shown: continue
    shown.add(key)
    t0 = c[0][1]
    print(f"--- search call: {len(c)} kernels, span {(c[-1][2]-t0)/1e3:.1f} us, busy {sum(e[2]-e[1] for e in c)/1e3:.1f} us")
    for name, st, en, grid in c:
        print(f"  {short(name):22s} start {(st-t0)/1e3:9.1f}  dur {(en-st)/1e3:9.1f} us  grid={grid}")
PY
