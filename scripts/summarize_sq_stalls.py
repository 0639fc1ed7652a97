#!/usr/bin/env python3
"""gpurun_out/pmc_stalls_<tag>/ (scripts/pmc_stalls.sh) -> profiles/<tag>_sq_stalls.json: where the waves of the search's
k_dots_filter launches spend their cycles, as shares of SQ_WAVE_CYCLES (issuing = SQ_ACTIVE_INST_ANY, issue-stalled =
SQ_WAIT_INST_ANY, parked on a wait or the barrier = SQ_WAIT_ANY; MI355X_MICROARCH.md "rocprofv3 PMC slots").  The redo
launches behind k_final exit at once (a few hundred wave cycles) and are left out."""
from __future__ import annotations

import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
src = ROOT / "gpurun_out" / f"pmc_stalls_{tag}"
f = sorted(glob.glob(str(src / "sq/*/*counter_collection.csv")))[-1]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "k_dots_filter" in r["Kernel_Name"]:
        per[(r["Dispatch_Id"], "sample" if "Lb1E" in r["Kernel_Name"] else "filter")][r["Counter_Name"]] = float(r["Counter_Value"])
groups = collections.defaultdict(list)
for (_, kind), c in per.items():
    if c.get("SQ_WAVE_CYCLES", 0) < 1e6:  # an empty (early-exit) redo launch
        continue
    groups[kind].append(c)
out = {"tag": tag, "method": "scripts/pmc_stalls.sh: rocprofv3 --kernel-trace --pmc SQ_* (one pass) over bench.py --steps 3 "
                             "(10M x 768 fp16, Q = 1024); shares of SQ_WAVE_CYCLES", "kernels": {}}
for kind, rows in groups.items():
    tot = lambda k: sum(r.get(k, 0.0) for r in rows)
    wc = tot("SQ_WAVE_CYCLES")
    out["kernels"][kind] = {
        "launches": len(rows),
        "issuing": round(tot("SQ_ACTIVE_INST_ANY") / wc, 3),
        "issue_stalled": round(tot("SQ_WAIT_INST_ANY") / wc, 3),
        "parked": round(tot("SQ_WAIT_ANY") / wc, 3),
        "lds_issue_stall": round(tot("SQ_WAIT_INST_LDS") / wc, 3),
        "mfma_busy_cycles": tot("SQ_VALU_MFMA_BUSY_CYCLES") / len(rows),
        "sq_busy_cycles": tot("SQ_BUSY_CYCLES") / len(rows),
    }
(ROOT / "profiles" / f"{tag}_sq_stalls.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out, indent=1))
