#!/usr/bin/env python3
"""profiles/<tag>_headline_trace.json from a rocprofv3 kernel trace of `python bench.py --no-sweep --no-secondary
--no-cpu-baseline` (scripts/trace_headline.sh).

The bench issues its searches in a known order -- W warm-up, K timed (event brackets off: `value`), K bracketed
(`roofline.kernel_ms_per_step`), 1 self-check (`roofline.search_calls_in_order` in the JSON line) -- and every search
starts with one `k_prep` launch, so the trace is cut into searches at `k_prep` and summarised per search: the
`k_dots_filter` time of every TIMED step, their mean, and the fraction of the fp16 MFMA peak that mean gives for the
algorithmic flops of the line.  Done = `frac_from_trace` reproduces the line's `roofline.frac` to 1 % and the
dominant-kernel time of a step is <= the line's `ms_per_step`.
"""
from __future__ import annotations

import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag, src = sys.argv[1], Path(sys.argv[2])
line = json.loads([l for l in (src / "bench.json").read_text().splitlines() if l.startswith("{")][-1])
order = line["roofline"]["search_calls_in_order"]
trace = sorted(glob.glob(str(src / "trace/*/*kernel_trace.csv")), key=lambda f: Path(f).stat().st_mtime)[-1]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
NAMES = ("k_prep", "k_dots_filter", "k_select", "k_final2", "k_final", "k_exact", "k_topk_merge")
searches: list[list[tuple[str, int, int]]] = []
for r in rows:
    name = next((n for n in NAMES if n in r["Kernel_Name"]), None)
    if name is None:
        continue
    if name == "k_prep":
        searches.append([])
    if searches:
        kind = name
        if name == "k_dots_filter":
            kn = r["Kernel_Name"]
            redo = any(f"Li{d}E" in kn or f", {d}," in kn for d in (20, 32, 33))
            kind = "k_dots_filter(redo, empty)" if redo else "k_dots_filter"
        searches[-1].append((kind, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
w, k = order["warmup"], order["timed"]
expect = w + 2 * k + order["selfcheck"]
timed = searches[w : w + k]
per_step = []
for sc in timed:
    filt = [e for e in sc if e[0] == "k_dots_filter"]
    per_step.append({
        "k_dots_filter_ms": round(sum(e[2] - e[1] for e in filt) / 1e6, 4),
        "k_dots_filter_launches": len(filt),
        "all_kernels_ms": round(sum(e[2] - e[1] for e in sc) / 1e6, 4),
        "span_ms": round((sc[-1][2] - sc[0][1]) / 1e6, 4),
    })
mean = sum(s["k_dots_filter_ms"] for s in per_step) / max(len(per_step), 1)
flops = line["roofline"]["algorithmic_flops_per_step"]
peak = line["roofline"]["peak"] if line["roofline"]["unit"] == "TFLOP/s" else 2500.0
frac_trace = flops / (mean / 1e3) / 1e12 / peak
out = {
    "tag": tag,
    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-sweep --no-secondary --no-cpu-baseline",
    "searches_in_trace": len(searches), "searches_expected": expect,
    "timed_steps": per_step,
    "k_dots_filter_ms_per_timed_step_mean": round(mean, 4),
    "frac_from_trace": round(frac_trace, 4),
    "line_of_the_same_run": {"value": line["value"], "ms_per_step": line["ms_per_step"],
                             "roofline_frac": line["roofline"]["frac"],
                             "kernel_ms_per_step": line["roofline"]["kernel_ms_per_step"],
                             "launches_per_step": line["roofline"]["launches_per_step"]},
    "frac_agrees_within_1pct": abs(frac_trace / line["roofline"]["frac"] - 1.0) <= 0.01,
    "dominant_kernel_ms_le_ms_per_step": mean <= line["ms_per_step"],
    "note": "profiled runs hold a 2 - 3 % lower clock than unprofiled ones (MI355X_MICROARCH.md, DVFS give-back item 2): "
            "compare this table with the line of THIS run, not with an unprofiled bench line",
}
# written next to the trace (gpurun_out/ travels back from the GPU box, profiles/ does not); copy them into profiles/ to commit
dst = src / "profiles"
dst.mkdir(exist_ok=True)
(dst / f"{tag}_headline_trace.json").write_text(json.dumps(out, indent=1))
(dst / f"{tag}_headline_trace_bench.json").write_text(json.dumps(line) + "\n")
stats = sorted(glob.glob(str(src / "trace/*/*kernel_stats.csv")), key=lambda f: Path(f).stat().st_mtime)
if stats:
    (dst / f"{tag}_headline_kernel_stats.csv").write_text(open(stats[-1]).read())
print(json.dumps({k_: out[k_] for k_ in ("searches_in_trace", "searches_expected", "k_dots_filter_ms_per_timed_step_mean",
                                         "frac_from_trace", "line_of_the_same_run", "frac_agrees_within_1pct",
                                         "dominant_kernel_ms_le_ms_per_step")}))
