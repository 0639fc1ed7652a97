#!/usr/bin/env python3
"""gpurun_out/pmc_headline_<tag>/ (scripts/pmc_headline.sh) -> profiles/<tag>_headline_traffic.json.

HBM bytes per `k_dots_filter` launch of the headline step, as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE and
WRITE_SIZE come from separate passes and are in units of 1024 B; on gfx950 FETCH_SIZE reports half of the bytes of
this kernel's 16 B / lane stream (calibrated in profiles/r01_summary.json: the single-pass 64-query launch reads its
bank rows exactly once), so the read side is doubled; WRITE_SIZE is exact.  bench.py copies `hbm_bytes_per_launch`
into `roofline.traffic` when the workload matches.
"""
from __future__ import annotations

import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
src = ROOT / "gpurun_out" / f"pmc_headline_{tag}"


def total(sub: str, counter: str) -> tuple[float, int, int]:
    """(sum of the counter over every k_dots_filter launch, those launches, searches profiled).  A search is one k_prep
    launch; its k_dots_filter launches include the redo launches behind k_final, which normally exit at once and move no
    bytes -- so the bytes are divided by SEARCHES and by the bench line's `launches_per_step` (the launches that stream)."""
    files = sorted(glob.glob(str(src / sub / "*/*counter_collection.csv")), key=lambda f: Path(f).stat().st_mtime, reverse=True)
    s, n, searches = 0.0, 0, 0
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != counter:
            continue
        if "k_dots_filter" in r["Kernel_Name"]:
            s += float(r["Counter_Value"])
            n += 1
        elif "k_prep" in r["Kernel_Name"]:
            searches += 1
    return s, n, searches


bench = json.loads((src / "bench_fetch.json").read_text().strip().splitlines()[-1])
fetch, n_f, s_f = total("fetch", "FETCH_SIZE")
write, n_w, s_w = total("write", "WRITE_SIZE")
assert n_f == n_w and n_f > 0 and s_f == s_w and s_f > 0
lps = bench["roofline"]["launches_per_step"]
read_b = fetch * 1024 * 2 / s_f / lps
write_b = write * 1024 / s_w / lps
algo = bench["roofline"]["algorithmic_bytes_per_step"] / lps
out = {
    "tag": tag,
    "config": bench["config"],
    "kernel": "k_dots_filter",
    "launches_profiled": n_f,
    "searches_profiled": s_f,
    "hbm_read_bytes_per_launch": read_b,
    "hbm_write_bytes_per_launch": write_b,
    "hbm_bytes_per_launch": read_b + write_b,
    "algorithmic_bytes_per_launch": algo,
    "launches_per_step": lps,
    "hbm_bytes_per_step": (read_b + write_b) * lps,
    "algorithmic_bytes_per_step": algo * lps,
    "ratio_to_algorithmic": (read_b + write_b) / algo,
    "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), x1024 B, FETCH_SIZE x2 (gfx950)",
}
(ROOT / "profiles" / f"{tag}_headline_traffic.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out, indent=1))
