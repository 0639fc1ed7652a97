#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (scripts/profile_gpu.sh) into profiles/<tag>_summary.json.

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are reported in units of 1024 B; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide (16 B / lane) coalesced streaming read, so the read side
is doubled for the streaming kernels; WRITE_SIZE is exact.
"""
from __future__ import annotations

import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag = sys.argv[1]
src = ROOT / "gpurun_out" / f"prof_{tag}"
out = ROOT / "profiles" / f"{tag}_summary.json"
KERNELS = ("k_dots_filter", "k_conv_f32", "k_gemm_f16", "k_attention_f16", "k_layernorm", "k_select", "k_rescore",
           "k_topk_merge", "k_maxpool")


def short(name: str) -> str | None:
    for k in KERNELS:
        if k in name:
            if k == "k_conv_f32":
                return "k_conv_f32<64,256>" if ("64, 256" in name or "Li64ELi256" in name) else "k_conv_f32<128,128>"
            if k == "k_dots_filter":  # the two tile shapes are different kernels
                return "k_dots_filter<256>" if ("Li256E" in name or ", 256," in name) else "k_dots_filter<64>"
            return k
    return None


summary: dict = {"tag": tag, "kernel_stats": {}, "pmc_per_launch": {}}
stats = sorted(glob.glob(str(src / "stats/*/*kernel_stats.csv")), key=lambda f: Path(f).stat().st_mtime, reverse=True)
if stats:
    for r in csv.DictReader(open(stats[0])):
        k = short(r["Name"])
        if k:  # template instantiations that share a short name are summed
            e = summary["kernel_stats"].setdefault(k, {"calls": 0, "total_ms": 0.0, "pct": 0.0, "max_us": 0.0})
            e["calls"] += int(r["Calls"])
            e["total_ms"] = round(e["total_ms"] + float(r["TotalDurationNs"]) / 1e6, 3)
            e["pct"] = round(e["pct"] + float(r["Percentage"]), 4)
            e["max_us"] = max(e["max_us"], round(float(r["MaxNs"]) / 1e3, 2))
            e["avg_us"] = round(e["total_ms"] * 1e3 / e["calls"], 2)
    (ROOT / "profiles" / f"{tag}_kernel_stats.csv").write_text(open(stats[0]).read())
for sub in ("pmc_fetch", "pmc_write", "pmc_mfma"):
    files = sorted(glob.glob(str(src / sub / "*/*counter_collection.csv")), key=lambda f: Path(f).stat().st_mtime, reverse=True)
    if not files:
        continue
    agg: dict = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        if k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, counters in agg.items():
        for c, vals in counters.items():
            entry = summary["pmc_per_launch"].setdefault(k, {})
            entry[c] = {"launches": len(vals), "max": max(vals), "mean": sum(vals) / len(vals)}
# derived numbers for the search kernels: the largest launch of a step is the level that streams the bank.
# Calibration of FETCH_SIZE (MI355X_MICROARCH.md "HBM"): the 64-query-tile launch reads its 9.74 M bank rows exactly
# once (14.96 GB) and reports 7.49e6 units of 1024 B, i.e. half the bytes -- the guide's factor 2 holds for this
# kernel's 16 B / lane LDS-DMA stream and is applied below.
for variant in ("k_dots_filter<256>", "k_dots_filter<64>"):
    d = summary["pmc_per_launch"].get(variant, {})
    if "FETCH_SIZE" not in d:
        continue
    big = {
        "hbm_read_bytes_corrected_max": d["FETCH_SIZE"]["max"] * 1024 * 2,
        "hbm_read_bytes_corrected_mean_per_launch": d["FETCH_SIZE"]["mean"] * 1024 * 2,
        "hbm_write_bytes_max": d.get("WRITE_SIZE", {}).get("max", 0) * 1024,
    }
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        busy = d["SQ_VALU_MFMA_BUSY_CYCLES"]["max"] / 1024.0  # per SIMD (256 CUs x 4)
        active = d["GRBM_GUI_ACTIVE"]["max"] / 8.0  # per XCD
        big["mfma_busy_frac_largest_launch"] = round(busy / active, 4)
    summary[variant + "_derived"] = big
out.write_text(json.dumps(summary, indent=1))
print(json.dumps({k: v for k, v in summary.items() if k.endswith("_derived") or k == "kernel_stats"}, indent=1))
