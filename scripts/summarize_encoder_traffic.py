#!/usr/bin/env python3
"""gpurun_out/pmc_<which>_<tag>/ (scripts/pmc_encoder.sh) -> profiles/<tag>_<which>_traffic.json: L2 <-> fabric bytes per
k_conv_f32 launch and per predict_step of one encoder (512 x 224 x 224).  Units of 1024 B, FETCH_SIZE x 2 as calibrated on
this kernel's own access pattern (profiles/r02_resnet50_layer_traffic.txt: 1 x 1 layers that read every input byte exactly
once report 0.503 - 0.510 of those bytes; LDS-DMA and register loads alike move 16 B per lane).  `bench.py` copies
`hbm_bytes_per_launch` into the encoder's `roofline.traffic`; `algorithmic_bytes_per_step` is the layer table's inputs +
residuals + outputs + weights (imagescry_amd.<model>.conv_bytes)."""
from __future__ import annotations

import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
tag, which = sys.argv[1], sys.argv[2]
src = ROOT / "gpurun_out" / f"pmc_{which}_{tag}"
STEPS = 3  # predict_steps in scripts/trace_encode.py / trace_effnet.py


def total(sub: str, counter: str) -> tuple[float, int]:
    files = sorted(glob.glob(str(src / sub / "*/*counter_collection.csv")), key=lambda f: Path(f).stat().st_mtime, reverse=True)
    s, n = 0.0, 0
    for r in csv.DictReader(open(files[0])):
        if ("k_conv_f32" in r["Kernel_Name"] or "k_conv_halo_f32" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
            s += float(r["Counter_Value"])
            n += 1
    return s, n


fetch, n_f = total("fetch", "FETCH_SIZE")
write, n_w = total("write", "WRITE_SIZE")
assert n_f == n_w and n_f > 0 and n_f % STEPS == 0, (n_f, n_w)
if which == "resnet50":
    from imagescry_amd import resnet50 as model
    algo = float(model.conv_bytes(512, 224, 224))
else:
    from imagescry_amd import efficientnet as model
    algo = float(model.conv_bytes("s", 512, 224, 224))
per_step = (fetch * 1024 * 2 + write * 1024) / STEPS
out = {
    "tag": tag,
    "config": {"model": which, "batch_per_gpu": 512, "image": "224x224"},
    "kernel": "k_conv_f32",
    "launches_profiled": n_f,
    "launches_per_step": n_f // STEPS,
    "hbm_read_bytes_per_step": fetch * 1024 * 2 / STEPS,
    "hbm_write_bytes_per_step": write * 1024 / STEPS,
    "hbm_bytes_per_step": per_step,
    "hbm_bytes_per_launch": per_step / (n_f // STEPS),
    "algorithmic_bytes_per_step": algo,
    "traffic_over_algorithmic": round(per_step / algo, 3),
    "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over three warm predict_steps of "
              "the persistent LDS-DMA k_conv_f32 (the kernel bench.py times), x1024 B, FETCH_SIZE x2 (calibrated)",
}
(ROOT / "profiles" / f"{tag}_{which}_traffic.json").write_text(json.dumps(out, indent=1))
print(json.dumps(out, indent=1))
