#!/bin/bash
# per-layer time / TFLOP/s / GB/s of the EfficientNetV2-S encode step (rocprofv3 kernel trace of scripts/trace_effnet.py,
# the convolution / depthwise launches of the last step matched in order with the stage table)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
if [ -z "$PARSE_ONLY" ]; then
rm -rf gpurun_out/prof_eff_layers
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_eff_layers -- python3 scripts/trace_effnet.py > gpurun_out/prof_eff_layers.log 2>&1
fi
python3 - <<'PY'
import csv, glob, os, sys

def conv_launches(m, kk, cout, resident=512):
    """How many k_conv_f32 launches isc_conv2d_nhwc makes for a layer (encoder.hip conv_launch): 2 when whole rounds
    and a half-tile remainder are launched separately."""
    cdiv = lambda a, b: -(-a // b)
    ksteps = cdiv(kk, 32)
    if cout <= 32:
        return 1
    narrow = cdiv(cout, 64) * 64 < cdiv(cout, 128) * 128
    blocks = cdiv(cout, 64) * cdiv(m, 256) if narrow else cdiv(cout, 128) * cdiv(m, 128)
    rounds, rem = divmod(blocks, resident)
    if rem > 0 and rem * 4 <= resident * 3 and (ksteps >= 16 or rounds == 0):
        return 2 if rounds > 0 else 1
    return 1
sys.path.insert(0, ".")
from imagescry_amd import efficientnet as E
B = 512
layers = []  # (name, kind, pixels_out, K, N, bytes)
def down(n, s): return (n + s - 1) // s
h = down(224, 2)
layers.append(("stem 3x3/2", "conv", B * h * h, 27, 24, B * 224 * 224 * 16 + B * h * h * 24 * 4))
for si, blocks in enumerate(E.block_specs("s"), 1):
    for bi, b in enumerate(blocks):
        h2 = down(h, b.stride)
        pin, pout = B * h * h, B * h2 * h2
        n = f"s{si}.{bi}"
        res = pout * b.cout * 4 if b.residual else 0
        if b.kind == "fused":
            if b.expand == 1:
                layers.append((f"{n} fused 3x3", "conv", pout, 9 * b.cin, b.cout, (pin * b.cin + pout * b.cout) * 4 + res))
            else:
                layers.append((f"{n} fused 3x3/{b.stride}", "conv", pout, 9 * b.cin, b.expanded, (pin * b.cin + pout * b.expanded) * 4))
                layers.append((f"{n} project", "conv", pout, b.expanded, b.cout, (pout * b.expanded + pout * b.cout) * 4 + res))
        else:
            layers.append((f"{n} expand", "conv", pin, b.cin, b.expanded, (pin * b.cin + pin * b.expanded) * 4))
            if b.stride == 1 and h <= 14:  # two sweeps of the depthwise kernel, plain projection
                layers.append((f"{n} dw pool", "dw", pout, 9, b.expanded, pin * b.expanded * 4))
                layers.append((f"{n} se gate", "se", B, b.expanded, 2 * b.squeeze, 0))
                layers.append((f"{n} dw gated", "dw", pout, 9, b.expanded, (pin + pout) * b.expanded * 4))
                layers.append((f"{n} project", "conv", pout, b.expanded, b.cout, (pout * b.expanded + pout * b.cout) * 4 + res))
            else:
                layers.append((f"{n} dw 3x3/{b.stride}", "dw", pout, 9, b.expanded, (pin + pout) * b.expanded * 4))
                layers.append((f"{n} avgpool", "pool", pout, 1, b.expanded, pout * b.expanded * 4))
                layers.append((f"{n} se gate", "se", B, b.expanded, 2 * b.squeeze, 0))
                layers.append((f"{n} gated project", "conv", pout, b.expanded, b.cout, (pout * b.expanded + pout * b.cout) * 4 + res))
        h = h2
layers.append(("head 1x1", "conv", B * h * h, 256, 1280, B * h * h * (256 + 1280) * 4))
f = max(glob.glob("gpurun_out/prof_eff_layers/*/*kernel_trace.csv"), key=os.path.getmtime)
pat = {"conv": ("k_conv_f32", "k_conv_halo_f32", "k_conv1x1_f32_stream"), "dw": ("k_dwconv",), "pool": ("k_global_avgpool",), "se": ("k_se_gate",)}
rows = [r for r in csv.DictReader(open(f)) if any(p in r["Kernel_Name"] for ps in pat.values() for p in ps)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a convolution layer may be two launches (whole rounds + half-tile remainder); gated convolutions never split
need = [conv_launches(m, k, n) if kind == "conv" and "gated" not in name else 1 for (name, kind, m, k, n, by) in layers]
last = rows[-sum(need):]
merged, pos = [], 0
for cnt in need:
    grp = last[pos:pos + cnt]; pos += cnt
    r0 = dict(grp[0])
    r0["dur"] = sum(int(g["End_Timestamp"]) - int(g["Start_Timestamp"]) for g in grp)
    if cnt > 1:
        r0["Kernel_Name"] = grp[0]["Kernel_Name"] + " + halves"
    merged.append(r0)
last = merged
print(f"{'layer':24s} {'M':>9s} {'K':>5s} {'N':>5s} {'us':>8s} {'TFLOP/s':>8s} {'of peak':>7s} {'GB/s':>7s}  kernel")
agg = {}
tot = {"conv": [0, 0], "dw": [0, 0], "pool": [0, 0], "se": [0, 0]}
for (name, kind, m, k, n, by), r in zip(layers, last):
    kn = r["Kernel_Name"]
    assert any(p in kn for p in pat[kind]), (name, kn)
    us = r["dur"] / 1e3
    fl = 2.0 * m * k * n
    tot[kind][0] += us; tot[kind][1] += fl
    short = kn.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40] + (" + halves" if kn.endswith("+ halves") else "")
    print(f"{name:24s} {m:9d} {k:5d} {n:5d} {us:8.1f} {fl/us/1e6:8.1f} {fl/us/1e6/157.3:7.2f} {by/us/1e3:7.0f}  {short}")
    key = name.split(" ", 1)[0][:2] + " " + name.split(" ", 1)[1].split("/")[0]
    a = agg.setdefault(key, [0, 0, 0]); a[0] += us; a[1] += fl; a[2] += 1
for kind, (us, fl) in tot.items():
    print(f"total {kind:5s} {us:8.0f} us" + (f", {fl/us/1e6:.1f} TFLOP/s = {fl/us/1e6/157.3:.3f} of the f32 MFMA peak" if kind == "conv" else ""))
for k2, (us, fl, cnt) in agg.items():
    print(f"  {k2:22s} x{cnt:<3d} {us:8.0f} us  {fl/us/1e6:6.1f} TFLOP/s")
PY
