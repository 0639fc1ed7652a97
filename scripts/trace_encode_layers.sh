#!/bin/bash
# per-layer time / TFLOP/s / GB/s of the ResNet-50 encode step (rocprofv3 kernel trace of scripts/trace_encode.py,
# k_conv_f32 launches matched in order with the layer table)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
if [ -z "$PARSE_ONLY" ]; then
rm -rf gpurun_out/prof_enc
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_enc -- python3 scripts/trace_encode.py > gpurun_out/prof_enc.log 2>&1
fi
python3 - <<'PY'
import csv, glob

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
B = 512
FUSED_SHORTCUT_STAGES = (1, 2, 3, 4)
layers = []  # (name, pixels_out, cin*k*k, cout, in_bytes, out_bytes, res_bytes)
def out(n, k, s, p): return (n + 2 * p - k) // s + 1
h = w = out(224, 7, 2, 3)
layers.append(("stem 7x7/2", B * h * w, 49 * 4, 64, B * 224 * 224 * 16, B * h * w * 64 * 4, 0))
h = w = out(h, 3, 2, 1)
inpl = 64
for li, (planes, nb, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
    for bi in range(nb):
        s = stride if bi == 0 else 1
        h2 = out(h, 3, s, 1)
        px_in, px_out = B * h * h, B * h2 * h2
        fused = bi == 0 and li in FUSED_SHORTCUT_STAGES  # resnet50.FUSED_SHORTCUT_STAGES: the shortcut runs inside conv3
        if bi == 0 and not fused:
            layers.append((f"l{li}.{bi}.down 1x1/{s}", px_out, inpl, planes * 4, px_in * inpl * 4, px_out * planes * 16, 0))
        layers.append((f"l{li}.{bi}.conv1 1x1", px_in, inpl, planes, px_in * inpl * 4, px_in * planes * 4, 0))
        layers.append((f"l{li}.{bi}.conv2 3x3/{s}", px_out, planes * 9, planes, px_in * planes * 4, px_out * planes * 4, 0))
        if fused:
            layers.append((f"l{li}.{bi}.conv3+shortcut", px_out, planes + inpl, planes * 4,
                           px_out * (planes + inpl) * 4, px_out * planes * 16, 0))
        else:
            layers.append((f"l{li}.{bi}.conv3 1x1+res", px_out, planes, planes * 4, px_out * planes * 4, px_out * planes * 16, px_out * planes * 16))
        inpl = planes * 4
        h = h2
# (the projection head is no longer a k_conv_f32 launch: k_pool_linear_l2norm, reported below)
import os
f = max(glob.glob("gpurun_out/prof_enc/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "k_conv_f32" in r["Kernel_Name"] or "k_conv_halo_f32" in r["Kernel_Name"] or "k_conv1x1_f32_stream" in r["Kernel_Name"]]
need = [1 if "shortcut" in name else conv_launches(m, k, nn) for (name, m, k, nn, ib, ob, rb) in layers]
last = rows[-sum(need):]
merged, pos = [], 0
for cnt in need:
    grp = last[pos:pos + cnt]; pos += cnt
    r0 = dict(grp[0])
    r0["dur"] = sum(int(g["End_Timestamp"]) - int(g["Start_Timestamp"]) for g in grp)
    if cnt > 1:
        r0["Kernel_Name"] = grp[0]["Kernel_Name"].replace(">(", "> + halves(", 1)
    merged.append(r0)
last = merged
tot_t = tot_f = 0
print(f"{'layer':22s} {'M':>9s} {'K':>5s} {'N':>5s} {'us':>8s} {'TFLOP/s':>8s} {'of peak':>7s} {'GB/s':>7s}  tile")
agg = {}
for (name, m, k, nn, ib, ob, rb), r in zip(layers, last):
    us = r["dur"] / 1e3
    fl = 2.0 * m * k * nn
    tot_t += us; tot_f += fl
    kn = r["Kernel_Name"]
    tile = "ring 128, 256" if "ring" in kn else ("halo MI=" if "halo" in kn else "stream " if "stream" in kn else "") + kn.split("<")[1].split(">")[0]
    print(f"{name:22s} {m:9d} {k:5d} {nn:5d} {us:8.1f} {fl/us/1e6:8.1f} {fl/us/1e6/157.3:7.2f} {(ib+ob+rb)/us/1e3:7.0f}  {tile}")
    kind = name.split(".")[-1].split(" ")[0] if name.startswith("l") else name.split(" ")[0]
    a = agg.setdefault(name[:2] + " " + kind, [0, 0]); a[0] += us; a[1] += fl
print(f"total conv {tot_t:.0f} us, {tot_f/tot_t/1e6:.1f} TFLOP/s = {tot_f/tot_t/1e6/157.3:.3f} of the f32 MFMA peak")
for k2, (us, fl) in agg.items():
    print(f"  {k2:12s} {us:8.0f} us  {fl/us/1e6:6.1f} TFLOP/s")
others = {}
for r in csv.DictReader(open(f)):
    for kn in ("k_pool_linear_l2norm", "k_maxpool_nhwc", "k_normalize_nhwc4", "k_stats_partial", "k_stats_final"):
        if kn in r["Kernel_Name"]:
            others.setdefault(kn, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for kn, v in others.items():
    print(f"  {kn:24s} {sum(v)/len(v):8.1f} us per launch ({len(v)} launches in the trace)")
PY
