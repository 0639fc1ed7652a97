#!/bin/bash
# Per-layer FETCH_SIZE / WRITE_SIZE of the ResNet-50 encode step against the layer table's known bytes: calibrates the
# gfx950 FETCH_SIZE correction for k_conv_f32's access pattern (16-byte register loads, 8 lanes per 128-byte pixel row).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/pmc_encode_layers
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 scripts/trace_encode.py > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 scripts/trace_encode.py > $OUT/write.log 2>&1
python3 - <<'PY'
import csv, glob, json
B = 512
layers = []  # (name, in_bytes, weight_bytes, out_bytes, res_bytes, cout_tiles)
def out(n, k, s, p): return (n + 2 * p - k) // s + 1
h = out(224, 7, 2, 3)
layers.append(("stem 7x7/2", B * 224 * 224 * 16, 64 * 56 * 16, B * h * h * 64 * 4, 0, 1))
h = out(h, 3, 2, 1)
inpl = 64
for li, (planes, nb, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
    for bi in range(nb):
        s = stride if bi == 0 else 1
        h2 = out(h, 3, s, 1)
        pin, pout = B * h * h, B * h2 * h2
        ct = lambda c: 1 if c <= 64 else (c + 127) // 128
        if bi == 0:
            layers.append((f"l{li}.{bi}.down", pin * inpl * 4 // (s * s) if s > 1 else pin * inpl * 4, planes * 4 * inpl * 4, pout * planes * 16, 0, ct(planes * 4)))
        layers.append((f"l{li}.{bi}.conv1", pin * inpl * 4, planes * inpl * 4, pin * planes * 4, 0, ct(planes)))
        layers.append((f"l{li}.{bi}.conv2", pin * planes * 4, planes * planes * 36, pout * planes * 4, 0, ct(planes)))
        layers.append((f"l{li}.{bi}.conv3", pout * planes * 4, planes * 4 * planes * 4, pout * planes * 16, pout * planes * 16, ct(planes * 4)))
        inpl = planes * 4
        h = h2
layers.append(("fc", B * 2048 * 4, 768 * 2048 * 4, B * 768 * 4, 0, 6))
def per_launch(sub, counter):
    f = glob.glob(f"gpurun_out/pmc_encode_layers/{sub}/*/*counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if ("k_conv_f32" in r["Kernel_Name"] or "k_conv1x1_f32_stream" in r["Kernel_Name"]) and r["Counter_Name"] == counter]
    return [float(r["Counter_Value"]) * 1024 for r in rows][-len(layers):]
fe, wr = per_launch("fetch", "FETCH_SIZE"), per_launch("write", "WRITE_SIZE")
print(f"{'layer':14s} {'in MB':>8s} {'res MB':>8s} {'out MB':>8s} {'FETCH MB':>9s} {'FETCH/(in+res)':>14s} {'WRITE/out':>9s} cout tiles")
tot = {"in": 0, "res": 0, "out": 0, "fetch": 0, "write": 0}
single = []
for (name, ib, wb, ob, rb, ct), f, w in zip(layers, fe, wr):
    print(f"{name:14s} {ib/1e6:8.1f} {rb/1e6:8.1f} {ob/1e6:8.1f} {f/1e6:9.1f} {f/(ib+rb):14.3f} {w/ob:9.3f} {ct}")
    for k2, v in (("in", ib), ("res", rb), ("out", ob), ("fetch", f), ("write", w)): tot[k2] += v
    if ct == 1 and rb == 0 and "conv1" in name: single.append(f / ib)
print("calibration layers (1x1, one output-channel tile, no residual: every input byte is read exactly once): FETCH/in =", [round(x, 3) for x in single])
print("step totals: in+res %.2f GB, out %.2f GB, FETCH %.2f GB (raw), WRITE %.2f GB" % ((tot["in"] + tot["res"]) / 1e9, tot["out"] / 1e9, tot["fetch"] / 1e9, tot["write"] / 1e9))
json.dump({"layers": [l[0] for l in layers], "fetch_raw_bytes": fe, "write_bytes": wr, "calibration_fetch_over_input": single, "totals": tot}, open("gpurun_out/pmc_encode_layers/summary.json", "w"))
PY
