"""Randomised differential tests of the non-search kernels against torch CPU / the oracle, on the GPU.

    python scripts/fuzz_kernels.py [seconds per family] [seed]

Families: normalize_per_channel, resize, conv2d (NHWC implicit GEMM, all epilogues), fp16 GEMM (both layouts, both
kernels), attention, layernorm.  Prints every failing configuration; exit status 1 if there was one.
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from imagescry_amd import _lib, normalize_per_channel, resize
from imagescry_amd.embedding import _conv
from imagescry_amd.resnet50 import FoldedConv
from imagescry_amd.vit import pack_rows, packed_elems, unpack_rows
from oracle import transforms_oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
lib = _lib.load()
fails = 0


def report(family, ok, desc):
    global fails
    if not ok:
        fails += 1
        print(f"FAIL {family}: {desc}", flush=True)


def gen():
    return torch.Generator().manual_seed(int(rng.integers(1 << 31)))


def run_family(name, fn):
    t_end, n = time.time() + budget, 0
    while time.time() < t_end:
        fn()
        n += 1
    print(f"{name}: {n} cases", flush=True)


def f_normalize():
    b, c = int(rng.integers(1, 9)), int(rng.integers(1, 5))
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    if b * h * w < 2:
        w += 1
    g = gen()
    x = torch.randint(0, 256, (b, c, h, w), dtype=torch.uint8, generator=g)
    if rng.random() < 0.3:
        x = x.float() * 0.37 - 11.0
    kw = {}
    if rng.random() < 0.6:
        kw = {"min_value": -3.0, "max_value": 3.0}
    exp = transforms_oracle.normalize_per_channel(x, **kw)
    got = normalize_per_channel(x.to(dev), **kw).cpu()
    report("normalize", torch.allclose(got, exp, rtol=0, atol=4e-6), f"{tuple(x.shape)} {x.dtype} {kw} max err {(got - exp).abs().max():.3g}")


def f_resize():
    b, c = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    h, w = int(rng.integers(1, 90)), int(rng.integers(1, 90))
    g = gen()
    x = torch.randint(0, 256, (b, c, h, w), dtype=torch.uint8, generator=g)
    if rng.random() < 0.3:
        x = x.float() / 3
    if rng.random() < 0.5:
        size = (int(rng.integers(1, 120)), int(rng.integers(1, 120)))
        kw = {}
    else:
        size = int(rng.integers(1, 150))
        kw = {"side_ref": str(rng.choice(["long", "short", "height", "width"]))}
    try:
        exp = transforms_oracle.resize(x, size, **kw)
    except Exception as e:  # a degenerate output size: the product must refuse it too
        try:
            resize(x.to(dev), size, **kw)
            report("resize", False, f"{tuple(x.shape)} -> {size} {kw}: oracle raised {type(e).__name__}, product did not")
        except Exception:
            pass
        return
    got = resize(x.to(dev), size, **kw).cpu()
    ok = got.shape == exp.shape and torch.allclose(got, exp, rtol=2e-6, atol=2e-5)
    report("resize", ok, f"{tuple(x.shape)} {x.dtype} -> {size} {kw}: shapes {tuple(got.shape)} / {tuple(exp.shape)}")


def f_conv():
    b = int(rng.integers(1, 4))
    h, w = int(rng.integers(1, 20)), int(rng.integers(1, 20))
    cin = int(rng.choice([4, 8, 24, 32, 48, 64, 80, 96, 160, 256]))  # Cin % 32 != 0: packed-K mode
    cout = int(rng.choice([4, 24, 32, 48, 64, 100, 128, 132, 160, 192, 256, 320]))
    k = int(rng.choice([1, 1, 3, 3, 5]))
    if rng.random() < 0.2:  # more tiles than resident workgroups: persistent walk, whole rounds + half-tile remainder
        b, h, w = int(rng.integers(2, 9)), int(rng.integers(60, 180)), int(rng.integers(60, 180))
        cin = int(rng.choice([8, 24, 32, 64]))
        k = int(rng.choice([1, 3]))
    stride = int(rng.choice([1, 1, 2]))
    pad = int(rng.choice([0, k // 2]))
    if h + 2 * pad < k or w + 2 * pad < k:
        return
    g = gen()
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    act_name = str(rng.choice(["none", "relu", "silu", "sigmoid", "gelu"]))
    res_mode = str(rng.choice(["no", "before", "after"]))
    y = F.conv2d(x, wt, bias, stride=stride, padding=pad)
    res = torch.randn(y.shape, generator=g) if res_mode != "no" else None
    fn = {"none": lambda t: t, "relu": F.relu, "silu": F.silu, "sigmoid": torch.sigmoid, "gelu": F.gelu}[act_name]
    exp = fn(y + res) if res_mode == "before" else fn(y) + (res if res is not None else 0)
    act = {"none": 0, "relu": 1, "gelu": 2, "silu": 3, "sigmoid": 4}[act_name] | (_lib.ISC_ACT_RESIDUAL_AFTER if res_mode == "after" else 0)
    wk = wt.permute(0, 2, 3, 1).reshape(cout, k * k * cin)
    if cin % 32:  # packed-K rows: zero-padded to whole 32-float K steps
        wk = F.pad(wk, (0, (-wk.shape[1]) % 32))
    conv = FoldedConv(wk.contiguous().to(dev), bias.to(dev), k, stride, pad)
    rn = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(dev)
    got = _conv(x.permute(0, 2, 3, 1).contiguous().to(dev), conv, act, residual=rn).permute(0, 3, 1, 2).cpu()
    err = float((got - exp).abs().max() / exp.abs().max().clamp_min(1e-30))
    report("conv", got.shape == exp.shape and err < 2e-5,
           f"b{b} {h}x{w} cin{cin} cout{cout} k{k} s{stride} p{pad} {act_name} res={res_mode}: rel err {err:.3g}")


def f_conv_dual():
    """isc_conv2d_nhwc_dual: a bottleneck's conv3 with its projection shortcut K-concatenated (any stride of the shortcut)."""
    from imagescry_amd.embedding import _conv_dual
    b = int(rng.integers(1, 5))
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    if rng.random() < 0.2:  # more tiles than resident workgroups
        b, h, w = int(rng.integers(2, 7)), int(rng.integers(60, 150)), int(rng.integers(60, 150))
    cin, cin2 = int(rng.choice([32, 64, 96, 128, 256])), int(rng.choice([32, 64, 160, 256, 512]))
    cout = int(rng.choice([4, 36, 64, 100, 128, 192, 256, 512]))
    s2 = int(rng.choice([1, 1, 2, 3]))
    h2, w2 = (h - 1) * s2 + 1 + int(rng.integers(0, s2)), (w - 1) * s2 + 1 + int(rng.integers(0, s2))
    g = gen()
    t = torch.randn(b, cin, h, w, generator=g)
    x = torch.randn(b, cin2, h2, w2, generator=g)
    w3 = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    wd = torch.randn(cout, cin2, 1, 1, generator=g) / cin2 ** 0.5
    b3, bd = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    act_name = str(rng.choice(["none", "relu", "silu"]))
    fn = {"none": lambda z: z, "relu": F.relu, "silu": F.silu}[act_name]
    exp = fn(F.conv2d(t, w3, b3) + F.conv2d(x, wd, bd, stride=s2))
    fused = FoldedConv(torch.cat([w3.reshape(cout, cin), wd.reshape(cout, cin2)], dim=1).contiguous().to(dev),
                       (b3 + bd).to(dev), 1, s2, 0)
    got = _conv_dual(t.permute(0, 2, 3, 1).contiguous().to(dev), x.permute(0, 2, 3, 1).contiguous().to(dev), fused,
                     {"none": 0, "relu": 1, "silu": 3}[act_name]).permute(0, 3, 1, 2).cpu()
    err = float((got - exp).abs().max() / exp.abs().max().clamp_min(1e-30))
    report("conv_dual", got.shape == exp.shape and err < 2e-5,
           f"b{b} {h}x{w} cin{cin}+{cin2} ({h2}x{w2}/{s2}) cout{cout} {act_name}: rel err {err:.3g}")


def f_gemm():
    m = int(rng.choice([1, 7, 128, 129, 255, 300, 513, 1100]))
    k = int(rng.choice([64, 128, 192, 320, 768]))
    n = int(rng.choice([4, 64, 68, 128, 192, 256, 260, 768]))
    pk = bool(rng.random() < 0.5)
    out_f32 = bool(rng.random() < 0.5)
    tile256 = bool(rng.random() < 0.3) and k >= 192
    gelu = bool(rng.random() < 0.4) and not tile256
    res = bool(rng.random() < 0.5)
    if pk and not out_f32 and n % 64:
        out_f32 = True
    g = gen()
    a = torch.randn(m, k, generator=g).half()
    w = (torch.randn(n, k, generator=g) * 0.05).half()
    bias = torch.randn(n, generator=g)
    r = torch.randn(m, n, generator=g) if res else None
    want = a.double() @ w.double().T + bias.double()
    if gelu:
        want = F.gelu(want)
    if res:
        want = want + r.double()
    ad, wd, bd = (pack_rows(a) if pk else a).to(dev), (pack_rows(w) if pk else w).to(dev), bias.to(dev)
    rd = r.to(dev) if res else None
    flags = (3 | (0 if out_f32 else 4)) if pk else 0
    flags |= 8 if tile256 else 0
    out = torch.full((packed_elems(m, n) if pk and not out_f32 else m * n,), float("nan"),
                     dtype=torch.float32 if out_f32 else torch.float16, device=dev)
    st = lib.isc_gemm_f16(ad.data_ptr(), m, k, wd.data_ptr(), n, bd.data_ptr(), _lib.ptr(rd), 2 if gelu else 0,
                          out.data_ptr(), _lib.ISC_F32 if out_f32 else _lib.ISC_F16, flags, _lib.stream_handle(dev))
    _lib.check(st, "isc_gemm_f16")
    got = (unpack_rows(out.cpu(), m, n) if pk and not out_f32 else out.cpu().view(m, n)).double()
    tol = dict(rtol=1e-4, atol=1e-4) if out_f32 else dict(rtol=2e-3, atol=1e-3)
    report("gemm", torch.allclose(got, want, **tol),
           f"m{m} k{k} n{n} packed={pk} f32={out_f32} tile256={tile256} gelu={gelu} res={res}: max err {(got - want).abs().max():.3g}")


def f_attention():
    b, t, heads = int(rng.integers(1, 4)), int(rng.choice([1, 2, 15, 16, 17, 50, 191, 192, 193, 197, 207, 208, 209, 223, 224])), int(rng.integers(1, 5))
    if rng.random() < 0.3:  # more (image, head) pairs than CUs: the persistent sixteen-wave form, one to three heads per workgroup
        heads = int(rng.choice([1, 3, 12]))
        b = int(rng.integers(257, 700)) // heads + 1
        t = int(rng.choice([1, 16, 17, 50, 193, 197, 208, 209, 224]))
    pk = bool(rng.random() < 0.5)
    d = heads * 64
    g = gen()
    qkv = (torch.randn(b, t, 3 * d, generator=g) * 1.5).half()
    q, k, v = (z.reshape(b, t, heads, 64).transpose(1, 2).double() for z in qkv.split(d, dim=-1))
    want = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) @ v).transpose(1, 2).reshape(b, t, d)
    qd = (pack_rows(qkv.reshape(b * t, 3 * d)) if pk else qkv).to(dev)
    out = torch.full((packed_elems(b * t, d) if pk else b * t * d,), float("nan"), dtype=torch.float16, device=dev)
    _lib.check(lib.isc_attention_f16(qd.data_ptr(), b, t, heads, 64, out.data_ptr(), int(pk), _lib.stream_handle(dev)), "att")
    got = (unpack_rows(out, b * t, d) if pk else out).view(b, t, d).double().cpu()
    report("attention", torch.allclose(got, want, rtol=4e-3, atol=2e-3), f"b{b} t{t} heads{heads} packed={pk}: max err {(got - want).abs().max():.3g}")


def f_layernorm():
    rows, d = int(rng.integers(1, 600)), int(rng.choice([4, 64, 192, 768, 1024, 2048]))
    pk = bool(rng.random() < 0.5) and d % 64 == 0
    f32 = bool(rng.random() < 0.5) and not pk
    g = gen()
    x = torch.randn(rows, d, generator=g) * 3 + 1.5
    gamma, beta = torch.rand(d, generator=g) + 0.5, torch.randn(d, generator=g)
    want = F.layer_norm(x, (d,), gamma, beta, 1e-6)
    xd, gd, bd = x.to(dev), gamma.to(dev), beta.to(dev)
    out = torch.empty(packed_elems(rows, d) if pk else rows * d, dtype=torch.float32 if f32 else torch.float16, device=dev)
    st = lib.isc_layernorm(xd.data_ptr(), rows, d, d, gd.data_ptr(), bd.data_ptr(), 1e-6, out.data_ptr(),
                           _lib.ISC_F32 if f32 else _lib.ISC_F16, d, int(pk), _lib.stream_handle(dev))
    _lib.check(st, "isc_layernorm")
    got = (unpack_rows(out, rows, d) if pk else out.view(rows, d)).float().cpu()
    tol = dict(rtol=1e-5, atol=1e-5) if f32 else dict(rtol=1e-3, atol=2e-3)
    report("layernorm", torch.allclose(got, want, **tol), f"rows{rows} d{d} packed={pk} f32={f32}: max err {(got - want).abs().max():.3g}")


def f_dwconv():
    b, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 24)), int(rng.integers(1, 24))
    c = int(rng.choice([4, 32, 96, 100, 256, 960]))
    k = int(rng.choice([3, 3, 5]))
    stride = int(rng.choice([1, 2]))
    pad = k // 2
    g = gen()
    x = torch.randn(b, c, h, w, generator=g)
    wt = torch.randn(c, 1, k, k, generator=g) * 0.3
    bias = torch.randn(c, generator=g)
    act_name = str(rng.choice(["none", "silu", "relu"]))
    fn = {"none": lambda t: t, "silu": F.silu, "relu": F.relu}[act_name]
    exp = fn(F.conv2d(x, wt, bias, stride=stride, padding=pad, groups=c))
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = wt.reshape(c, k, k).permute(1, 2, 0).contiguous().to(dev)
    bd = bias.to(dev)
    ho, wo = exp.shape[-2:]
    out = torch.empty((b, ho, wo, c), device=dev)
    act = {"none": 0, "relu": 1, "silu": 3}[act_name]
    st = lib.isc_dwconv2d_nhwc(xd.data_ptr(), b, h, w, c, wd.data_ptr(), k, stride, pad, bd.data_ptr(), act,
                               out.data_ptr(), _lib.stream_handle(dev))
    _lib.check(st, "isc_dwconv2d_nhwc")
    got = out.permute(0, 3, 1, 2).cpu()
    desc = f"b{b} {h}x{w} c{c} k{k} s{stride} {act_name}"
    report("dwconv", torch.allclose(got, exp, rtol=1e-5, atol=1e-5), f"{desc}: {(got - exp).abs().max():.3g}")
    # the pooled mean and the gated output (isc_dwconv2d_nhwc_pool); shapes without them must say so
    gate = torch.rand(b, c, generator=g)
    gd = gate.to(dev)
    pooled = torch.full((b, c), float("nan"), device=dev)
    out2 = torch.full_like(out, float("nan"))
    st = lib.isc_dwconv2d_nhwc_pool(xd.data_ptr(), b, h, w, c, wd.data_ptr(), k, stride, pad, bd.data_ptr(), act,
                                    gd.data_ptr(), out2.data_ptr(), pooled.data_ptr(), _lib.stream_handle(dev))
    sweep = k == 3 and stride == 1
    if sweep and w <= 14:
        ok = st == 0 and torch.allclose(out2.permute(0, 3, 1, 2).cpu(), exp * gate[:, :, None, None], rtol=1e-5, atol=1e-5) \
            and torch.allclose(pooled.cpu(), exp.mean(dim=(2, 3)), rtol=1e-5, atol=1e-6)
        report("dwconv", ok, f"{desc} pooled + gated: status {st}")
    else:
        report("dwconv", st == _lib.ISC_ERR_UNSUPPORTED, f"{desc} pooled + gated outside the sweep shapes: status {st}")


def f_gated_and_centered():
    b, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 12)), int(rng.integers(1, 12))
    cin, cout = int(rng.choice([32, 96, 256])), int(rng.choice([4, 64, 100, 256]))
    g = gen()
    x = torch.randn(b, cin, h, w, generator=g)
    gate = torch.rand(b, cin, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin**0.5
    bias = torch.randn(cout, generator=g)
    exp = F.conv2d(x * gate[:, :, None, None], wt, bias)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    out = torch.empty((b, h, w, cout), device=dev)
    gd, wd, bd = gate.to(dev), wt.reshape(cout, cin).contiguous().to(dev), bias.to(dev)
    st = lib.isc_conv2d_nhwc_gated(xd.data_ptr(), b, h, w, cin, gd.data_ptr(), wd.data_ptr(), cout, 1, 1, 1, 0, bd.data_ptr(),
                                   None, 0, out.data_ptr(), _lib.stream_handle(dev))
    _lib.check(st, "isc_conv2d_nhwc_gated")
    got = out.permute(0, 3, 1, 2).cpu()
    err = float((got - exp).abs().max() / exp.abs().max().clamp_min(1e-30))
    report("gated conv", err < 2e-5, f"b{b} {h}x{w} cin{cin} cout{cout}: rel err {err:.3g}")
    # PCA projection: (x - mean) . w^T + bias
    n, f, k = int(rng.integers(1, 700)), int(rng.choice([32, 64, 1280])), int(rng.choice([4, 8, 64, 100]))
    xx, mean = torch.randn(n, f, generator=g) + 2.0, torch.randn(f, generator=g)
    ww = torch.randn(k, f, generator=g) / f**0.5
    exp2 = (xx - mean) @ ww.T
    xd2, md, wd2 = xx.to(dev), mean.to(dev), ww.to(dev)
    out2 = torch.empty((n, k), device=dev)
    _lib.check(lib.isc_linear_centered(xd2.data_ptr(), n, f, md.data_ptr(), wd2.data_ptr(), k, None, out2.data_ptr(),
                                       _lib.stream_handle(dev)), "isc_linear_centered")
    err2 = float((out2.cpu() - exp2).abs().max() / exp2.abs().max().clamp_min(1e-30))
    report("linear_centered", err2 < 2e-5, f"n{n} f{f} k{k}: rel err {err2:.3g}")


def f_pools_l2norm():
    b, h, w, c = int(rng.integers(1, 4)), int(rng.integers(1, 30)), int(rng.integers(1, 30)), int(rng.choice([4, 64, 100, 256]))
    g = gen()
    x = torch.randn(b, c, h, w, generator=g)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    s = _lib.stream_handle(dev)
    if h >= 2 and w >= 2:
        exp = F.max_pool2d(x, 3, 2, 1)
        ho, wo = exp.shape[-2:]
        out = torch.empty((b, ho, wo, c), device=dev)
        _lib.check(lib.isc_maxpool_nhwc(xd.data_ptr(), b, h, w, c, 3, 2, 1, out.data_ptr(), s), "isc_maxpool_nhwc")
        report("maxpool", torch.equal(out.permute(0, 3, 1, 2).cpu(), exp), f"b{b} {h}x{w} c{c}")
    avg = torch.empty((b, c), device=dev)
    _lib.check(lib.isc_global_avgpool_nhwc(xd.data_ptr(), b, h, w, c, avg.data_ptr(), s), "isc_global_avgpool_nhwc")
    report("avgpool", torch.allclose(avg.cpu(), x.mean((2, 3)), rtol=1e-5, atol=1e-6), f"b{b} {h}x{w} c{c}")
    from imagescry_amd import l2_normalize_channels
    got = l2_normalize_channels(x.to(dev)).cpu()
    report("l2norm", torch.allclose(got, F.normalize(x, dim=1), rtol=1e-5, atol=1e-6), f"b{b} c{c} {h}x{w}")
    got2 = l2_normalize_channels(xd.permute(0, 3, 1, 2)).cpu()  # channels-last view: the row-normalising path
    report("l2norm nhwc", torch.allclose(got2, F.normalize(x, dim=1), rtol=1e-5, atol=1e-6), f"b{b} c{c} {h}x{w}")


def f_merge():
    gq, q = int(rng.integers(1, 9)), int(rng.integers(1, 40))
    kin = int(rng.choice([1, 5, 10, 16, 100]))
    kout = int(rng.integers(1, min(120, gq * kin) + 1))
    g = gen()
    sc = torch.randn(gq, q, kin, generator=g)
    if rng.random() < 0.5:
        sc = (sc * 2).round() / 2  # many exact ties
    ix = torch.randint(0, 1 << 40, (gq, q, kin), generator=g)
    if rng.random() < 0.3:
        sc[0, :, -1] = -float("inf")
        ix[0, :, -1] = torch.iinfo(torch.int64).max
    from oracle import search_oracle
    exp_s, exp_i = search_oracle.topk_merge(sc.numpy(), ix.numpy(), kout)
    sd, idd = sc.to(dev), ix.to(dev)
    out_s = torch.empty((q, kout), device=dev)
    out_i = torch.empty((q, kout), dtype=torch.int64, device=dev)
    _lib.check(lib.isc_topk_merge(sd.data_ptr(), idd.data_ptr(), gq, q, kin, kout, 0, 0, out_s.data_ptr(), out_i.data_ptr(),
                                  _lib.stream_handle(dev)), "isc_topk_merge")
    ok = np.array_equal(out_i.cpu().numpy(), exp_i) and np.array_equal(out_s.cpu().numpy(), exp_s)
    report("merge", ok, f"G{gq} q{q} kin{kin} kout{kout}")


def f_head():
    """isc_pool_linear_l2norm against torch: random image counts (odd ones leave a half-empty workgroup), map sizes, widths."""
    b = int(rng.integers(1, 12))
    h, w = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    c = int(rng.choice([4, 64, 100, 512, 2048]))
    e = int(rng.choice([1, 7, 64, 768, 1000]))
    if 2 * (c + e) * 4 > 64 * 1024:
        e = 64
    g = gen()
    x = torch.randn(b, h, w, c, generator=g)
    wt = torch.randn(e, c, generator=g) / c**0.5
    bias = torch.randn(e, generator=g) if rng.random() < 0.7 else None
    norm = int(rng.random() < 0.5)
    xd, wd = x.to(dev), wt.to(dev)
    bd = bias.to(dev) if bias is not None else None
    out = torch.empty((b, e), device=dev)
    _lib.check(lib.isc_pool_linear_l2norm(xd.data_ptr(), b, h, w, c, wd.data_ptr(), _lib.ptr(bd), e, norm, 1e-12, out.data_ptr(),
                                          _lib.stream_handle(dev)), "isc_pool_linear_l2norm")
    exp = x.double().mean(dim=(1, 2)) @ wt.double().T + (bias.double() if bias is not None else 0.0)
    if norm:
        exp = F.normalize(exp, dim=1)
    exp = exp.float()
    err = float((out.cpu() - exp).abs().max() / exp.abs().max().clamp_min(1e-30))
    report("head", err < 5e-6, f"B{b} {h}x{w} C{c} E{e} bias={bias is not None} norm={norm} rel err {err:.3g}")


def f_pca_fit():
    """PCA.fit on the device (feature sums, centred transpose, Gram on the matrix cores, host eigh) against the float64
    covariance of the same rows: eigenvalues and the projector onto the kept axes."""
    from imagescry_amd import PCA
    n, f = int(rng.integers(2, 3000)), int(rng.choice([1, 3, 4, 17, 64, 130]))
    g = gen()
    x = torch.randn(n, f, generator=g) @ torch.randn(f, f, generator=g) + 10.0 * torch.randn(1, f, generator=g)
    kmax = int(rng.integers(1, f + 1))
    PCA.GRAM_CHUNK_ROWS = int(rng.choice([32, 512, 32768]))  # several chunks on small inputs too
    pca = PCA(max_num_components=kmax, min_explained_variance=1.0).fit(x.to(dev))
    PCA.GRAM_CHUNK_ROWS = 32768
    xc = x.double() - x.double().mean(dim=0, keepdim=True)
    ev = torch.linalg.eigvalsh(xc.T @ xc / (n - 1)).flip(0).clamp_min(0)
    rank = min(n, f)
    want = (ev[:rank] / ev[:rank].sum()).float()
    got = pca.explained_variance.cpu()
    ok = got.shape == want.shape and torch.allclose(got, want, rtol=2e-3, atol=2e-5)
    comp = pca.component_vectors.cpu().double()
    ok = ok and torch.allclose(comp.T @ comp, torch.eye(comp.shape[1], dtype=torch.float64), atol=1e-4)
    ok = ok and torch.allclose(pca.feature_means.cpu(), x.mean(dim=0, keepdim=True), rtol=1e-5, atol=1e-5)
    report("pca_fit", ok, f"n{n} f{f} kmax{kmax} max ev err {(got - want).abs().max():.3g}")


for name, fn in (("head", f_head), ("pca_fit", f_pca_fit), ("normalize", f_normalize), ("resize", f_resize), ("conv", f_conv), ("conv_dual", f_conv_dual), ("gemm", f_gemm),
                 ("attention", f_attention), ("layernorm", f_layernorm), ("dwconv", f_dwconv),
                 ("gated conv + linear_centered", f_gated_and_centered), ("pools + l2norm", f_pools_l2norm), ("merge", f_merge)):
    if only is None or any(o in name for o in only):
        run_family(name, fn)
print(f"{fails} failures", flush=True)
sys.exit(1 if fails else 0)
