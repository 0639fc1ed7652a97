"""GPU parity of the encoder blocks and of `ResNet50Embedder.predict_step` against the CPU oracle.

The reference pins encoder SHAPES only (tests/test_models/test_embedding.py:78-106); values are "parity
unpinned" by the reference and are held to the oracle here.  Tolerance policy: every kernel within
1e-5 relative of the torch float32 result of the same op; the end-to-end L2-normalised embedding within
1e-5 absolute (north_star's float32 bound) of the oracle, i.e. cosine >= 0.99999.
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import cases  # noqa: E402

from oracle import encoder_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


def _rel_err(got: torch.Tensor, exp: torch.Tensor) -> float:
    return float((got - exp).abs().max() / exp.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize(
    "b,h,w,cin,cout,k,stride,pad",
    [
        (2, 14, 14, 64, 256, 1, 1, 0),  # bottleneck expand
        (3, 9, 11, 256, 64, 1, 1, 0),  # bottleneck reduce, Cout = 64 tile, ragged pixel count
        (2, 15, 13, 64, 64, 3, 1, 1),  # 3x3 same
        (2, 16, 16, 128, 128, 3, 2, 1),  # 3x3 stride 2
        (2, 15, 15, 256, 512, 1, 2, 0),  # downsample 1x1 stride 2
        (5, 1, 1, 2048, 768, 1, 1, 0),  # the projection head as a 1x1 conv
        (1, 7, 7, 160, 64, 1, 1, 0),  # stem GEMM over im2col rows
        (1, 10, 10, 32, 100, 5, 1, 2),  # Cout not a tile multiple, 5x5
        (2, 13, 12, 64, 24, 1, 1, 0),  # Cout <= 32: the 32-channel tile, eight channels of it past Cout
        (3, 17, 17, 32, 32, 3, 1, 1),  # ... all 32 used, ragged pixel tile
        # large 1 x 1 layers with short K loops: ragged last pixel tile, 1 / 2 / 4 channel blocks of 256
        (8, 130, 130, 64, 256, 1, 1, 0),  # ResNet layer1 expand: two K steps per tile
        (4, 129, 129, 256, 512, 1, 1, 0),
        (2, 200, 200, 32, 1024, 1, 1, 0),  # ONE K step per tile
        # more tiles than resident workgroups (two per CU): every workgroup walks several tiles, staging the next
        # tile's first K step under the last one of the current tile -- odd and even K-step counts, all three tile shapes
        (8, 150, 150, 32, 64, 1, 1, 0),  # 64 x 256 tiles, one K step
        (8, 150, 150, 96, 64, 1, 1, 0),  # ... three
        (8, 150, 150, 64, 24, 1, 1, 0),  # 32 x 256 tiles, two K steps
        (6, 75, 75, 32, 128, 3, 1, 1),  # 128 x 128 tiles, nine K steps, padding taps
        # long tiles (>= 16 K steps) with a thin last round: whole rounds in one launch, the remainder as half tiles
        (5, 120, 120, 64, 128, 3, 1, 1),  # 563 tiles of 128 x 128 on 512 resident workgroups: 512 + 51 -> 102 halves
        (5, 170, 170, 64, 64, 3, 1, 1),  # 565 tiles of 64 x 256: 512 + 53 -> 106 halves of 64 x 128, ragged last half
        (1, 33, 35, 512, 192, 1, 1, 0),  # no whole round: 15 tiles of 64 x 256 as 30 half tiles (the last one empty)
    ],
)
@pytest.mark.parametrize("epilogue", ["plain", "bias_relu", "bias_res_relu"])
def test_conv2d_nhwc(b, h, w, cin, cout, k, stride, pad, epilogue, device: torch.device) -> None:
    from imagescry_amd import _lib
    from imagescry_amd.embedding import _conv
    from imagescry_amd.resnet50 import FoldedConv

    g = cases.gen(b * 1000 + cin + cout + k)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) if epilogue != "plain" else None
    exp = F.conv2d(x, wt, bias, stride=stride, padding=pad)
    res = None
    if epilogue == "bias_res_relu":
        res = torch.randn(exp.shape, generator=g)
        exp = exp + res
    if epilogue != "plain":
        exp = F.relu(exp)
    conv = FoldedConv(
        wt.permute(0, 2, 3, 1).contiguous().to(device),
        (bias if bias is not None else torch.zeros(cout)).to(device), k, stride, pad,
    )
    xn = x.permute(0, 2, 3, 1).contiguous().to(device)
    rn = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(device)
    act = _lib.ISC_ACT_NONE if epilogue == "plain" else _lib.ISC_ACT_RELU
    got = _conv(xn, conv, act, residual=rn).permute(0, 3, 1, 2).cpu()
    assert got.shape == exp.shape
    assert _rel_err(got, exp) < 1e-5


@pytest.mark.parametrize(
    "b,h,w,cin,cin2,cout,stride2",
    [
        (3, 56, 56, 64, 64, 256, 1),  # ResNet-50 layer1.0: conv3 + projection shortcut at one resolution, 64 x 256 ... tiles
        (8, 130, 130, 64, 64, 256, 1),  # more tiles than resident workgroups, ragged last pixel tile
        (2, 28, 28, 128, 256, 512, 2),  # layer2.0: the shortcut reads every second pixel of the 56 x 56 map
        (1, 9, 7, 32, 96, 36, 3),  # one + three K steps, stride 3, Cout not a tile multiple, odd sizes
    ],
)
def test_conv1x1_dual_input(b, h, w, cin, cin2, cout, stride2, device: torch.device) -> None:
    """`isc_conv2d_nhwc_dual`: a bottleneck's last 1 x 1 convolution with its projection shortcut folded in, against
    torch's `relu(conv3(t) + bias3 + downsample(x) + bias_d)` (torchvision Bottleneck.forward; the reference reaches its
    backbone at src/imagescry/models/embedding.py:167-177).  K-concatenation changes the order of the float32 sum only."""
    from imagescry_amd import _lib
    from imagescry_amd.embedding import _conv_dual
    from imagescry_amd.resnet50 import FoldedConv

    g = cases.gen(b + cin + cin2 + cout)
    h2, w2 = (h - 1) * stride2 + 1 + (stride2 > 1), (w - 1) * stride2 + 1  # (h2 - 1) // stride2 + 1 == h either way
    t = torch.randn(b, cin, h, w, generator=g)
    x = torch.randn(b, cin2, h2, w2, generator=g)
    w3 = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    wd = torch.randn(cout, cin2, 1, 1, generator=g) / cin2 ** 0.5
    b3, bd = torch.randn(cout, generator=g), torch.randn(cout, generator=g)
    exp = F.relu(F.conv2d(t, w3, b3) + F.conv2d(x, wd, bd, stride=stride2))
    fused = FoldedConv(torch.cat([w3.reshape(cout, cin), wd.reshape(cout, cin2)], dim=1).contiguous().to(device),
                       (b3 + bd).to(device), 1, stride2, 0)
    tn = t.permute(0, 2, 3, 1).contiguous().to(device)
    xn = x.permute(0, 2, 3, 1).contiguous().to(device)
    got = _conv_dual(tn, xn, fused, _lib.ISC_ACT_RELU).permute(0, 3, 1, 2).cpu()
    assert got.shape == exp.shape
    assert _rel_err(got, exp) < 1e-5
    # the shapes the kernel has no second-input form for are refused, not mis-computed
    lib = _lib.load()
    out = torch.empty((b, h, w, cout), device=device)
    bad = lib.isc_conv2d_nhwc_dual(tn.data_ptr(), b, h, w, cin, xn.data_ptr(), h2 + stride2, w2, cin2, stride2,
                                   fused.weight.data_ptr(), cout, fused.bias.data_ptr(), None, _lib.ISC_ACT_RELU,
                                   out.data_ptr(), _lib.stream_handle(device))
    assert bad == _lib.ISC_ERR_INVALID_ARG  # the second map does not cover the output grid at that stride


def test_conv1x1_silu_and_residual_after_activation(device: torch.device) -> None:
    """A large short-K 1 x 1 layer with the EfficientNetV2 epilogues: SiLU, and `act(conv + bias) + residual`."""
    from imagescry_amd import _lib
    from imagescry_amd.embedding import _conv
    from imagescry_amd.resnet50 import FoldedConv

    g = cases.gen(77)
    b, h, w, cin, cout = 4, 150, 150, 96, 256
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    bias = torch.randn(cout, generator=g)
    res = torch.randn(b, cout, h, w, generator=g)
    conv = FoldedConv(wt.permute(0, 2, 3, 1).contiguous().to(device), bias.to(device), 1, 1, 0)
    xn = x.permute(0, 2, 3, 1).contiguous().to(device)
    rn = res.permute(0, 2, 3, 1).contiguous().to(device)
    base = F.conv2d(x, wt, bias)
    got = _conv(xn, conv, _lib.ISC_ACT_SILU).permute(0, 3, 1, 2).cpu()
    assert _rel_err(got, F.silu(base)) < 1e-5
    got = _conv(xn, conv, _lib.ISC_ACT_SILU | _lib.ISC_ACT_RESIDUAL_AFTER, residual=rn).permute(0, 3, 1, 2).cpu()
    assert _rel_err(got, F.silu(base) + res) < 1e-5
    got = _conv(xn, conv, _lib.ISC_ACT_NONE, residual=rn).permute(0, 3, 1, 2).cpu()
    assert _rel_err(got, base + res) < 1e-5


@pytest.mark.parametrize("b,h,w,cout,k,stride,pad", [(2, 37, 29, 64, 7, 2, 3), (1, 16, 16, 128, 3, 1, 1), (3, 9, 20, 64, 5, 2, 2),
                                                      (4, 224, 224, 64, 7, 2, 3),  # ResNet-50's stem at its own size (halo-tile kernel)
                                                      (4, 224, 224, 24, 3, 2, 1)])  # EfficientNetV2's stem
def test_conv2d_stem_mode(b, h, w, cout, k, stride, pad, device: torch.device) -> None:
    """Cin == 4 (RGB + zero channel), the packed-K mode with eight filter taps per K step; weights [Cout, ceil(k*k/8)*8, 4]."""
    from imagescry_amd import _lib

    g = cases.gen(b + h + cout)
    x = torch.randn(b, 3, h, w, generator=g)
    wt = torch.randn(cout, 3, k, k, generator=g) / (3 * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    exp = F.relu(F.conv2d(x, wt, bias, stride=stride, padding=pad))
    taps = (k * k + 7) // 8 * 8
    wk = torch.zeros(cout, taps, 4)
    wk[:, : k * k, :3] = wt.permute(0, 2, 3, 1).reshape(cout, k * k, 3)
    lib = _lib.load()
    xd = x.to(device)
    x4 = torch.empty((b, h, w, 4), device=device)
    stream = _lib.stream_handle(device)
    _lib.check(lib.isc_nchw_to_nhwc(xd.data_ptr(), b, 3, h, w, 4, x4.data_ptr(), stream), "nchw_to_nhwc")
    ho, wo = exp.shape[-2:]
    out = torch.empty((b, ho, wo, cout), device=device)
    wd, bd = wk.to(device), bias.to(device)
    st = lib.isc_conv2d_nhwc(x4.data_ptr(), b, h, w, 4, wd.data_ptr(), cout, k, k, stride, pad, bd.data_ptr(), None,
                             _lib.ISC_ACT_RELU, out.data_ptr(), stream)
    _lib.check(st, "isc_conv2d_nhwc")
    assert _rel_err(out.permute(0, 3, 1, 2).cpu(), exp) < 1e-5


@pytest.mark.parametrize(
    "b,h,w,cin,cout,k,stride,pad",
    [
        (2, 31, 29, 24, 24, 3, 1, 1),  # EfficientNetV2-S stage 1 (6 chunks per tap: K steps straddle taps)
        (2, 30, 30, 24, 96, 3, 2, 1),  # stage 2 expand
        (2, 17, 19, 48, 192, 3, 1, 1),  # 48 channels: 12 chunks per tap
        (3, 20, 20, 96, 48, 1, 1, 0),  # Cin % 32 == 0 but Cout = 48: the plain mode, unpadded output
        (3, 20, 20, 48, 64, 1, 1, 0),  # 1 x 1 packed-K: 1.5 K steps, tail zero-padded
        (2, 9, 9, 80, 160, 1, 1, 0),  # EfficientNetV2-M: 80 = 2.5 K steps
        (2, 9, 9, 8, 132, 5, 2, 2),  # 2 chunks per tap, Cout over one 128 tile
        (4, 1, 1, 40, 960, 1, 1, 0),  # squeeze-excitation fc2 from an unpadded squeeze width
        (8, 150, 150, 24, 24, 3, 1, 1),  # packed-K over more tiles than resident workgroups (32 x 256 tiles)
        (8, 150, 150, 8, 48, 3, 2, 1),  # ... 64 x 256 tiles would need Cout > 32: 48 -> 64 x 256, stride 2
        # the halo-tile kernel (conv_halo.hip: Cin <= 24, Cout <= 64, square odd filter, "same" padding): image smaller than
        # one 16 x 16 tile, ragged tiles in both directions, 5 x 5 and 7 x 7 windows, stride 2 on odd sizes, 1 - 6 chunks per
        # pixel, Cout of one / two / three / four 16-channel blocks, more tiles than resident workgroups
        (1, 5, 7, 24, 24, 3, 1, 1),
        (3, 33, 47, 12, 40, 5, 1, 2),
        (2, 45, 31, 20, 64, 7, 2, 3),
        (2, 64, 64, 16, 4, 3, 2, 1),
        (5, 200, 208, 24, 24, 3, 1, 1),
    ],
)
@pytest.mark.parametrize("epilogue", ["bias_silu", "bias_silu_then_res"])
def test_conv2d_packed_k_mode(b, h, w, cin, cout, k, stride, pad, epilogue, device: torch.device) -> None:
    """Cin % 32 != 0: weights [Cout, ceil(k*k*Cin/32)*32], every 16-byte chunk of a K step finds its own filter tap."""
    from imagescry_amd import _lib

    g = cases.gen(b * 100 + cin + cout + k)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    exp = F.silu(F.conv2d(x, wt, bias, stride=stride, padding=pad))
    res = torch.randn(exp.shape, generator=g) if epilogue == "bias_silu_then_res" else None
    if res is not None:
        exp = exp + res
    kk = k * k * cin
    wk = torch.zeros(cout, (kk + 31) // 32 * 32)
    wk[:, :kk] = wt.permute(0, 2, 3, 1).reshape(cout, kk)
    lib = _lib.load()
    stream = _lib.stream_handle(device)
    xn = x.permute(0, 2, 3, 1).contiguous().to(device)
    rn = None if res is None else res.permute(0, 2, 3, 1).contiguous().to(device)
    ho, wo = exp.shape[-2:]
    out = torch.empty((b, ho, wo, cout), device=device)
    wd, bd = wk.to(device), bias.to(device)
    act = _lib.ISC_ACT_SILU | (_lib.ISC_ACT_RESIDUAL_AFTER if res is not None else 0)
    st = lib.isc_conv2d_nhwc(xn.data_ptr(), b, h, w, cin, wd.data_ptr(), cout, k, k, stride, pad, bd.data_ptr(),
                             _lib.ptr(rn), act, out.data_ptr(), stream)
    _lib.check(st, "isc_conv2d_nhwc")
    assert _rel_err(out.permute(0, 3, 1, 2).cpu(), exp) < 1e-5


def test_conv2d_packed_k_mode_refuses_gate(device: torch.device) -> None:
    from imagescry_amd import _lib

    lib = _lib.load()
    x = torch.zeros((1, 4, 4, 24), device=device)
    w = torch.zeros((32, 32), device=device)
    gate = torch.ones((1, 24), device=device)
    out = torch.empty((1, 4, 4, 32), device=device)
    st = lib.isc_conv2d_nhwc_gated(x.data_ptr(), 1, 4, 4, 24, gate.data_ptr(), w.data_ptr(), 32, 1, 1, 1, 0, None, None,
                                   _lib.ISC_ACT_NONE, out.data_ptr(), _lib.stream_handle(device))
    assert st == _lib.ISC_ERR_UNSUPPORTED


def test_im2col_maxpool_avgpool(device: torch.device) -> None:
    from imagescry_amd import _lib

    lib = _lib.load()
    stream = _lib.stream_handle(device)
    g = cases.gen(21)
    x = torch.randn(2, 3, 37, 29, generator=g)
    ho, wo = (37 + 6 - 7) // 2 + 1, (29 + 6 - 7) // 2 + 1
    patches = torch.empty((2, ho, wo, 160), device=device)
    xd = x.to(device)
    _lib.check(lib.isc_im2col_nchw(xd.data_ptr(), 2, 3, 37, 29, 7, 7, 2, 3, 160, patches.data_ptr(), stream), "im2col")
    unf = F.unfold(x, kernel_size=7, stride=2, padding=3)  # [B, C*49, L], K ordered (c, r, s)
    exp = unf.reshape(2, 3, 49, ho * wo).permute(0, 3, 2, 1).reshape(2, ho, wo, 147)  # -> (r, s, c)
    got = patches.cpu()
    np.testing.assert_array_equal(got[..., :147].numpy(), exp.numpy())
    assert float(got[..., 147:].abs().max()) == 0.0

    y = torch.randn(2, 64, 21, 17, generator=g)
    yd = y.permute(0, 2, 3, 1).contiguous().to(device)
    hp, wp = (21 + 2 - 3) // 2 + 1, (17 + 2 - 3) // 2 + 1
    pooled = torch.empty((2, hp, wp, 64), device=device)
    _lib.check(lib.isc_maxpool_nhwc(yd.data_ptr(), 2, 21, 17, 64, 3, 2, 1, pooled.data_ptr(), stream), "maxpool")
    np.testing.assert_array_equal(pooled.permute(0, 3, 1, 2).cpu().numpy(), F.max_pool2d(y, 3, 2, 1).numpy())

    avg = torch.empty((2, 64), device=device)
    _lib.check(lib.isc_global_avgpool_nhwc(yd.data_ptr(), 2, 21, 17, 64, avg.data_ptr(), stream), "avgpool")
    np.testing.assert_allclose(avg.cpu().numpy(), y.mean(dim=(2, 3)).numpy(), rtol=1e-5, atol=1e-6)

    z = torch.empty((2, 37, 29, 8), device=device)
    _lib.check(lib.isc_nchw_to_nhwc(xd.data_ptr(), 2, 3, 37, 29, 8, z.data_ptr(), stream), "nchw_to_nhwc")
    np.testing.assert_array_equal(z[..., :3].cpu().numpy(), x.permute(0, 2, 3, 1).numpy())
    assert float(z[..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize(
    "b,h,w,c,r,stride,pad",
    [(3, 112, 112, 64, 3, 2, 1), (2, 2, 2, 8, 3, 2, 1), (1, 3, 5, 4, 3, 2, 1), (2, 7, 64, 12, 3, 2, 1), (1, 1, 9, 4, 3, 2, 1),
     (2, 10, 11, 8, 2, 2, 0), (1, 9, 9, 16, 3, 1, 1), (1, 12, 7, 4, 5, 2, 2)],
)
def test_maxpool_shapes(b, h, w, c, r, stride, pad, device: torch.device) -> None:
    """Bit-exact against F.max_pool2d, -inf rows included (ResNet-50's 3 x 3 / 2 / 1 at its own size, degenerate images,
    other windows)."""
    from imagescry_amd import _lib

    g = cases.gen(b * h + w + c)
    x = torch.randn(b, c, h, w, generator=g)
    x[0, :, 0, :] = float("-inf")
    xd = x.permute(0, 2, 3, 1).contiguous().to(device)
    ho, wo = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - r) // stride + 1
    out = torch.full((b, ho, wo, c), float("nan"), device=device)
    _lib.check(_lib.load().isc_maxpool_nhwc(xd.data_ptr(), b, h, w, c, r, stride, pad, out.data_ptr(), _lib.stream_handle(device)),
               "maxpool")
    np.testing.assert_array_equal(out.permute(0, 3, 1, 2).cpu().numpy(), F.max_pool2d(x, r, stride, pad).numpy())


@pytest.mark.parametrize("shape", [(2, 3, 64, 64), (3, 3, 96, 80), (1, 3, 35, 42)])
def test_resnet50_forward_matches_oracle(shape: tuple[int, ...], device: torch.device) -> None:
    from imagescry_amd import ResNet50Embedder, resnet50

    sd = resnet50.make_state_dict(seed=3, randomize_bn=True)
    model = ResNet50Embedder(state_dict=sd).to(device)
    x = torch.randn(shape, generator=cases.gen(shape[2])).clip(-3, 3)
    with torch.no_grad():
        exp = encoder_oracle.resnet50_forward(x, sd)
    got = model.forward(x.to(device)).cpu()
    assert got.shape == (shape[0], 768, 1, 1)
    assert _rel_err(got, exp) < 2e-5


@pytest.mark.parametrize("max_side_length", [640, 48])
def test_predict_step_matches_oracle(max_side_length: int, device: torch.device) -> None:
    """preprocess (with and without the resize branch) -> forward -> L2 normalise, end to end."""
    from imagescry_amd import ImageBatch, ResNet50Embedder, resnet50

    sd = resnet50.make_state_dict(seed=1, randomize_bn=True)
    model = ResNet50Embedder(state_dict=sd, max_side_length=max_side_length).to(device)
    images = cases.images_u8((4, 3, 70, 50), seed=17)
    batch = ImageBatch(indices=torch.tensor([7, 3, 9, 11]), images=images).to(device)
    out = model.predict_step(batch)
    exp = encoder_oracle.predict_step_embeddings(images, sd, max_side_length)
    assert out.embeddings.shape == (4, 768, 1, 1) and out.embedding_dim == 768 and out.spatial_dims == (1, 1)
    assert out.indices.cpu().tolist() == [7, 3, 9, 11]
    got = out.embeddings.cpu()
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=0, atol=1e-5)
    cos = (got.flatten(1) * exp.flatten(1)).sum(1)
    assert float(cos.min()) >= 0.99999
    assert torch.allclose(got.flatten(1).norm(dim=1), torch.ones(4), atol=1e-6)


@pytest.mark.parametrize("height", [35, 64, 128])
@pytest.mark.parametrize("width", [42, 73, 96])
@pytest.mark.parametrize("batch_size", [1, 2, 3])
def test_embedding_predict_step_shapes(batch_size: int, height: int, width: int, device: torch.device) -> None:
    """The reference's own encoder test (tests/test_models/test_embedding.py:78-106), for this encoder's
    output geometry: one `embedding_dim` vector per image."""
    from imagescry_amd import ImageBatch, ResNet50Embedder

    model = _shared_model(device)
    batch = ImageBatch(
        indices=torch.arange(batch_size),
        images=torch.randint(0, 256, (batch_size, 3, height, width)).to(torch.uint8),
    ).to(model.device)
    out = model.predict_step(batch)
    assert out.embeddings.shape == (batch_size, model.embedding_dim, 1, 1)
    assert isinstance(model, ResNet50Embedder)


_MODEL = None


def _shared_model(device: torch.device):
    global _MODEL
    if _MODEL is None:
        from imagescry_amd import ResNet50Embedder

        _MODEL = ResNet50Embedder().to(device)
    return _MODEL


def test_embed_images_order_and_types(device: torch.device) -> None:
    """One EmbeddingBatch per input batch, loader order, indices untouched; equal to direct predict_step calls
    (SURVEY.md section 8 row a8)."""
    from imagescry_amd import EmbeddingBatch, ImageBatch

    model = _shared_model(device)
    loader = [
        ImageBatch(indices=torch.tensor([5, 6]), images=cases.images_u8((2, 3, 40, 40), seed=1)),
        ImageBatch(indices=torch.tensor([0]), images=cases.images_u8((1, 3, 33, 47), seed=2)),
        ImageBatch(indices=torch.tensor([9, 8, 7]), images=cases.images_u8((3, 3, 40, 40), seed=3)),
    ]
    results = model.embed_images(loader)
    assert len(results) == 3 and all(isinstance(r, EmbeddingBatch) for r in results)
    for batch, res in zip(loader, results):
        assert res.indices.cpu().tolist() == batch.indices.tolist()
        direct = model.predict_step(batch.to(device))
        assert torch.equal(direct.embeddings, res.embeddings)
    with pytest.raises(ValueError):
        model.embed_images(loader, accelerator="cpu")


def test_config0_encode_then_self_search(device: torch.device) -> None:
    """BASELINE config 0 at its own size on the GPU path: encode 256 random 224x224 images with the repo embedder, search
    the 256 embeddings against themselves (brute-force cosine top-10), every image finds itself first with score
    1 +- 1e-5; the whole search result equals the oracle's on the embeddings the GPU produced, and a 64-image slice of the
    encode -- normalisation statistics are batch-wide, so a slice is its own batch -- is compared with the CPU oracle."""
    from imagescry_amd import EmbeddingBank, ImageBatch, ResNet50Embedder, resnet50
    from oracle import search_oracle

    sd = resnet50.make_state_dict(seed=0)
    model = ResNet50Embedder(state_dict=sd).to(device)
    n = 256
    images = cases.images_u8((n, 3, 224, 224))
    out = model.predict_step(ImageBatch(indices=torch.arange(n), images=images).to(device))
    flat = out.get_flat_vectors()
    assert flat.shape == (n, 768) and flat.dtype == torch.float32
    bank = EmbeddingBank.from_batches([out], dtype=torch.float32)
    scores, indices = bank.search(flat, 10)
    assert indices[:, 0].cpu().tolist() == list(range(n))
    assert torch.allclose(scores[:, 0].cpu(), torch.ones(n), atol=1e-5)
    exp_s, exp_i = search_oracle.cosine_topk(bank.bank.cpu(), flat.cpu(), 10)
    np.testing.assert_array_equal(indices.cpu().numpy(), exp_i)
    np.testing.assert_allclose(scores.cpu().numpy(), exp_s, rtol=0, atol=1e-6)
    part = model.predict_step(ImageBatch(indices=torch.arange(64), images=images[:64]).to(device))
    exp = encoder_oracle.predict_step_embeddings(images[:64], sd)
    np.testing.assert_allclose(part.embeddings.cpu().numpy(), exp.numpy(), rtol=0, atol=1e-5)


@pytest.mark.parametrize("name", ["resnet50", "efficientnet_s"])
def test_predict_step_equals_the_unfused_composition(name: str, device: torch.device) -> None:
    """`predict_step` hands the stem a channels-last normalised batch straight from the normalisation kernel; the public
    `preprocess` (NCHW) -> `forward` -> L2-normalise composition (reference: embedding.py:70-74) gives the same bits."""
    from imagescry_amd import EfficientNetEmbedder, ImageBatch, ResNet50Embedder
    from imagescry_amd.embedding import l2_normalize_channels

    model = (ResNet50Embedder(seed=1) if name == "resnet50" else EfficientNetEmbedder(backbone_size="s", seed=1)).to(device)
    images = cases.images_u8((3, 3, 70, 90)).to(device)
    out = model.predict_step(ImageBatch(indices=torch.arange(3, device=device), images=images)).embeddings
    ref = l2_normalize_channels(model.forward(model.preprocess(images)))
    assert out.shape == ref.shape
    assert torch.equal(out, ref)



@pytest.mark.parametrize("b,hw,c,e", [(5, (7, 7), 2048, 768), (1, (3, 2), 64, 8), (8, (1, 1), 512, 1000)])
def test_fused_pool_linear_l2norm_tail(b: int, hw: tuple[int, int], c: int, e: int, device: torch.device) -> None:
    """`isc_pool_linear_l2norm` (one launch) against its three-launch composition -- isc_global_avgpool_nhwc,
    isc_conv2d_nhwc 1 x 1, isc_l2norm_channels -- and against torch: pooled values and the normalisation arithmetic are
    the same bits as the separate kernels', the projection agrees with a float64 product to float32 rounding."""
    from imagescry_amd import _lib
    from imagescry_amd.embedding import l2_normalize_channels

    g = cases.gen(b + c)
    h, w = hw
    x = torch.randn(b, h, w, c, generator=g)
    wt = torch.randn(e, c, generator=g) / c**0.5
    bias = torch.randn(e, generator=g)
    lib = _lib.load()
    xd, wd, bd = x.to(device), wt.to(device), bias.to(device)
    stream = _lib.stream_handle(device)
    plain = torch.empty((b, e), device=device)
    normed = torch.empty((b, e), device=device)
    for out, flag in ((plain, 0), (normed, 1)):
        _lib.check(lib.isc_pool_linear_l2norm(xd.data_ptr(), b, h, w, c, wd.data_ptr(), bd.data_ptr(), e, flag, 1e-12,
                                              out.data_ptr(), stream), "isc_pool_linear_l2norm")
    pooled = torch.empty((b, c), device=device)
    _lib.check(lib.isc_global_avgpool_nhwc(xd.data_ptr(), b, h, w, c, pooled.data_ptr(), stream), "avgpool")
    exp = (pooled.cpu().double() @ wt.double().T + bias.double()).float()
    assert _rel_err(plain.cpu(), exp) < 2e-6
    # the normalisation: the same bits as the separate kernel applied to the fused projection
    assert torch.equal(normed, l2_normalize_channels(plain[:, :, None, None]).reshape(b, e))
    np.testing.assert_allclose(normed.cpu().numpy(), F.normalize(exp, dim=1).numpy(), rtol=0, atol=1e-6)
