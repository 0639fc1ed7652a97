"""EfficientNetV2 feature extractor on the GPU against the oracle, and the
reference's own encoder test (tests/test_models/test_embedding.py:78-106: output shape for 27 input geometries)."""

from __future__ import annotations

import math
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import cases  # noqa: E402

from oracle import efficientnet_oracle, encoder_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


def _stages(size: str):
    from imagescry_amd import efficientnet

    return [[(b.kind, b.expand, b.stride, b.cin, b.cout) for b in stage] for stage in efficientnet.block_specs(size)]


def test_depthwise_and_gated_conv_kernels(device: torch.device) -> None:
    from imagescry_amd import _lib

    lib = _lib.load()
    stream = _lib.stream_handle(device)
    g = cases.gen(3)
    for stride, h, w, c in ((1, 9, 11, 96), (2, 14, 14, 256), (2, 7, 5, 32)):
        x = torch.randn(2, c, h, w, generator=g)
        wt = torch.randn(c, 1, 3, 3, generator=g) * 0.3
        bias = torch.randn(c, generator=g)
        exp = F.silu(F.conv2d(x, wt, bias, stride=stride, padding=1, groups=c))
        xd = x.permute(0, 2, 3, 1).contiguous().to(device)
        wd = wt[:, 0].permute(1, 2, 0).contiguous().to(device)
        bd = bias.to(device)
        out = torch.empty((2, exp.shape[2], exp.shape[3], c), device=device)
        st = lib.isc_dwconv2d_nhwc(xd.data_ptr(), 2, h, w, c, wd.data_ptr(), 3, stride, 1, bd.data_ptr(), _lib.ISC_ACT_SILU,
                                   out.data_ptr(), stream)
        _lib.check(st, "isc_dwconv2d_nhwc")
        np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)
    # gated 1x1 projection with residual: out = conv(x * gate) + bias + res
    x = torch.randn(3, 64, 6, 5, generator=g)
    gate = torch.sigmoid(torch.randn(3, 64, generator=g))
    wt = torch.randn(32, 64, 1, 1, generator=g) * 0.2
    bias = torch.randn(32, generator=g)
    res = torch.randn(3, 32, 6, 5, generator=g)
    exp = F.conv2d(x * gate[:, :, None, None], wt, bias) + res
    xd = x.permute(0, 2, 3, 1).contiguous().to(device)
    out = torch.empty((3, 6, 5, 32), device=device)
    gd, wd, bd = gate.to(device), wt.permute(0, 2, 3, 1).contiguous().to(device), bias.to(device)
    rd = res.permute(0, 2, 3, 1).contiguous().to(device)
    st = lib.isc_conv2d_nhwc_gated(xd.data_ptr(), 3, 6, 5, 64, gd.data_ptr(), wd.data_ptr(), 32, 1, 1, 1, 0, bd.data_ptr(),
                                   rd.data_ptr(), _lib.ISC_ACT_NONE, out.data_ptr(), stream)
    _lib.check(st, "isc_conv2d_nhwc_gated")
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)
    # residual added after the activation (expand == 1 FusedMBConv)
    exp = F.silu(F.conv2d(x, wt, bias)) + res
    st = lib.isc_conv2d_nhwc(xd.data_ptr(), 3, 6, 5, 64, wd.data_ptr(), 32, 1, 1, 1, 0, bd.data_ptr(), rd.data_ptr(),
                             _lib.ISC_ACT_SILU | _lib.ISC_ACT_RESIDUAL_AFTER, out.data_ptr(), stream)
    _lib.check(st, "isc_conv2d_nhwc")
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize(
    "b,h,w,c,k,stride",
    [
        (3, 14, 14, 64, 3, 1),  # row-sweep kernel, two strips: pooled by one atomicAdd per strip
        (5, 7, 7, 96, 3, 1),  # one strip: pooled written directly
        (2, 9, 10, 40, 3, 1),  # ragged second strip (3 of 7 columns), H not a multiple of 3
        (2, 1, 1, 32, 3, 1),  # a single pixel
        (2, 5, 30, 32, 3, 1),  # five strips: sweep kernel, pooled by the separate pass
        (2, 15, 15, 64, 3, 2),  # stride 2: the one-pixel-per-thread kernel + separate pooling
        (2, 12, 12, 32, 5, 1),  # 5 x 5: same
    ],
)
def test_depthwise_with_pooled_mean(b, h, w, c, k, stride, device: torch.device) -> None:
    """isc_dwconv2d_nhwc_pool against torch: y = SiLU(dwconv + bias) [* gate], pooled = mean of the un-gated y over the
    image; the pooling alone and the gated output in the shapes that have them; isc_dwconv2d_nhwc returns the same y."""
    from imagescry_amd import _lib

    lib = _lib.load()
    stream = _lib.stream_handle(device)
    g = cases.gen(b + h * 31 + w + c)
    x = torch.randn(b, c, h, w, generator=g)
    wt = torch.randn(c, 1, k, k, generator=g) * 0.3
    bias = torch.randn(c, generator=g)
    gate = torch.rand(b, c, generator=g)
    exp = F.silu(F.conv2d(x, wt, bias, stride=stride, padding=k // 2, groups=c))
    xd = x.permute(0, 2, 3, 1).contiguous().to(device)
    wd = wt[:, 0].permute(1, 2, 0).contiguous().to(device)
    bd, gd = bias.to(device), gate.to(device)
    ho, wo = exp.shape[2], exp.shape[3]

    def run(gate_t, want_y, want_pooled):
        out = torch.full((b, ho, wo, c), float("nan"), device=device) if want_y else None
        pooled = torch.full((b, c), float("nan"), device=device) if want_pooled else None
        st = lib.isc_dwconv2d_nhwc_pool(xd.data_ptr(), b, h, w, c, wd.data_ptr(), k, stride, k // 2, bd.data_ptr(),
                                        _lib.ISC_ACT_SILU, _lib.ptr(gate_t), _lib.ptr(out), _lib.ptr(pooled), stream)
        return st, out, pooled

    st, out, pooled = run(None, True, True)
    assert st == 0
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pooled.cpu().numpy(), exp.mean(dim=(2, 3)).numpy(), rtol=1e-5, atol=1e-6)
    out2 = torch.empty_like(out)
    st = lib.isc_dwconv2d_nhwc(xd.data_ptr(), b, h, w, c, wd.data_ptr(), k, stride, k // 2, bd.data_ptr(),
                               _lib.ISC_ACT_SILU, out2.data_ptr(), stream)
    _lib.check(st, "isc_dwconv2d_nhwc")
    assert torch.equal(out, out2)
    st, out3, _ = run(None, True, False)  # pooled == NULL is the plain depthwise convolution
    assert st == 0 and torch.equal(out, out3)

    sweep = k == 3 and stride == 1
    st, _, pooled_only = run(None, False, True)
    st_g, out_g, pooled_g = run(gd, True, True)
    if sweep and w <= 14:
        assert st == 0 and torch.equal(pooled_only, pooled)  # same sums in the same order
        assert st_g == 0 and torch.equal(pooled_g, pooled)  # the mean is of the un-gated output
        # y * gate is one float32 multiply of the un-gated y
        assert torch.equal(out_g, out * gd[:, None, None, :])
    elif sweep:
        assert st == _lib.ISC_ERR_UNSUPPORTED  # pooling wider images needs y
        st_g, out_g, _ = run(gd, True, False)
        assert st_g == 0 and torch.equal(out_g, out * gd[:, None, None, :])
    else:
        assert st == _lib.ISC_ERR_UNSUPPORTED and st_g == _lib.ISC_ERR_UNSUPPORTED


@pytest.mark.parametrize("act_name", ["none", "relu", "gelu", "sigmoid"])
def test_depthwise_sweep_other_activations(act_name: str, device: torch.device) -> None:
    """The row-sweep depthwise kernel is specialised for SiLU; every other activation takes its generic instantiation."""
    from imagescry_amd import _lib

    lib = _lib.load()
    act, fn = {"none": (_lib.ISC_ACT_NONE, lambda v: v), "relu": (_lib.ISC_ACT_RELU, F.relu),
               "gelu": (_lib.ISC_ACT_GELU, F.gelu), "sigmoid": (_lib.ISC_ACT_SIGMOID, torch.sigmoid)}[act_name]
    g = cases.gen(5)
    b, c, h, w = 2, 32, 9, 11
    x = torch.randn(b, c, h, w, generator=g)
    wt = torch.randn(c, 1, 3, 3, generator=g) * 0.3
    bias = torch.randn(c, generator=g)
    exp = fn(F.conv2d(x, wt, bias, padding=1, groups=c))
    xd = x.permute(0, 2, 3, 1).contiguous().to(device)
    wd = wt[:, 0].permute(1, 2, 0).contiguous().to(device)
    bd = bias.to(device)
    out = torch.empty((b, h, w, c), device=device)
    pooled = torch.empty((b, c), device=device)
    st = lib.isc_dwconv2d_nhwc_pool(xd.data_ptr(), b, h, w, c, wd.data_ptr(), 3, 1, 1, bd.data_ptr(), act, None,
                                    out.data_ptr(), pooled.data_ptr(), _lib.stream_handle(device))
    _lib.check(st, "isc_dwconv2d_nhwc_pool")
    np.testing.assert_allclose(out.permute(0, 3, 1, 2).cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pooled.cpu().numpy(), exp.mean(dim=(2, 3)).numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("b,c,s,ld1,ld2", [(5, 1536, 64, 1536, 64), (3, 256, 16, 256, 32), (1, 40, 12, 64, 32), (2, 3840, 160, 3840, 160)])
def test_se_gate(b, c, s, ld1, ld2, device: torch.device) -> None:
    """isc_se_gate against torch: sigmoid(fc2(silu(fc1(pooled)))), weight rows with a stride wider than the row."""
    from imagescry_amd import _lib

    lib = _lib.load()
    g = cases.gen(b + c + s)
    pooled = torch.randn(b, c, generator=g)
    w1 = torch.randn(s, c, generator=g) / c ** 0.5
    b1 = torch.randn(s, generator=g)
    w2 = torch.randn(c, s, generator=g) / s ** 0.5
    b2 = torch.randn(c, generator=g)
    exp = torch.sigmoid(F.silu(pooled.double() @ w1.double().T + b1.double()) @ w2.double().T + b2.double()).float()
    w1p = torch.zeros(s, ld1)
    w1p[:, :c] = w1
    w2p = torch.zeros(c, ld2)
    w2p[:, :s] = w2
    pd, w1d, b1d, w2d, b2d = (t.to(device) for t in (pooled, w1p, b1, w2p, b2))
    gate = torch.empty((b, c), device=device)
    st = lib.isc_se_gate(pd.data_ptr(), b, c, w1d.data_ptr(), ld1, b1d.data_ptr(), s, w2d.data_ptr(), ld2, b2d.data_ptr(),
                         gate.data_ptr(), _lib.stream_handle(device))
    _lib.check(st, "isc_se_gate")
    np.testing.assert_allclose(gate.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(2, 3, 64, 96), (1, 3, 35, 42)])
def test_efficientnet_s_forward_matches_oracle(shape: tuple[int, ...], device: torch.device) -> None:
    from imagescry_amd import EfficientNetEmbedder, efficientnet

    sd = efficientnet.make_state_dict("s", seed=5, randomize_bn=True)
    model = EfficientNetEmbedder(state_dict=sd).to(device)
    x = torch.randn(shape, generator=cases.gen(shape[2])).clip(-3, 3)
    with torch.no_grad():
        exp = efficientnet_oracle.features(x, sd, _stages("s"))
    got = model.forward(x.to(device)).cpu()
    assert got.shape == exp.shape == (shape[0], 1280, math.ceil(shape[2] / 32), math.ceil(shape[3] / 32))
    err = float((got - exp).abs().max() / exp.abs().max())
    assert err < 5e-5, err


@pytest.mark.parametrize("size", ["m", "l"])
def test_efficientnet_m_l_forward_matches_oracle(size: str, device: torch.device) -> None:
    """The wider variants: channel counts that are not multiples of 32 (80, 176, 304 in "m": packed-K mode), squeeze
    widths up to 160, expanded widths up to 3840 in the SE gate kernel."""
    from imagescry_amd import EfficientNetEmbedder, efficientnet

    sd = efficientnet.make_state_dict(size, seed=7, randomize_bn=True)
    model = EfficientNetEmbedder(backbone_size=size, state_dict=sd).to(device)
    x = torch.randn((1, 3, 64, 96), generator=cases.gen(11)).clip(-3, 3)
    with torch.no_grad():
        exp = efficientnet_oracle.features(x, sd, _stages(size))
    got = model.forward(x.to(device)).cpu()
    assert got.shape == exp.shape == (1, 1280, 2, 3)
    err = float((got - exp).abs().max() / exp.abs().max())
    assert err < 5e-5, err


def test_efficientnet_predict_step_matches_oracle(device: torch.device) -> None:
    from imagescry_amd import EfficientNetEmbedder, ImageBatch, efficientnet

    sd = efficientnet.make_state_dict("s", seed=6, randomize_bn=True)
    model = EfficientNetEmbedder(state_dict=sd, max_side_length=64).to(device)
    images = cases.images_u8((2, 3, 90, 70), seed=3)  # long side 90 > 64: resize branch
    out = model.predict_step(ImageBatch(indices=torch.tensor([1, 0]), images=images).to(device))
    with torch.no_grad():
        x = encoder_oracle.preprocess(images, 64)
        exp = encoder_oracle.l2_normalize_channels(efficientnet_oracle.features(x, sd, _stages("s")))
    got = out.embeddings.cpu()
    assert got.shape == exp.shape == (2, 1280, 2, 2)
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=0, atol=1e-5)
    flat = out.get_flat_vectors()
    assert flat.shape == (8, 1280) and torch.allclose(flat.norm(dim=1).cpu(), torch.ones(8), atol=1e-5)
    assert torch.equal(flat.cpu(), got.permute(0, 2, 3, 1).reshape(-1, 1280))


@pytest.mark.parametrize("height", [35, 64, 128])
@pytest.mark.parametrize("width", [42, 73, 96])
@pytest.mark.parametrize("batch_size", [1, 2, 3])
def test_embedding_predict_step(batch_size: int, height: int, width: int, device: torch.device) -> None:
    """The reference's test, verbatim in structure (tests/test_models/test_embedding.py:78-106)."""
    from imagescry_amd import ImageBatch

    model = _model(device)
    image_batch = ImageBatch(
        indices=torch.arange(batch_size),
        images=torch.randint(0, 256, (batch_size, 3, height, width)).to(torch.uint8),
    ).to(model.device)
    embedding_batch = model.predict_step(image_batch)
    downsample_factor = 32
    assert embedding_batch.embeddings.shape == (
        batch_size, model.embedding_dim, math.ceil(height / downsample_factor), math.ceil(width / downsample_factor),
    )


_MODEL = None


def _model(device: torch.device):
    global _MODEL
    if _MODEL is None:
        from imagescry_amd import EfficientNetEmbedder

        _MODEL = EfficientNetEmbedder().to(device)
    return _MODEL


def test_constructor_contract(device: torch.device) -> None:
    from imagescry_amd import EfficientNetEmbedder

    assert _model(device).embedding_dim == 1280 and _model(device).hparams == {"backbone_size": "s", "max_side_length": 640}
    with pytest.raises(ValueError):
        EfficientNetEmbedder(backbone_size="xl")
    with pytest.raises(RuntimeError):
        EfficientNetEmbedder(pretrained=True)
