"""GPU parity of the preprocess kernels against the CPU oracle (reference transforms.py:58-126, embedding.py:74)."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import cases  # noqa: E402

from oracle import encoder_oracle, transforms_oracle  # noqa: E402

pytestmark = pytest.mark.gpu

GOLDEN = np.load(Path(__file__).resolve().parent / "golden" / "preprocess.npz")
# float32 tolerance of north_star is 1e-5; normalised pixels are O(1), resized raw pixels are O(255)
NORM_ATOL = 2e-6
RESIZE_RTOL, RESIZE_ATOL = 2e-6, 1e-5


def test_normalize_matches_golden_and_reference_property(device: torch.device) -> None:
    from imagescry_amd import normalize_per_channel

    small = cases.images_u8((4, 3, 30, 45))
    out = normalize_per_channel(small.to(device), min_value=-3, max_value=3).cpu()
    np.testing.assert_allclose(out.numpy(), GOLDEN["small_out"], rtol=0, atol=NORM_ATOL)
    # the reference's own test (tests/test_image/test_transform.py:14-24): mean ~ 0, std ~ 1 at atol 1e-4
    img = cases.reference_test_image().float().unsqueeze(0)
    norm = normalize_per_channel(img.to(device)).cpu()
    np.testing.assert_allclose(norm.numpy(), GOLDEN["ref_norm"], rtol=0, atol=NORM_ATOL)
    assert torch.allclose(norm.mean((-2, -1)), torch.zeros(1, 3), atol=1e-4)
    assert torch.allclose(norm.std((-2, -1)), torch.ones(1, 3), atol=1e-4)


@pytest.mark.parametrize("shape", [(8, 3, 224, 224), (3, 3, 64, 48), (2, 1, 7, 9), (1, 3, 1, 5), (5, 4, 33, 17)])
@pytest.mark.parametrize("clip", [None, (-3.0, 3.0), (None, 1.5)])
def test_normalize_u8_vs_oracle(shape: tuple[int, ...], clip, device: torch.device) -> None:
    from imagescry_amd import normalize_per_channel

    x = cases.images_u8(shape, seed=sum(shape))
    kw = {} if clip is None else {"min_value": clip[0], "max_value": clip[1]}
    exp = transforms_oracle.normalize_per_channel(x, **kw)
    got = normalize_per_channel(x.to(device), **kw).cpu()
    assert got.dtype == torch.float32 and got.shape == x.shape
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=0, atol=NORM_ATOL)


@pytest.mark.parametrize("shape", [(8, 3, 224, 224), (3, 3, 64, 48), (2, 1, 7, 9), (1, 3, 1, 5), (5, 4, 33, 17)])
@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
def test_normalize_channels_last_is_the_same_values(shape: tuple[int, ...], dtype, device: torch.device) -> None:
    """`_nhwc4=True` (what `predict_step` feeds the convolution stems): bit for bit the NCHW result, moved to
    `[B, H, W, 4]` with the channels past C zero -- vectorised (H*W % 4 == 0) and scalar paths, per-image statistics."""
    from imagescry_amd import normalize_per_channel

    x = cases.images_u8(shape)
    if dtype == torch.float32:
        x = x.float() * 0.37 - 11.0
    x = x.to(device)
    b, c = shape[:2]
    for kwargs in ({}, {"channel_means": torch.linspace(-1, 1, b * c).reshape(b, c, 1, 1).to(device),
                        "channel_stds": torch.linspace(0.5, 2, c).reshape(1, c, 1, 1).to(device)}):
        ref = normalize_per_channel(x, min_value=-3, max_value=3, **kwargs)
        got = normalize_per_channel(x, min_value=-3, max_value=3, _nhwc4=True, **kwargs)
        assert got.shape == (b, shape[2], shape[3], 4)
        assert torch.equal(got[..., :c], ref.permute(0, 2, 3, 1))
        assert not got[..., c:].any()


def test_channel_statistics_are_exact_for_u8(device: torch.device) -> None:
    """u8 sums are integer-exact, so mean / unbiased std equal the float64 values rounded to float32."""
    from imagescry_amd.transforms import _channel_stats

    x = cases.images_u8((16, 3, 224, 224), seed=2)
    mean, std = _channel_stats(x.to(device))
    m64, s64 = transforms_oracle.channel_stats_f64(x)
    np.testing.assert_array_equal(mean.cpu().numpy(), m64.float().numpy())
    np.testing.assert_array_equal(std.cpu().numpy(), s64.float().numpy())


def test_normalize_float_input_and_supplied_statistics(device: torch.device) -> None:
    from imagescry_amd import normalize_per_channel

    g = cases.gen(3)
    x = torch.rand(6, 3, 32, 32, generator=g)
    means = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    stds = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    exp = transforms_oracle.normalize_per_channel(x, channel_means=means, channel_stds=stds)
    got = normalize_per_channel(x.to(device), channel_means=means.to(device), channel_stds=stds.to(device)).cpu()
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=0, atol=NORM_ATOL)
    # per-image statistics (#B = B) for the means only; stds computed from the batch
    pm = torch.rand(6, 3, 1, 1, generator=g)
    exp = transforms_oracle.normalize_per_channel(x, channel_means=pm, min_value=-2, max_value=2)
    got = normalize_per_channel(x.to(device), channel_means=pm.to(device), min_value=-2, max_value=2).cpu()
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=0, atol=NORM_ATOL)
    # float statistics path
    exp = transforms_oracle.normalize_per_channel(x * 255.0)
    got = normalize_per_channel((x * 255.0).to(device)).cpu()
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=0, atol=NORM_ATOL)


@pytest.mark.parametrize("add_batch", [False, True])
@pytest.mark.parametrize("output_size", [(4, 4), (5, 5), (5, 7), (7, 5), (33, 38)])
def test_resize_exact_output_size(output_size: tuple[int, int], add_batch: bool, device: torch.device) -> None:
    """reference tests/test_image/test_transform.py:29-49, plus value parity with the oracle."""
    from imagescry_amd import resize

    img = cases.reference_test_image()
    if add_batch:
        img = img.unsqueeze(0)
    got = resize(img.to(device), output_size=output_size, side_ref="height").cpu()
    assert got.shape[-2:] == output_size and got.dtype.is_floating_point and got.ndim == img.ndim
    exp = transforms_oracle.resize(img, output_size=output_size, side_ref="height")
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=RESIZE_RTOL, atol=RESIZE_ATOL)


@pytest.mark.parametrize("transpose_input", [False, True])
@pytest.mark.parametrize("side_ref", ["height", "width", "long", "short"])
@pytest.mark.parametrize("output_size", [16, 31, 46])
def test_resize_side_ref(output_size: int, side_ref: str, transpose_input: bool, device: torch.device) -> None:
    """reference tests/test_image/test_transform.py:52-104: shapes, plus values against the oracle."""
    from imagescry_amd import resize

    img = cases.reference_test_image()
    if transpose_input:
        img = img.transpose(-2, -1)
    exp = transforms_oracle.resize(img, output_size, side_ref=side_ref)
    got = resize(img.to(device), output_size, side_ref=side_ref).cpu()
    assert got.shape == exp.shape
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=RESIZE_RTOL, atol=RESIZE_ATOL)


def test_resize_large_and_float_and_golden(device: torch.device) -> None:
    from imagescry_amd import resize

    big = cases.images_u8((2, 3, 80, 100), seed=cases.SEED + 1)
    got = resize(big.to(device), 64, side_ref="long").cpu()
    np.testing.assert_allclose(got.numpy(), GOLDEN["big_resized"], rtol=RESIZE_RTOL, atol=RESIZE_ATOL)
    hd = cases.images_u8((1, 3, 1080, 1920), seed=8)
    exp = transforms_oracle.resize(hd, 640, side_ref="long")
    got = resize(hd.to(device), 640, side_ref="long").cpu()
    assert got.shape == (1, 3, 360, 640)
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=RESIZE_RTOL, atol=RESIZE_ATOL)
    f = torch.randn(2, 3, 37, 53, generator=cases.gen(4))
    exp = transforms_oracle.resize(f, (80, 21))  # upsample one side, downsample the other
    got = resize(f.to(device), (80, 21)).cpu()
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=RESIZE_RTOL, atol=RESIZE_ATOL)
    exp2 = transforms_oracle.resize(f[0, 0], 20, side_ref="short")  # 2-D input keeps its rank
    got2 = resize(f[0, 0].to(device), 20, side_ref="short").cpu()
    assert got2.shape == exp2.shape and got2.ndim == 2
    np.testing.assert_allclose(got2.numpy(), exp2.numpy(), rtol=RESIZE_RTOL, atol=RESIZE_ATOL)


def test_l2norm_channels(device: torch.device) -> None:
    from imagescry_amd import _lib

    golden = np.load(Path(__file__).resolve().parent / "golden" / "l2norm.npz")["out"]
    x = torch.randn(3, 128, 7, 10, generator=cases.gen())
    cases_ = [(x, golden), (torch.randn(5, 768, 1, 1, generator=cases.gen(6)), None)]
    z = torch.randn(2, 64, 3, 3, generator=cases.gen(7))
    z[0, :, 1, 1] = 0  # zero vector stays zero (eps clamp), as F.normalize
    cases_.append((z, None))
    lib = _lib.load()
    for t, gold in cases_:
        exp = encoder_oracle.l2_normalize_channels(t).numpy() if gold is None else gold
        xd = t.to(device).contiguous()
        y = torch.empty_like(xd)
        b, e, h, w = t.shape
        st = lib.isc_l2norm_channels(xd.data_ptr(), b, e, h * w, 1e-12, y.data_ptr(), _lib.stream_handle(device))
        _lib.check(st, "isc_l2norm_channels")
        np.testing.assert_allclose(y.cpu().numpy(), exp, rtol=0, atol=1e-6)
