"""The CPU oracle for the transforms, pinned by the reference's own property tests and by the golden fixtures.

The reference holds no golden vectors; what its tests pin is re-run here against the
oracle with the reference's seeds and sizes:
  * tests/test_image/test_transform.py:14-24   normalize -> per-channel mean ~ 0, std ~ 1 at atol 1e-4
  * tests/test_image/test_transform.py:29-49   resize to an exact (h, w), dtype floating
  * tests/test_image/test_transform.py:52-104  resize by one reference side, other side within +-1
  * src/imagescry/image/transforms.py:142-155  to_4d doctest shapes
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import cases  # noqa: E402

from oracle import encoder_oracle, transforms_oracle  # noqa: E402

GOLDEN = np.load(Path(__file__).resolve().parent / "golden" / "preprocess.npz")


def test_normalize_per_channel_property() -> None:
    image = cases.reference_test_image()
    out = transforms_oracle.normalize_per_channel(image.float().unsqueeze(0))
    means, stds = out.mean((-2, -1)), out.std((-2, -1))
    assert torch.allclose(torch.zeros_like(means), means, atol=1e-4)
    assert torch.allclose(torch.ones_like(means), stds, atol=1e-4)


@pytest.mark.parametrize("add_batch", [False, True])
@pytest.mark.parametrize("output_size", [(4, 4), (5, 5), (5, 7), (7, 5), (33, 38)])
def test_resize_exact_output_size(output_size: tuple[int, int], add_batch: bool) -> None:
    image = cases.reference_test_image()
    if add_batch:
        image = image.unsqueeze(0)
    out = transforms_oracle.resize(image, output_size=output_size, side_ref="height")
    assert out.shape[-2:] == output_size
    assert out.dtype.is_floating_point


@pytest.mark.parametrize("transpose_input", [False, True])
@pytest.mark.parametrize("side_ref", ["height", "width", "long", "short"])
@pytest.mark.parametrize("output_size", [16, 31, 46])
def test_resize_side_ref(output_size: int, side_ref: str, transpose_input: bool) -> None:
    image = cases.reference_test_image()
    if transpose_input:
        image = image.transpose(-2, -1)
    h0, w0 = image.shape[-2:]
    h1, w1 = transforms_oracle.resize(image, output_size, side_ref=side_ref).shape[-2:]
    height_is_ref = (
        side_ref == "height" or (side_ref == "long" and h0 >= w0) or (side_ref == "short" and h0 < w0)
    )
    if height_is_ref:
        assert h1 == output_size
        assert w1 == pytest.approx(w0 * output_size / h0, abs=1)
    else:
        assert w1 == output_size
        assert h1 == pytest.approx(h0 * output_size / w0, abs=1)


def test_survey_probes() -> None:
    """Output sizes probed in SURVEY.md section 8 row a3: floor(side * scale)."""
    assert transforms_oracle.resize(torch.zeros(3, 30, 45), 16, side_ref="long").shape == (3, 10, 16)
    assert transforms_oracle.resize(torch.zeros(1, 1080, 1920), 640, side_ref="long").shape == (1, 360, 640)


def test_to_4d_shapes() -> None:
    assert transforms_oracle.to_4d(torch.randn(3, 4)).shape == (1, 1, 3, 4)
    assert transforms_oracle.to_4d(torch.randn(3, 5, 7)).shape == (1, 3, 5, 7)
    assert transforms_oracle.to_4d(torch.randn(16, 3, 5, 7)).shape == (16, 3, 5, 7)
    with pytest.raises(ValueError):
        transforms_oracle.to_4d(torch.randn(2, 2, 2, 2, 2))


def test_oracle_reproduces_golden_fixtures() -> None:
    torch.set_num_threads(1)
    small = cases.images_u8((4, 3, 30, 45))
    big = cases.images_u8((2, 3, 80, 100), seed=cases.SEED + 1)
    ref = cases.reference_test_image()
    np.testing.assert_allclose(encoder_oracle.preprocess(small, 640).numpy(), GOLDEN["small_out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(
        transforms_oracle.resize(big, 64, side_ref="long").numpy(), GOLDEN["big_resized"], rtol=1e-6, atol=1e-5
    )
    np.testing.assert_allclose(encoder_oracle.preprocess(big, 64).numpy(), GOLDEN["big_out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(
        transforms_oracle.resize(ref, (5, 7)).numpy(), GOLDEN["ref_resize_5x7"], rtol=1e-6, atol=1e-5
    )
    assert GOLDEN["big_resized"].shape == (2, 3, 51, 64)  # floor(80 * 64 / 100) = 51
    assert float(np.abs(GOLDEN["small_out"]).max()) <= 3.0  # clip(-3, 3)


def test_statistics_are_batch_wide_and_unbiased() -> None:
    """The two facts of transforms.py:62-65 that shape the kernels: statistics span the whole batch and the
    std is the unbiased (n - 1) one, in a `sigma + eps` denominator."""
    x = cases.images_u8((4, 3, 30, 45))
    out = transforms_oracle.normalize_per_channel(x)
    xf = x.double()
    n = 4 * 30 * 45
    mean = xf.mean(dim=(0, 2, 3), keepdim=True)
    std = ((xf - mean).pow(2).sum(dim=(0, 2, 3), keepdim=True) / (n - 1)).sqrt()
    exp = ((xf - mean) / (std + 1e-6)).float()
    np.testing.assert_allclose(out.numpy(), exp.numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(GOLDEN["small_mean"], mean.flatten().float().numpy(), rtol=1e-6)
    np.testing.assert_allclose(GOLDEN["small_std"], std.flatten().float().numpy(), rtol=1e-6)
    # a different batch composition changes the result (so a batch must never be split across GPUs)
    half = transforms_oracle.normalize_per_channel(x[:2])
    assert not torch.allclose(half, out[:2])


def test_l2_normalize_channels_golden_and_zero_vector() -> None:
    golden = np.load(Path(__file__).resolve().parent / "golden" / "l2norm.npz")["out"]
    x = torch.randn(3, 128, 7, 10, generator=cases.gen())
    out = encoder_oracle.l2_normalize_channels(x)
    np.testing.assert_allclose(out.numpy(), golden, rtol=0, atol=1e-7)
    assert torch.allclose(out.pow(2).sum(dim=1), torch.ones(3, 7, 10), atol=1e-5)
    assert float(encoder_oracle.l2_normalize_channels(torch.zeros(1, 8, 2, 2)).abs().max()) == 0.0
