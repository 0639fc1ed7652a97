"""CPU restatement of the reference image transforms (test infrastructure, see oracle/__init__.py).

Every function performs the same torch calls, in the same order and with the same
arguments, as the reference function it names; the jaxtyping decorators are the only
thing left out.
"""

from __future__ import annotations

import torch
from torch import Tensor
from torch.nn.functional import interpolate


def to_4d(image_tensor: Tensor) -> Tensor:
    """reference: src/imagescry/image/transforms.py:130-164."""
    if image_tensor.ndim == 2:
        return image_tensor.unsqueeze(0).unsqueeze(0)
    if image_tensor.ndim == 3:
        return image_tensor.unsqueeze(0)
    if image_tensor.ndim == 4:
        return image_tensor
    raise ValueError(f"Invalid image tensor shape: {image_tensor.shape}")


def calc_scale_factor(height: int, width: int, output_size: int, side_ref: str) -> float:
    """reference: src/imagescry/image/transforms.py:168-197."""
    if side_ref == "height":
        return output_size / height
    if side_ref == "width":
        return output_size / width
    if side_ref == "long":
        return output_size / max(height, width)
    if side_ref == "short":
        return output_size / min(height, width)
    raise ValueError(f"Invalid side_ref: {side_ref}")


def resize(image_tensor: Tensor, output_size: int | tuple[int, int], *, side_ref: str = "long") -> Tensor:
    """reference: src/imagescry/image/transforms.py:78-126 (float cast, bilinear, align_corners=False)."""
    squeeze_dims = tuple(range(4 - image_tensor.ndim))
    image_tensor = to_4d(image_tensor)
    image_tensor = image_tensor.float()
    if isinstance(output_size, int):
        height, width = image_tensor.shape[-2:]
        scale_factor = calc_scale_factor(height, width, output_size, side_ref)
        image_tensor = interpolate(
            image_tensor,
            scale_factor=scale_factor,
            mode="bilinear",
            align_corners=False,
            recompute_scale_factor=True,
        )
    else:
        image_tensor = interpolate(image_tensor, size=output_size, mode="bilinear", align_corners=False)
    return image_tensor.squeeze(squeeze_dims)


def normalize_per_channel(
    image_tensor: Tensor,
    *,
    channel_means: Tensor | None = None,
    channel_stds: Tensor | None = None,
    min_value: float | None = None,
    max_value: float | None = None,
    eps: float = 1e-6,
) -> Tensor:
    """reference: src/imagescry/image/transforms.py:58-74 (batch-wide mean, unbiased std, sigma+eps, clip)."""
    image_tensor = image_tensor.float()
    if channel_means is None:
        channel_means = image_tensor.mean(dim=(0, 2, 3), keepdim=True)
    if channel_stds is None:
        channel_stds = image_tensor.std(dim=(0, 2, 3), keepdim=True)
    image_tensor = (image_tensor - channel_means) / (channel_stds + eps)
    if min_value is not None or max_value is not None:
        image_tensor = image_tensor.clip(min_value, max_value)
    return image_tensor


def channel_stats_f64(image_tensor: Tensor) -> tuple[Tensor, Tensor]:
    """Batch-wide per-channel mean and unbiased std accumulated in float64.

    Not a reference function: a wide-accumulator cross-check for the statistics the
    reference computes in float32 at transforms.py:62-65.
    """
    x = image_tensor.double()
    return x.mean(dim=(0, 2, 3)), x.std(dim=(0, 2, 3))
