"""Image transforms on MI355X: bilinear resize and batch-statistics normalisation.

Same names, arguments and error behaviour as the reference module
(src/imagescry/image/transforms.py:16-197); the arithmetic runs in the HIP kernels of
`csrc/preprocess.hip` through the C ABI (`isc_channel_stats`, `isc_normalize_clip`,
`isc_resize_bilinear`).  Inputs must live on a HIP device -- there is no CPU path.
"""

from __future__ import annotations

import math
from typing import Literal

import torch
from torch import Tensor

from imagescry_amd import _lib

__all__ = ["normalize_per_channel", "resize", "to_4d"]

SideRef = Literal["height", "width", "long", "short"]


def to_4d(image_tensor: Tensor) -> Tensor:
    """Add leading phantom dimensions until the tensor is `[B, C, H, W]` (reference: transforms.py:130-164)."""
    if image_tensor.ndim == 2:
        return image_tensor.unsqueeze(0).unsqueeze(0)
    if image_tensor.ndim == 3:
        return image_tensor.unsqueeze(0)
    if image_tensor.ndim == 4:
        return image_tensor
    raise ValueError(f"Invalid image tensor shape: {image_tensor.shape}")


def _calc_scale_factor(height: int, width: int, output_size: int, side_ref: str) -> float:
    """reference: transforms.py:168-197."""
    if side_ref == "height":
        return output_size / height
    if side_ref == "width":
        return output_size / width
    if side_ref == "long":
        return output_size / max(height, width)
    if side_ref == "short":
        return output_size / min(height, width)
    raise ValueError(f"Invalid side_ref: {side_ref}")


def _kernel_input(image_tensor: Tensor, name: str) -> Tensor:
    """uint8 and float32 are read directly by the kernels; any other numeric dtype is cast to float32 first,
    which is what the reference's `.float()` does (transforms.py:59,103)."""
    if not isinstance(image_tensor, Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(image_tensor).__name__}")
    if image_tensor.dtype == torch.bool or image_tensor.is_complex():
        raise TypeError(f"{name} must be a real numeric tensor, got {image_tensor.dtype}")
    _lib.require_device(image_tensor, name)
    if image_tensor.dtype not in (torch.uint8, torch.float32):
        image_tensor = image_tensor.float()
    return image_tensor.contiguous()


def resize(image_tensor: Tensor, output_size: int | tuple[int, int], *, side_ref: SideRef = "long") -> Tensor:
    """Bilinear resize (`align_corners=False`, no antialias) to float32 (reference: transforms.py:78-126).

    `output_size` as an int fixes the side named by `side_ref` and scales the other proportionally, the output
    being `floor(side * scale)` as `F.interpolate(..., recompute_scale_factor=True)` computes it; a tuple is exact.
    """
    if image_tensor.ndim < 2 or image_tensor.ndim > 4:
        raise ValueError(f"Invalid image tensor shape: {image_tensor.shape}")
    squeeze = 4 - image_tensor.ndim
    x = _kernel_input(to_4d(image_tensor), "image_tensor")
    b, c, h1, w1 = x.shape
    if isinstance(output_size, int):
        scale = _calc_scale_factor(h1, w1, output_size, side_ref)
        h2, w2 = math.floor(h1 * scale), math.floor(w1 * scale)
    else:
        h2, w2 = (int(v) for v in output_size)
    if h2 <= 0 or w2 <= 0:
        raise ValueError(f"Input and output sizes should be greater than 0, got output ({h2}, {w2})")
    y = torch.empty((b, c, h2, w2), dtype=torch.float32, device=x.device)
    if b * c > 0:
        lib = _lib.load()
        with torch.cuda.device(x.device):
            st = lib.isc_resize_bilinear(
                x.data_ptr(), _lib.dtype_code(x.dtype), b * c, h1, w1, h2, w2, y.data_ptr(), _lib.stream_handle(x.device)
            )
        _lib.check(st, "isc_resize_bilinear")
    for _ in range(squeeze):
        y = y.squeeze(0)
    return y


def _channel_stats(x: Tensor) -> tuple[Tensor, Tensor]:
    """Batch-wide per-channel mean and unbiased std -> two float32 `[C]` tensors (reference: transforms.py:62-65)."""
    b, c, h, w = x.shape
    lib = _lib.load()
    code = _lib.dtype_code(x.dtype)
    need = _lib.c_size_t()
    _lib.check(lib.isc_channel_stats_workspace_bytes(code, b, c, h, w, need), "isc_channel_stats_workspace_bytes")
    ws = torch.empty(need.value, dtype=torch.uint8, device=x.device)
    stats = torch.empty((2, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        st = lib.isc_channel_stats(
            x.data_ptr(), code, b, c, h, w, stats[0].data_ptr(), stats[1].data_ptr(), ws.data_ptr(), need.value,
            _lib.stream_handle(x.device),
        )
    _lib.check(st, "isc_channel_stats")
    return stats[0], stats[1]


def _stat_arg(t: Tensor, name: str, b: int, c: int, device: torch.device) -> tuple[Tensor, int]:
    """Validate a caller-supplied `#B C 1 1` statistic and flatten it to `[stat_batch * C]`."""
    if not isinstance(t, Tensor) or not t.dtype.is_floating_point:
        raise TypeError(f"{name} must be a floating point tensor")
    if t.ndim != 4 or t.shape[1] != c or t.shape[2:] != (1, 1) or t.shape[0] not in (1, b):
        raise ValueError(f"{name} must have shape [1 or {b}, {c}, 1, 1], got {tuple(t.shape)}")
    return t.to(device=device, dtype=torch.float32).reshape(-1).contiguous(), t.shape[0]


def normalize_per_channel(
    image_tensor: Tensor,
    *,
    channel_means: Tensor | None = None,
    channel_stds: Tensor | None = None,
    min_value: float | None = None,
    max_value: float | None = None,
    eps: float = 1e-6,
    _nhwc4: bool = False,
) -> Tensor:
    """`clip((x - mean_c) / (std_c + eps), min_value, max_value)` as float32 (reference: transforms.py:16-74).

    Without caller-supplied statistics the mean and the UNBIASED standard deviation are taken per channel over
    the whole batch (dims 0, 2, 3), so the result depends on the batch composition, exactly as in the reference.

    `_nhwc4` (not part of the reference's signature; used by `predict_step`): write the same values channels-last as
    `[B, H, W, 4]` with the channels past C zero -- the layout the convolution stems read (C <= 4).
    """
    if not isinstance(image_tensor, Tensor):
        raise TypeError(f"image_tensor must be a torch.Tensor, got {type(image_tensor).__name__}")
    if image_tensor.ndim != 4:
        raise ValueError(f"image_tensor must have shape [B, C, H, W], got {tuple(image_tensor.shape)}")
    x = _kernel_input(image_tensor, "image_tensor")
    b, c, h, w = x.shape
    if _nhwc4 and c > 4:
        raise ValueError(f"the channels-last output holds at most 4 channels, got {c}")
    y = torch.empty((b, h, w, 4) if _nhwc4 else (b, c, h, w), dtype=torch.float32, device=x.device)
    if x.numel() == 0:
        return y
    mean = std = None
    mean_b = std_b = 1
    if channel_means is not None:
        mean, mean_b = _stat_arg(channel_means, "channel_means", b, c, x.device)
    if channel_stds is not None:
        std, std_b = _stat_arg(channel_stds, "channel_stds", b, c, x.device)
    if mean is None or std is None:
        auto_mean, auto_std = _channel_stats(x)
        mean = auto_mean if mean is None else mean
        std = auto_std if std is None else std
    stat_batch = max(mean_b, std_b)
    if stat_batch > 1:  # one of the two is per-image: broadcast the other one
        if mean_b == 1:
            mean = mean.repeat(stat_batch)
        if std_b == 1:
            std = std.repeat(stat_batch)
    lo = -math.inf if min_value is None else float(min_value)
    hi = math.inf if max_value is None else float(max_value)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        st = (lib.isc_normalize_clip_nhwc4 if _nhwc4 else lib.isc_normalize_clip)(
            x.data_ptr(), _lib.dtype_code(x.dtype), b, c, h, w, mean.data_ptr(), std.data_ptr(), stat_batch,
            float(eps), lo, hi, y.data_ptr(), _lib.stream_handle(x.device),
        )
    _lib.check(st, "isc_normalize_clip")
    return y
