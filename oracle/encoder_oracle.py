"""CPU restatement of the encode half of the hot path (test infrastructure, see oracle/__init__.py).

`preprocess` / `predict_step_embeddings` follow the reference call order.  The
ResNet-50 -> 768-d encoder named by BASELINE.json's configs does not exist in the
reference (its only concrete encoder is torchvision EfficientNetV2,
src/imagescry/models/embedding.py:133-147); `resnet50_forward` is the build's own
definition written with plain `torch.nn.functional` calls on an un-fused,
torchvision-named state dict (conv / BatchNorm(eval) / ReLU kept separate) so that
it checks the product's BN folding, weight re-layout and kernels independently.
Encoder values: parity unpinned by the reference.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor

from oracle.transforms_oracle import normalize_per_channel, resize

RESNET50_STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # (planes, blocks, stride)
BN_EPS = 1e-5


def preprocess(images: Tensor, max_side_length: int = 640) -> Tensor:
    """reference: src/imagescry/models/embedding.py:149-165."""
    h, w = images.shape[-2:]
    if max(h, w) > max_side_length:
        images = resize(images, output_size=max_side_length, side_ref="long")
    return normalize_per_channel(images, min_value=-3, max_value=3)


def l2_normalize_channels(x: Tensor) -> Tensor:
    """reference: src/imagescry/models/embedding.py:74 (`F.normalize(x, p=2, dim=1)`, eps=1e-12)."""
    return F.normalize(x, p=2, dim=1)


def _bn(x: Tensor, sd: dict[str, Tensor], prefix: str) -> Tensor:
    return F.batch_norm(
        x,
        sd[f"{prefix}.running_mean"],
        sd[f"{prefix}.running_var"],
        sd[f"{prefix}.weight"],
        sd[f"{prefix}.bias"],
        training=False,
        eps=BN_EPS,
    )


def resnet50_forward(x: Tensor, sd: dict[str, Tensor]) -> Tensor:
    """ResNet-50 v1.5 trunk (stride on the 3x3) -> global average pool -> `fc` 2048->E; returns `[B, E, 1, 1]`."""
    x = F.relu(_bn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), sd, "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, (_planes, blocks, stride) in enumerate(RESNET50_STAGES, start=1):
        for bi in range(blocks):
            p = f"layer{li}.{bi}"
            s = stride if bi == 0 else 1
            identity = x
            out = F.relu(_bn(F.conv2d(x, sd[f"{p}.conv1.weight"]), sd, f"{p}.bn1"))
            out = F.relu(_bn(F.conv2d(out, sd[f"{p}.conv2.weight"], stride=s, padding=1), sd, f"{p}.bn2"))
            out = _bn(F.conv2d(out, sd[f"{p}.conv3.weight"]), sd, f"{p}.bn3")
            if f"{p}.downsample.0.weight" in sd:
                identity = _bn(F.conv2d(x, sd[f"{p}.downsample.0.weight"], stride=s), sd, f"{p}.downsample.1")
            x = F.relu(out + identity)
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    x = F.linear(x, sd["fc.weight"], sd["fc.bias"])
    return x[:, :, None, None]


def predict_step_embeddings(images: Tensor, sd: dict[str, Tensor], max_side_length: int = 640) -> Tensor:
    """preprocess -> forward -> L2-normalize over channels (reference: embedding.py:57-76)."""
    with torch.no_grad():
        x = preprocess(images, max_side_length)
        x = resnet50_forward(x, sd)
        return l2_normalize_channels(x)
