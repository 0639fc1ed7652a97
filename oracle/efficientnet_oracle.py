"""CPU restatement of torchvision's EfficientNetV2 `features` stack (test infrastructure, see oracle/__init__.py).

The reference calls `torchvision.models.efficientnet_v2_{s,m,l}(weights).features`
(src/imagescry/models/embedding.py:133-147).  torchvision 0.23.0 (the reference's uv.lock pin) is not installed
here and there is no network, so this follows the published architecture: Conv2dNormActivation stem (3x3, stride 2,
BatchNorm eps 1e-3, SiLU); FusedMBConv = 3x3 conv-BN-SiLU [+ 1x1 conv-BN when expand != 1]; MBConv = 1x1
conv-BN-SiLU, 3x3 depthwise conv-BN-SiLU, SqueezeExcitation (avg-pool, fc1, SiLU, fc2, sigmoid, scale), 1x1 conv-BN;
`result += input` when stride == 1 and in == out (stochastic depth is the identity in eval mode); 1x1 conv-BN-SiLU
head to 1280 channels.  It is written with plain `torch.nn.functional` calls on an un-fused, torchvision-named
state dict, independently of the product's BN folding / channel padding / kernels.

PARITY UNPINNED by the reference beyond the output shape `(B, 1280, ceil(H/32), ceil(W/32))`
(tests/test_models/test_embedding.py:97-106), which tests/test_gpu_efficientnet.py re-runs.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor

BN_EPS = 1e-3


def _bn(x: Tensor, sd: dict[str, Tensor], p: str) -> Tensor:
    return F.batch_norm(x, sd[f"{p}.running_mean"], sd[f"{p}.running_var"], sd[f"{p}.weight"], sd[f"{p}.bias"],
                        training=False, eps=BN_EPS)


def _cna(x: Tensor, sd: dict[str, Tensor], p: str, *, stride: int = 1, groups: int = 1, act: bool = True) -> Tensor:
    """Conv2dNormActivation: conv (no bias, 'same' padding for odd kernels) -> BatchNorm -> SiLU."""
    w = sd[f"{p}.0.weight"]
    x = _bn(F.conv2d(x, w, stride=stride, padding=w.shape[-1] // 2, groups=groups), sd, f"{p}.1")
    return F.silu(x) if act else x


def features(x: Tensor, sd: dict[str, Tensor], stages: list[list[tuple[str, int, int, int, int]]]) -> Tensor:
    """`stages[i][j] = (kind, expand, stride, cin, cout)` for block j of stage i + 1."""
    x = _cna(x, sd, "features.0", stride=2)
    for si, blocks in enumerate(stages, start=1):
        for bi, (kind, expand, stride, cin, cout) in enumerate(blocks):
            p = f"features.{si}.{bi}.block"
            inp = x
            if kind == "fused":
                if expand == 1:
                    y = _cna(x, sd, f"{p}.0", stride=stride)
                else:
                    y = _cna(x, sd, f"{p}.0", stride=stride)
                    y = _cna(y, sd, f"{p}.1", act=False)
            else:
                y = _cna(x, sd, f"{p}.0")
                y = _cna(y, sd, f"{p}.1", stride=stride, groups=y.shape[1])
                s = F.adaptive_avg_pool2d(y, 1)
                s = F.silu(F.conv2d(s, sd[f"{p}.2.fc1.weight"], sd[f"{p}.2.fc1.bias"]))
                s = torch.sigmoid(F.conv2d(s, sd[f"{p}.2.fc2.weight"], sd[f"{p}.2.fc2.bias"]))
                y = _cna(y * s, sd, f"{p}.3", act=False)
            x = y + inp if (stride == 1 and cin == cout) else y
    return _cna(x, sd, f"features.{len(stages) + 1}")
