"""ResNet-50 (v1.5: stride on the 3x3) -> 768-d encoder definition: weight naming, seeded init, BN folding.

BASELINE.json config 2 names "ResNet-50 (random weights) -> 768-d, fp32".  The reference has no ResNet (its
encoder is torchvision EfficientNetV2, src/imagescry/models/embedding.py:133-147); this file is the build's own
definition.  State dicts use torchvision's parameter names (`conv1.weight`, `layer1.0.bn2.running_var`,
`layer2.0.downsample.0.weight`, ...) so a real torchvision ResNet-50 checkpoint drops in; only `fc` differs
(2048 -> embedding_dim projection instead of the 1000-way classifier).
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor

STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))  # (bottleneck width, blocks, stride of first block)
EXPANSION = 4
BN_EPS = 1e-5
STEM_TAPS = 56  # 7 * 7 = 49 filter taps padded to a multiple of 8 (one K step of the stem = 8 taps x 4 channels)


def _kaiming_normal_fan_out(shape: tuple[int, int, int, int], g: torch.Generator) -> Tensor:
    cout, _cin, r, s = shape
    std = math.sqrt(2.0 / (cout * r * s))
    return torch.randn(shape, generator=g) * std


def make_state_dict(*, embedding_dim: int = 768, seed: int = 0, randomize_bn: bool = False) -> dict[str, Tensor]:
    """Seeded random ResNet-50 parameters (CPU float32).  Conv: kaiming-normal (fan_out, ReLU); BatchNorm: identity
    statistics, or -- with `randomize_bn` -- random affine and running statistics so that BN folding is exercised;
    `fc`: torch.nn.Linear's default uniform(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
    g = torch.Generator().manual_seed(seed)
    sd: dict[str, Tensor] = {}

    def conv(name: str, cout: int, cin: int, k: int) -> None:
        sd[f"{name}.weight"] = _kaiming_normal_fan_out((cout, cin, k, k), g)

    def bn(name: str, c: int) -> None:
        if randomize_bn:
            sd[f"{name}.weight"] = torch.rand(c, generator=g) * 0.5 + 0.5
            sd[f"{name}.bias"] = torch.randn(c, generator=g) * 0.1
            sd[f"{name}.running_mean"] = torch.randn(c, generator=g) * 0.1
            sd[f"{name}.running_var"] = torch.rand(c, generator=g) + 0.5
        else:
            sd[f"{name}.weight"] = torch.ones(c)
            sd[f"{name}.bias"] = torch.zeros(c)
            sd[f"{name}.running_mean"] = torch.zeros(c)
            sd[f"{name}.running_var"] = torch.ones(c)

    conv("conv1", 64, 3, 7)
    bn("bn1", 64)
    inplanes = 64
    for li, (planes, blocks, stride) in enumerate(STAGES, start=1):
        for bi in range(blocks):
            p = f"layer{li}.{bi}"
            conv(f"{p}.conv1", planes, inplanes, 1)
            bn(f"{p}.bn1", planes)
            conv(f"{p}.conv2", planes, planes, 3)
            bn(f"{p}.bn2", planes)
            conv(f"{p}.conv3", planes * EXPANSION, planes, 1)
            bn(f"{p}.bn3", planes * EXPANSION)
            if bi == 0 and (stride != 1 or inplanes != planes * EXPANSION):
                conv(f"{p}.downsample.0", planes * EXPANSION, inplanes, 1)
                bn(f"{p}.downsample.1", planes * EXPANSION)
            inplanes = planes * EXPANSION
    bound = 1.0 / math.sqrt(inplanes)
    sd["fc.weight"] = (torch.rand(embedding_dim, inplanes, generator=g) * 2 - 1) * bound
    sd["fc.bias"] = (torch.rand(embedding_dim, generator=g) * 2 - 1) * bound
    return sd


@dataclass
class FoldedConv:
    """One convolution with its BatchNorm folded in, laid out for `isc_conv2d_nhwc`."""

    weight: Tensor  # float32 [Cout, R, S, Cin] (KRSC), or [Cout, Kpad] for the im2col'd stem
    bias: Tensor  # float32 [Cout]
    kernel: int
    stride: int
    pad: int

    def to(self, device: torch.device | str) -> "FoldedConv":
        return FoldedConv(self.weight.to(device), self.bias.to(device), self.kernel, self.stride, self.pad)


def fold_conv_bn(sd: dict[str, Tensor], conv: str, bn: str, stride: int, pad: int) -> FoldedConv:
    """`bn(conv(x))` in eval mode == `conv'(x) + b'` with `w' = w * gamma / sqrt(var + eps)` per output channel and
    `b' = beta - mean * gamma / sqrt(var + eps)`; folded in float64, stored as float32 KRSC."""
    w = sd[f"{conv}.weight"].double()
    scale = sd[f"{bn}.weight"].double() / torch.sqrt(sd[f"{bn}.running_var"].double() + BN_EPS)
    bias = sd[f"{bn}.bias"].double() - sd[f"{bn}.running_mean"].double() * scale
    w = (w * scale[:, None, None, None]).permute(0, 2, 3, 1).contiguous()  # OIHW -> KRSC
    return FoldedConv(w.float(), bias.float(), int(w.shape[1]), stride, pad)


# Stages whose first block runs its projection shortcut INSIDE conv3 (`isc_conv2d_nhwc_dual`: the two 1 x 1 convolutions
# K-concatenated, `relu(w3 . h + wd . x[:, ::s, ::s] + b3 + bd)`), so the shortcut's [B, H, W, 4 * planes] map is never
# written and read back as a residual.  Stage 1 gains most -- the map is largest there (1.6 GB at batch 512 / 224 px) and both
# convolutions are memory-bound: 660 + 834 us -> 1 056 us in one launch; in stages 2 - 4 the same matrix work moves into a
# conv3 three to seven times as long and the saving is the map's round trip and a launch.  Same-device A/B of the whole step
# (gpurun_out/r4/call6.out, batch 512): no stage 41.2 ms, stage 1 40.75, stages 1 - 2 40.55, 1 - 3 40.45, all four 40.35.
FUSED_SHORTCUT_STAGES = (1, 2, 3, 4)


@dataclass
class Bottleneck:
    conv1: FoldedConv
    conv2: FoldedConv
    conv3: FoldedConv
    downsample: FoldedConv | None
    fused: FoldedConv | None = None  # conv3 and the shortcut side by side: weight [Cout, Cin3 + Cin_shortcut], bias b3 + bd

    def to(self, device: torch.device | str) -> "Bottleneck":
        return Bottleneck(
            self.conv1.to(device), self.conv2.to(device), self.conv3.to(device),
            None if self.downsample is None else self.downsample.to(device),
            None if self.fused is None else self.fused.to(device),
        )


@dataclass
class FoldedResNet50:
    stem: FoldedConv  # weight [64, STEM_TAPS, 4]: taps (r, s) x (R, G, B, 0)
    blocks: list[Bottleneck]
    fc: FoldedConv  # weight [E, 1, 1, 2048]

    def to(self, device: torch.device | str) -> "FoldedResNet50":
        return FoldedResNet50(self.stem.to(device), [b.to(device) for b in self.blocks], self.fc.to(device))


def fold_state_dict(sd: dict[str, Tensor]) -> FoldedResNet50:
    stem = fold_conv_bn(sd, "conv1", "bn1", stride=2, pad=3)
    w = stem.weight.reshape(stem.weight.shape[0], 49, 3)  # [64, taps (r, s), c]
    stem.weight = torch.nn.functional.pad(w, (0, 1, 0, STEM_TAPS - 49)).contiguous()  # channel 3 and taps 49.. are zero
    blocks: list[Bottleneck] = []
    for li, (_planes, nblocks, stride) in enumerate(STAGES, start=1):
        for bi in range(nblocks):
            p = f"layer{li}.{bi}"
            s = stride if bi == 0 else 1
            ds = None
            if f"{p}.downsample.0.weight" in sd:
                ds = fold_conv_bn(sd, f"{p}.downsample.0", f"{p}.downsample.1", stride=s, pad=0)
            conv3 = fold_conv_bn(sd, f"{p}.conv3", f"{p}.bn3", stride=1, pad=0)
            fused = None
            if ds is not None and li in FUSED_SHORTCUT_STAGES:
                cout = conv3.weight.shape[0]
                wcat = torch.cat([conv3.weight.reshape(cout, -1), ds.weight.reshape(cout, -1)], dim=1).contiguous()
                fused = FoldedConv(wcat, (conv3.bias.double() + ds.bias.double()).float(), 1, s, 0)  # stride = the shortcut's
            blocks.append(
                Bottleneck(
                    conv1=fold_conv_bn(sd, f"{p}.conv1", f"{p}.bn1", stride=1, pad=0),
                    conv2=fold_conv_bn(sd, f"{p}.conv2", f"{p}.bn2", stride=s, pad=1),
                    conv3=conv3,
                    downsample=ds,
                    fused=fused,
                )
            )
    fcw = sd["fc.weight"].float()
    fc = FoldedConv(fcw.reshape(fcw.shape[0], 1, 1, fcw.shape[1]).contiguous(), sd["fc.bias"].float().contiguous(), 1, 1, 0)
    return FoldedResNet50(stem, blocks, fc)


def conv_flops(batch: int, height: int, width: int, embedding_dim: int = 768) -> int:
    """Multiply-add FLOPs (2 per MAC) of one forward pass, from the layer table; used for the MFMA roofline."""
    def out(n: int, k: int, s: int, p: int) -> int:
        return (n + 2 * p - k) // s + 1

    h, w = out(height, 7, 2, 3), out(width, 7, 2, 3)
    total = 2 * batch * h * w * 64 * 147
    h, w = out(h, 3, 2, 1), out(w, 3, 2, 1)
    inplanes = 64
    for planes, nblocks, stride in STAGES:
        for bi in range(nblocks):
            s = stride if bi == 0 else 1
            total += 2 * batch * h * w * planes * inplanes  # conv1 (1x1, input resolution)
            h2, w2 = out(h, 3, s, 1), out(w, 3, s, 1)
            total += 2 * batch * h2 * w2 * planes * planes * 9  # conv2 (3x3, strided)
            total += 2 * batch * h2 * w2 * planes * EXPANSION * planes  # conv3
            if bi == 0 and (s != 1 or inplanes != planes * EXPANSION):
                total += 2 * batch * h2 * w2 * planes * EXPANSION * inplanes
            inplanes = planes * EXPANSION
            h, w = h2, w2
    total += 2 * batch * inplanes * embedding_dim
    return total


def conv_bytes(batch: int, height: int, width: int, embedding_dim: int = 768) -> int:
    """Algorithmic HBM bytes of the `k_conv_f32` launches of one forward pass: every convolution reads its float32 input
    map once (the stem its 4-channel NHWC input), its residual once where it has one, its weights once, and writes its
    output once.  What `roofline.traffic` is compared with (the measured L2 <-> fabric bytes re-read 3 x 3 inputs)."""
    def out(n: int, k: int, s: int, p: int) -> int:
        return (n + 2 * p - k) // s + 1

    h, w = out(height, 7, 2, 3), out(width, 7, 2, 3)
    total = 4 * (batch * height * width * 4 + batch * h * w * 64 + 64 * 7 * 7 * 4)
    h, w = out(h, 3, 2, 1), out(w, 3, 2, 1)
    inplanes = 64
    for planes, nblocks, stride in STAGES:
        for bi in range(nblocks):
            s = stride if bi == 0 else 1
            h2, w2 = out(h, 3, s, 1), out(w, 3, s, 1)
            pin, pout = batch * h * w, batch * h2 * w2
            total += 4 * (pin * inplanes + pin * planes + planes * inplanes)  # conv1
            total += 4 * (pin * planes + pout * planes + planes * planes * 9)  # conv2
            shortcut = bi == 0 and (s != 1 or inplanes != planes * EXPANSION)
            fused = shortcut and (STAGES.index((planes, nblocks, stride)) + 1) in FUSED_SHORTCUT_STAGES
            if fused:  # conv3 and the shortcut in one launch: both inputs and weights once, the output once, no residual
                total += 4 * (pout * planes + pout * inplanes + pout * planes * EXPANSION
                              + planes * EXPANSION * (planes + inplanes))
            else:
                total += 4 * (pout * planes + 2 * pout * planes * EXPANSION + planes * EXPANSION * planes)  # conv3 + residual
                if shortcut:  # downsample reads the strided input pixels
                    total += 4 * (pout * inplanes + pout * planes * EXPANSION + planes * EXPANSION * inplanes)
            inplanes = planes * EXPANSION
            h, w = h2, w2
    return total
