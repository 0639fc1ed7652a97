"""EfficientNetV2 feature extractor (the reference's actual backbone) on the HIP encoder blocks.

The reference builds `torchvision.models.efficientnet_v2_{s,m,l}(weights).features`
(src/imagescry/models/embedding.py:133-147).  torchvision is a third-party dependency that is absent from this
image (pinned 0.23.0 in the reference's uv.lock), so the architecture is restated here from its published
definition: a 3x3 / 2 stem, stages of FusedMBConv (3x3 expand + 1x1 project) and MBConv (1x1 expand, 3x3 depthwise,
squeeze-excitation, 1x1 project) blocks, a 1x1 head to 1280 channels; BatchNorm eps 1e-3, SiLU activations, residual
connections where stride is 1 and the channel count is unchanged; overall stride 32.  State dicts use torchvision's
`features.*` parameter names so that a torchvision checkpoint loads unchanged.

Values of this encoder are *parity unpinned* by the reference (its only test is the output shape,
tests/test_models/test_embedding.py:97-106); they are held to `oracle/efficientnet_oracle.py`.

Execution: NHWC float32 activations at their TRUE channel counts (all multiples of 4; a count that is not would be
zero-padded to the next one), BatchNorm folded into the convolutions; kernels: `isc_conv2d_nhwc` (3x3 and 1x1
convolutions, squeeze-excitation linears; inputs whose channel count is not a multiple of 32 -- the RGB stem, the 24-
and 48-channel stages -- run in its packed-K mode instead of being padded to 32),
`isc_dwconv2d_nhwc_pool` (depthwise + the SE mean), `isc_se_gate` (both SE linears), `isc_conv2d_nhwc_gated` (projection with the SE gate fused in).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
from torch import Tensor

from imagescry_amd import _lib

BN_EPS = 1e-3
LAST_CHANNELS = 1280

# (block type, expand ratio, kernel, stride, in channels, out channels, layers) -- torchvision's _efficientnet_conf
STAGES = {
    "s": [
        ("fused", 1, 3, 1, 24, 24, 2),
        ("fused", 4, 3, 2, 24, 48, 4),
        ("fused", 4, 3, 2, 48, 64, 4),
        ("mb", 4, 3, 2, 64, 128, 6),
        ("mb", 6, 3, 1, 128, 160, 9),
        ("mb", 6, 3, 2, 160, 256, 15),
    ],
    "m": [
        ("fused", 1, 3, 1, 24, 24, 3),
        ("fused", 4, 3, 2, 24, 48, 5),
        ("fused", 4, 3, 2, 48, 80, 5),
        ("mb", 4, 3, 2, 80, 160, 7),
        ("mb", 6, 3, 1, 160, 176, 14),
        ("mb", 6, 3, 2, 176, 304, 18),
        ("mb", 6, 3, 1, 304, 512, 5),
    ],
    "l": [
        ("fused", 1, 3, 1, 32, 32, 4),
        ("fused", 4, 3, 2, 32, 64, 7),
        ("fused", 4, 3, 2, 64, 96, 7),
        ("mb", 4, 3, 2, 96, 192, 10),
        ("mb", 6, 3, 1, 192, 224, 19),
        ("mb", 6, 3, 2, 224, 384, 25),
        ("mb", 6, 3, 1, 384, 640, 7),
    ],
}


def make_divisible(v: float, divisor: int = 8) -> int:
    """torchvision's `_make_divisible` (channel counts of the expanded blocks)."""
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


@dataclass(frozen=True)
class BlockSpec:
    kind: str  # "fused" | "mb"
    expand: int
    kernel: int
    stride: int
    cin: int
    cout: int

    @property
    def expanded(self) -> int:
        return make_divisible(self.cin * self.expand)

    @property
    def squeeze(self) -> int:
        return max(1, self.cin // 4)

    @property
    def residual(self) -> bool:
        return self.stride == 1 and self.cin == self.cout


def block_specs(size: str) -> list[list[BlockSpec]]:
    """Per stage, the list of its blocks (the first one carries the stride and the channel change)."""
    stages = []
    for kind, expand, kernel, stride, cin, cout, layers in STAGES[size]:
        blocks = []
        for i in range(layers):
            blocks.append(BlockSpec(kind, expand, kernel, stride if i == 0 else 1, cin if i == 0 else cout, cout))
        stages.append(blocks)
    return stages


def conv_flops(size: str, batch: int, height: int, width: int) -> int:
    """Multiply-add FLOPs (2 per MAC) of the DENSE convolutions of one forward pass (stem, expand / fused / project /
    head convolutions -- the launches of `k_conv_f32`), from the stage table with the true channel counts (no
    padding); depthwise convolutions and the squeeze-excitation linears (`k_se_gate`) are not matrix-core work and are
    left out.  For the MFMA roofline."""
    def down(n: int, s: int) -> int:
        return (n + s - 1) // s

    h, w = down(height, 2), down(width, 2)
    total = 2 * batch * h * w * STAGES[size][0][4] * 3 * 9  # stem 3x3 / 2
    last = STAGES[size][0][4]
    for blocks in block_specs(size):
        for b in blocks:
            h2, w2 = down(h, b.stride), down(w, b.stride)
            k2 = b.kernel * b.kernel
            if b.kind == "fused":
                if b.expand == 1:
                    total += 2 * batch * h2 * w2 * b.cout * b.cin * k2
                else:
                    total += 2 * batch * h2 * w2 * b.expanded * b.cin * k2 + 2 * batch * h2 * w2 * b.cout * b.expanded
            else:
                total += 2 * batch * h * w * b.expanded * b.cin  # expand 1x1 (input resolution)
                total += 2 * batch * h2 * w2 * b.cout * b.expanded  # project
            h, w, last = h2, w2, b.cout
    total += 2 * batch * h * w * LAST_CHANNELS * last  # head 1x1
    return total


def conv_bytes(size: str, batch: int, height: int, width: int) -> int:
    """Algorithmic HBM bytes of the `k_conv_f32` launches of one forward pass: every dense convolution reads its float32
    input once, its residual once where the block has one (stride 1, cin == cout), its weights once, writes its output
    once.  Depthwise sweeps and SE gates are other kernels and not counted.  Compared with `roofline.traffic`."""
    def down(n: int, s: int) -> int:
        return (n + s - 1) // s

    h, w = down(height, 2), down(width, 2)
    c0 = STAGES[size][0][4]
    total = 4 * (batch * height * width * 4 + batch * h * w * c0 + c0 * 27)
    last = c0
    for blocks in block_specs(size):
        for b in blocks:
            h2, w2 = down(h, b.stride), down(w, b.stride)
            k2 = b.kernel * b.kernel
            pin, pout = batch * h * w, batch * h2 * w2
            res = pout * b.cout if (b.stride == 1 and b.cin == b.cout) else 0
            if b.kind == "fused":
                if b.expand == 1:
                    total += 4 * (pin * b.cin + pout * b.cout + res + b.cout * b.cin * k2)
                else:
                    total += 4 * (pin * b.cin + pout * b.expanded + b.expanded * b.cin * k2)
                    total += 4 * (pout * b.expanded + pout * b.cout + res + b.cout * b.expanded)
            else:
                total += 4 * (pin * b.cin + pin * b.expanded + b.expanded * b.cin)  # expand 1x1
                total += 4 * (pout * b.expanded + pout * b.cout + res + b.cout * b.expanded)  # project
            h, w, last = h2, w2, b.cout
    total += 4 * (batch * h * w * last + batch * h * w * LAST_CHANNELS + LAST_CHANNELS * last)  # head 1x1
    return total


# ------------------------------------------------------------------------------------------- parameters
def make_state_dict(size: str = "s", *, seed: int = 0, randomize_bn: bool = False) -> dict[str, Tensor]:
    """Seeded random parameters with torchvision's names and init (conv: kaiming-normal fan_out; BatchNorm identity,
    or random affine / statistics with `randomize_bn`; SE biases zero)."""
    g = torch.Generator().manual_seed(seed)
    sd: dict[str, Tensor] = {}

    def conv(name: str, cout: int, cin_per_group: int, k: int) -> None:
        std = math.sqrt(2.0 / (cout * k * k))
        sd[f"{name}.weight"] = torch.randn((cout, cin_per_group, k, k), generator=g) * std

    def bn(name: str, c: int) -> None:
        if randomize_bn:
            sd[f"{name}.weight"] = torch.rand(c, generator=g) * 0.5 + 0.75
            sd[f"{name}.bias"] = torch.randn(c, generator=g) * 0.1
            sd[f"{name}.running_mean"] = torch.randn(c, generator=g) * 0.1
            sd[f"{name}.running_var"] = torch.rand(c, generator=g) * 0.5 + 0.75
        else:
            sd[f"{name}.weight"] = torch.ones(c)
            sd[f"{name}.bias"] = torch.zeros(c)
            sd[f"{name}.running_mean"] = torch.zeros(c)
            sd[f"{name}.running_var"] = torch.ones(c)

    stages = block_specs(size)
    stem = stages[0][0].cin
    conv("features.0.0", stem, 3, 3)
    bn("features.0.1", stem)
    for si, blocks in enumerate(stages, start=1):
        for bi, b in enumerate(blocks):
            p = f"features.{si}.{bi}.block"
            if b.kind == "fused":
                if b.expand == 1:
                    conv(f"{p}.0.0", b.cout, b.cin, b.kernel)
                    bn(f"{p}.0.1", b.cout)
                else:
                    conv(f"{p}.0.0", b.expanded, b.cin, b.kernel)
                    bn(f"{p}.0.1", b.expanded)
                    conv(f"{p}.1.0", b.cout, b.expanded, 1)
                    bn(f"{p}.1.1", b.cout)
            else:
                conv(f"{p}.0.0", b.expanded, b.cin, 1)
                bn(f"{p}.0.1", b.expanded)
                conv(f"{p}.1.0", b.expanded, 1, b.kernel)  # depthwise
                bn(f"{p}.1.1", b.expanded)
                conv(f"{p}.2.fc1", b.squeeze, b.expanded, 1)
                sd[f"{p}.2.fc1.bias"] = torch.zeros(b.squeeze) if not randomize_bn else torch.randn(b.squeeze, generator=g) * 0.1
                conv(f"{p}.2.fc2", b.expanded, b.squeeze, 1)
                sd[f"{p}.2.fc2.bias"] = torch.zeros(b.expanded) if not randomize_bn else torch.randn(b.expanded, generator=g) * 0.1
                conv(f"{p}.3.0", b.cout, b.expanded, 1)
                bn(f"{p}.3.1", b.cout)
    head = len(stages) + 1
    conv(f"features.{head}.0", LAST_CHANNELS, stages[-1][-1].cout, 1)
    bn(f"features.{head}.1", LAST_CHANNELS)
    return sd


def pad4(c: int) -> int:
    return (c + 3) // 4 * 4


def _krsc(w: Tensor, cout_pad: int, cin_pad: int) -> Tensor:
    """[Cout, Cin, k, k] -> the kernel's [Cout, K] rows, K = k * k * Cin rounded up to whole 32-float K steps (a no-op
    when Cin % 32 == 0; otherwise the packed-K layout of include/imagescry_hip.h)."""
    cout, cin, k, _ = w.shape
    wk = torch.zeros((cout_pad, k, k, cin_pad), dtype=torch.float32)
    wk[:cout, :, :, :cin] = w.permute(0, 2, 3, 1).float()
    flat = wk.reshape(cout_pad, k * k * cin_pad)
    kpad = (flat.shape[1] + 31) // 32 * 32
    if kpad != flat.shape[1]:
        flat = torch.cat([flat, torch.zeros((cout_pad, kpad - flat.shape[1]), dtype=torch.float32)], dim=1)
    return flat.contiguous()


@dataclass
class Conv:
    """A convolution with BatchNorm folded in, weights as the kernel's [Cout, K] rows (`_krsc`)."""

    weight: Tensor
    bias: Tensor
    kernel: int
    stride: int
    pad: int
    cout: int  # output channels (a multiple of 4)

    def to(self, device: torch.device | str) -> "Conv":
        return Conv(self.weight.to(device), self.bias.to(device), self.kernel, self.stride, self.pad, self.cout)


@dataclass
class Block:
    spec: BlockSpec
    convs: dict[str, Conv] = field(default_factory=dict)

    def to(self, device: torch.device | str) -> "Block":
        return Block(self.spec, {k: v.to(device) for k, v in self.convs.items()})


@dataclass
class FoldedEfficientNet:
    stem: Conv
    blocks: list[Block]
    head: Conv

    def to(self, device: torch.device | str) -> "FoldedEfficientNet":
        return FoldedEfficientNet(self.stem.to(device), [b.to(device) for b in self.blocks], self.head.to(device))


def _bn_scale_shift(sd: dict[str, Tensor], bn: str) -> tuple[Tensor, Tensor]:
    scale = sd[f"{bn}.weight"].double() / torch.sqrt(sd[f"{bn}.running_var"].double() + BN_EPS)
    shift = sd[f"{bn}.bias"].double() - sd[f"{bn}.running_mean"].double() * scale
    return scale, shift


def _fold(sd: dict[str, Tensor], conv: str, bn: str | None, stride: int) -> Conv:
    w = sd[f"{conv}.weight"].double()
    cout, cin, k, _ = w.shape
    if bn is not None:
        scale, shift = _bn_scale_shift(sd, bn)
        w = w * scale[:, None, None, None]
    else:
        shift = sd[f"{conv}.bias"].double()
    cop = pad4(cout)
    bias = torch.zeros(cop, dtype=torch.float32)
    bias[:cout] = shift.float()
    return Conv(_krsc(w, cop, pad4(cin)), bias, k, stride, k // 2, cop)


def _fold_depthwise(sd: dict[str, Tensor], conv: str, bn: str, stride: int) -> Conv:
    w = sd[f"{conv}.weight"].double()  # [C, 1, k, k]
    c, _, k, _ = w.shape
    scale, shift = _bn_scale_shift(sd, bn)
    w = w[:, 0] * scale[:, None, None]
    cp = pad4(c)
    wk = torch.zeros((k, k, cp), dtype=torch.float32)
    wk[:, :, :c] = w.permute(1, 2, 0).float()
    bias = torch.zeros(cp, dtype=torch.float32)
    bias[:c] = shift.float()
    return Conv(wk.contiguous(), bias, k, stride, k // 2, cp)


def _fold_stem(sd: dict[str, Tensor]) -> Conv:
    w = sd["features.0.0.weight"].double()  # [C, 3, 3, 3]
    scale, shift = _bn_scale_shift(sd, "features.0.1")
    w = w * scale[:, None, None, None]
    cout = w.shape[0]
    cop = pad4(cout)
    bias = torch.zeros(cop, dtype=torch.float32)
    bias[:cout] = shift.float()
    return Conv(_krsc(w, cop, 4), bias, 3, 2, 1, cop)  # RGB + one zero channel, 9 taps -> two K steps


def fold_state_dict(sd: dict[str, Tensor], size: str = "s") -> FoldedEfficientNet:
    stages = block_specs(size)
    blocks: list[Block] = []
    for si, stage in enumerate(stages, start=1):
        for bi, b in enumerate(stage):
            p = f"features.{si}.{bi}.block"
            blk = Block(b)
            if b.kind == "fused":
                blk.convs["conv"] = _fold(sd, f"{p}.0.0", f"{p}.0.1", b.stride)
                if b.expand != 1:
                    blk.convs["project"] = _fold(sd, f"{p}.1.0", f"{p}.1.1", 1)
            else:
                blk.convs["expand"] = _fold(sd, f"{p}.0.0", f"{p}.0.1", 1)
                blk.convs["depthwise"] = _fold_depthwise(sd, f"{p}.1.0", f"{p}.1.1", b.stride)
                blk.convs["fc1"] = _fold(sd, f"{p}.2.fc1", None, 1)
                blk.convs["fc2"] = _fold(sd, f"{p}.2.fc2", None, 1)
                blk.convs["project"] = _fold(sd, f"{p}.3.0", f"{p}.3.1", 1)
            blocks.append(blk)
    head = len(stages) + 1
    return FoldedEfficientNet(_fold_stem(sd), blocks, _fold(sd, f"features.{head}.0", f"features.{head}.1", 1))


# ------------------------------------------------------------------------------------------- execution
def _conv(x: Tensor, c: Conv, act: int, *, residual: Tensor | None = None, gate: Tensor | None = None) -> Tensor:
    b, h, w, cin = x.shape
    ho = (h + 2 * c.pad - c.kernel) // c.stride + 1
    wo = (w + 2 * c.pad - c.kernel) // c.stride + 1
    out = torch.empty((b, ho, wo, c.cout), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    stream = _lib.stream_handle(x.device)
    if gate is None:
        st = lib.isc_conv2d_nhwc(x.data_ptr(), b, h, w, cin, c.weight.data_ptr(), c.cout, c.kernel, c.kernel, c.stride,
                                 c.pad, c.bias.data_ptr(), _lib.ptr(residual), act, out.data_ptr(), stream)
    else:
        st = lib.isc_conv2d_nhwc_gated(x.data_ptr(), b, h, w, cin, gate.data_ptr(), c.weight.data_ptr(), c.cout,
                                       c.kernel, c.kernel, c.stride, c.pad, c.bias.data_ptr(), _lib.ptr(residual), act,
                                       out.data_ptr(), stream)
    _lib.check(st, "isc_conv2d_nhwc")
    return out


def _depthwise(x: Tensor, c: Conv, act: int, *, gate: Tensor | None = None, want_y: bool = True,
               want_pooled: bool = False) -> tuple[Tensor | None, Tensor | None]:
    """`y = act(dwconv(x) + bias) [* gate]` and / or the squeeze-excitation mean `[B, 1, 1, C]` of the un-gated output
    (include/imagescry_hip.h: isc_dwconv2d_nhwc_pool)."""
    b, h, w, ch = x.shape
    ho = (h + 2 * c.pad - c.kernel) // c.stride + 1
    wo = (w + 2 * c.pad - c.kernel) // c.stride + 1
    out = torch.empty((b, ho, wo, ch), dtype=torch.float32, device=x.device) if want_y else None
    pooled = torch.empty((b, 1, 1, ch), dtype=torch.float32, device=x.device) if want_pooled else None
    lib = _lib.load()
    st = lib.isc_dwconv2d_nhwc_pool(x.data_ptr(), b, h, w, ch, c.weight.data_ptr(), c.kernel, c.stride, c.pad,
                                    c.bias.data_ptr(), act, _lib.ptr(gate), _lib.ptr(out), _lib.ptr(pooled),
                                    _lib.stream_handle(x.device))
    _lib.check(st, "isc_dwconv2d_nhwc_pool")
    return out, pooled


def _sweep_shape(c: Conv, w: int) -> bool:
    """The shapes for which the depthwise kernel can pool without writing and can gate its output."""
    return c.kernel == 3 and c.stride == 1 and c.pad == 1 and w <= 14


def _se_gate(pooled: Tensor, fc1: Conv, fc2: Conv) -> Tensor:
    """`sigmoid(fc2(silu(fc1(pooled))))` as one launch (include/imagescry_hip.h: isc_se_gate): `[B, 1, 1, C]` -> `[B, C]`."""
    b, ch = pooled.shape[0], pooled.shape[-1]
    gate = torch.empty((b, ch), dtype=torch.float32, device=pooled.device)
    lib = _lib.load()
    st = lib.isc_se_gate(pooled.data_ptr(), b, ch, fc1.weight.data_ptr(), fc1.weight.shape[1], fc1.bias.data_ptr(),
                         fc1.cout, fc2.weight.data_ptr(), fc2.weight.shape[1], fc2.bias.data_ptr(), gate.data_ptr(),
                         _lib.stream_handle(pooled.device))
    _lib.check(st, "isc_se_gate")
    return gate


def forward_features(net: FoldedEfficientNet, x: Tensor) -> Tensor:
    """float32 NCHW `[B, 3, H, W]` -> float32 NHWC `[B, ceil(H/32), ceil(W/32), 1280]`."""
    lib = _lib.load()
    b, c, h, w = x.shape
    x4 = torch.empty((b, h, w, 4), dtype=torch.float32, device=x.device)
    _lib.check(lib.isc_nchw_to_nhwc(x.data_ptr(), b, c, h, w, 4, x4.data_ptr(), _lib.stream_handle(x.device)),
               "isc_nchw_to_nhwc")
    return forward_features_nhwc4(net, x4)


def forward_features_nhwc4(net: FoldedEfficientNet, x4: Tensor) -> Tensor:
    """float32 NHWC `[B, H, W, 4]` (RGB + a zero channel) -> float32 NHWC `[B, ceil(H/32), ceil(W/32), 1280]`."""
    silu, none = _lib.ISC_ACT_SILU, _lib.ISC_ACT_NONE
    b = x4.shape[0]
    y = _conv(x4, net.stem, silu)
    for blk in net.blocks:
        s = blk.spec
        skip = y if s.residual else None
        if s.kind == "fused":
            if s.expand == 1:
                # torchvision adds the skip AFTER the activation here (the block is conv-BN-SiLU only)
                y = _conv(y, blk.convs["conv"], silu | _lib.ISC_ACT_RESIDUAL_AFTER, residual=skip)
            else:
                t = _conv(y, blk.convs["conv"], silu)
                y = _conv(t, blk.convs["project"], none, residual=skip)
        else:
            t = _conv(y, blk.convs["expand"], silu)
            dw = blk.convs["depthwise"]
            if _sweep_shape(dw, t.shape[2]):
                # two sweeps of the depthwise kernel (mean first, then the gated output): the projection stays plain
                _, pooled = _depthwise(t, dw, silu, want_y=False, want_pooled=True)
                g = _se_gate(pooled, blk.convs["fc1"], blk.convs["fc2"])
                t, _ = _depthwise(t, dw, silu, gate=g)
                y = _conv(t, blk.convs["project"], none, residual=skip)
            else:
                t, pooled = _depthwise(t, dw, silu, want_pooled=True)
                g = _se_gate(pooled, blk.convs["fc1"], blk.convs["fc2"])
                y = _conv(t, blk.convs["project"], none, residual=skip, gate=g)
    return _conv(y, net.head, silu)
