"""ViT-B/16 encoder definition (BASELINE.json configs[4]: "ViT-B/16 (random weights) -> 768-d, fp16"): parameter
naming, seeded init, device-side weight preparation and the forward pass over the HIP transformer blocks.

The reference has no ViT (its encoder is torchvision EfficientNetV2, src/imagescry/models/embedding.py:133-147); this
is the build's own definition behind the same `EmbeddingModule` contract.  State dicts use timm's
`vit_base_patch16_224` parameter names (`cls_token`, `pos_embed`, `patch_embed.proj.weight`, `blocks.3.attn.qkv.weight`,
`blocks.3.mlp.fc1.bias`, `norm.weight`, ...) so such a checkpoint drops in.  The embedding of an image is the class
token after the final LayerNorm (pre-LN encoder, erf-form GELU, LayerNorm eps 1e-6 by default).
"""

from __future__ import annotations

from dataclasses import dataclass, field

import torch
from torch import Tensor

from imagescry_amd import _lib


@dataclass(frozen=True)
class ViTConfig:
    image_size: int = 224
    patch_size: int = 16
    dim: int = 768
    depth: int = 12
    heads: int = 12
    mlp_dim: int = 3072
    ln_eps: float = 1e-6

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1


VIT_B16 = ViTConfig()


def make_state_dict(cfg: ViTConfig = VIT_B16, *, seed: int = 0, randomize_affine: bool = False) -> dict[str, Tensor]:
    """Seeded random parameters (CPU float32): truncated-normal(std 0.02) weights / tokens, zero biases and identity
    LayerNorms -- or, with `randomize_affine`, random biases and LayerNorm affines so that every term is exercised."""
    g = torch.Generator().manual_seed(seed)

    def tn(*shape: int) -> Tensor:
        return torch.nn.init.trunc_normal_(torch.empty(*shape), std=0.02, a=-0.04, b=0.04, generator=g)

    def bias(n: int) -> Tensor:
        return torch.randn(n, generator=g) * 0.05 if randomize_affine else torch.zeros(n)

    def gamma(n: int) -> Tensor:
        return torch.rand(n, generator=g) * 0.5 + 0.75 if randomize_affine else torch.ones(n)

    d = cfg.dim
    sd: dict[str, Tensor] = {
        "cls_token": tn(1, 1, d),
        "pos_embed": tn(1, cfg.tokens, d),
        "patch_embed.proj.weight": tn(d, 3, cfg.patch_size, cfg.patch_size),
        "patch_embed.proj.bias": bias(d),
    }
    for i in range(cfg.depth):
        p = f"blocks.{i}"
        sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"] = gamma(d), bias(d)
        sd[f"{p}.attn.qkv.weight"], sd[f"{p}.attn.qkv.bias"] = tn(3 * d, d), bias(3 * d)
        sd[f"{p}.attn.proj.weight"], sd[f"{p}.attn.proj.bias"] = tn(d, d), bias(d)
        sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"] = gamma(d), bias(d)
        sd[f"{p}.mlp.fc1.weight"], sd[f"{p}.mlp.fc1.bias"] = tn(cfg.mlp_dim, d), bias(cfg.mlp_dim)
        sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc2.bias"] = tn(d, cfg.mlp_dim), bias(d)
    sd["norm.weight"], sd["norm.bias"] = gamma(d), bias(d)
    return sd


def gemm_flops(cfg: ViTConfig = VIT_B16) -> int:
    """Multiply-add FLOPs (2 per MAC) of the GEMMs and the attention products for ONE image."""
    t, d = cfg.tokens, cfg.dim
    per_layer = 2 * t * d * (3 * d) + 2 * t * d * d + 2 * 2 * t * d * cfg.mlp_dim + 2 * 2 * t * t * d
    return 2 * (t - 1) * d * (3 * cfg.patch_size**2) + cfg.depth * per_layer


def pack_rows(x: Tensor) -> Tensor:
    """Row-major `[R, C]` (C % 64 == 0) -> the PACKED layout of include/imagescry_hip.h: rows in tiles of 256 (zero
    padded), columns in K steps of 64, stored `[tile][K step][row][64]`; returned flat."""
    r, c = x.shape
    if c % 64:
        raise ValueError(f"packed matrices need a multiple of 64 columns, got {c}")
    tiles = (r + 255) // 256
    padded = torch.zeros((tiles * 256, c), dtype=x.dtype, device=x.device)
    padded[:r] = x
    return padded.view(tiles, 256, c // 64, 64).permute(0, 2, 1, 3).contiguous().view(-1)


def unpack_rows(flat: Tensor, rows: int, cols: int) -> Tensor:
    """Inverse of `pack_rows`."""
    tiles = (rows + 255) // 256
    return flat.view(tiles, cols // 64, 256, 64).permute(0, 2, 1, 3).reshape(tiles * 256, cols)[:rows].contiguous()


def packed_elems(rows: int, cols: int) -> int:
    return (rows + 255) // 256 * 256 * cols


@dataclass
class Linear:
    weight: Tensor  # fp16, PACKED [out, in] (pack_rows of the torch.nn.Linear weight)
    bias: Tensor  # float32 [out]
    out_features: int = 0
    in_features: int = 0

    def __post_init__(self) -> None:
        if not self.out_features:
            raise ValueError("Linear needs its logical shape")

    def to(self, device: torch.device) -> "Linear":
        return Linear(self.weight.to(device), self.bias.to(device), self.out_features, self.in_features)


@dataclass
class Norm:
    weight: Tensor
    bias: Tensor

    def to(self, device: torch.device) -> "Norm":
        return Norm(self.weight.to(device), self.bias.to(device))


@dataclass
class Block:
    norm1: Norm
    qkv: Linear
    proj: Linear
    norm2: Norm
    fc1: Linear
    fc2: Linear

    def to(self, device: torch.device) -> "Block":
        return Block(*(getattr(self, f).to(device) for f in ("norm1", "qkv", "proj", "norm2", "fc1", "fc2")))


@dataclass
class PreparedViT:
    cfg: ViTConfig
    cls_token: Tensor  # float32 [D]
    pos_embed: Tensor  # float32 [T, D]
    patch: Linear  # [D, 3 * P * P]
    blocks: list[Block] = field(default_factory=list)
    norm: Norm | None = None

    def to(self, device: torch.device) -> "PreparedViT":
        return PreparedViT(self.cfg, self.cls_token.to(device), self.pos_embed.to(device), self.patch.to(device),
                           [b.to(device) for b in self.blocks], self.norm.to(device))


def prepare(sd: dict[str, Tensor], cfg: ViTConfig = VIT_B16) -> PreparedViT:
    """Cast GEMM weights to fp16 (round-to-nearest-even) and store them in the packed layout the GEMM kernels stream; keep
    biases / LayerNorm / tokens in float32."""
    d = cfg.dim
    if cfg.dim % 64 or cfg.mlp_dim % 64 or (3 * cfg.patch_size**2) % 64 or cfg.dim // cfg.heads != 64:
        raise ValueError("this build supports head size 64 and GEMM inner dimensions that are multiples of 64")
    if tuple(sd["pos_embed"].shape) != (1, cfg.tokens, d):
        raise ValueError(f"pos_embed has shape {tuple(sd['pos_embed'].shape)}, expected (1, {cfg.tokens}, {d})")

    def lin(name: str) -> Linear:
        w = sd[f"{name}.weight"]
        w2 = w.reshape(w.shape[0], -1).to(torch.float16)
        return Linear(pack_rows(w2), sd[f"{name}.bias"].float().contiguous(), w2.shape[0], w2.shape[1])

    def norm(name: str) -> Norm:
        return Norm(sd[f"{name}.weight"].float().contiguous(), sd[f"{name}.bias"].float().contiguous())

    net = PreparedViT(cfg, sd["cls_token"].reshape(d).float().contiguous(),
                      sd["pos_embed"].reshape(cfg.tokens, d).float().contiguous(), lin("patch_embed.proj"))
    for i in range(cfg.depth):
        p = f"blocks.{i}"
        net.blocks.append(Block(norm(f"{p}.norm1"), lin(f"{p}.attn.qkv"), lin(f"{p}.attn.proj"), norm(f"{p}.norm2"),
                                lin(f"{p}.mlp.fc1"), lin(f"{p}.mlp.fc2")))
    net.norm = norm("norm")
    return net


def _gemm(a: Tensor, m: int, lin: Linear, out: Tensor, *, act: int = _lib.ISC_ACT_NONE, residual: Tensor | None = None,
          stream: int = 0) -> Tensor:
    """`out = act(a . W^T + b) + residual`; `a` is a packed fp16 activation buffer of `m` rows, `out` either a packed
    fp16 buffer (next GEMM operand) or a row-major float32 `[m, N]` tensor (residual stream)."""
    f16 = out.dtype == torch.float16
    flags = _lib.ISC_GEMM_A_PACKED | _lib.ISC_GEMM_W_PACKED | (_lib.ISC_GEMM_OUT_PACKED if f16 else 0)
    st = _lib.load().isc_gemm_f16(a.data_ptr(), m, lin.in_features, lin.weight.data_ptr(), lin.out_features,
                                  lin.bias.data_ptr(), _lib.ptr(residual), act, out.data_ptr(),
                                  _lib.ISC_F16 if f16 else _lib.ISC_F32, flags, stream)
    _lib.check(st, "isc_gemm_f16")
    return out


def _layernorm_packed(x: Tensor, rows: int, nrm: Norm, eps: float, out: Tensor, stream: int) -> Tensor:
    d = nrm.weight.shape[0]
    st = _lib.load().isc_layernorm(x.data_ptr(), rows, d, d, nrm.weight.data_ptr(), nrm.bias.data_ptr(), eps,
                                   out.data_ptr(), _lib.ISC_F16, d, 1, stream)
    _lib.check(st, "isc_layernorm")
    return out


def forward_cls(net: PreparedViT, x: Tensor) -> Tensor:
    """float32 `[B, 3, S, S]` (already preprocessed) on a HIP device -> float32 `[B, D]` class-token features.

    Every fp16 activation (patches, LayerNorm outputs, qkv, attention output, MLP hidden) lives in the packed layout;
    the float32 residual stream is row-major."""
    cfg = net.cfg
    b = x.shape[0]
    t, d = cfg.tokens, cfg.dim
    m = b * t
    dev = x.device
    lib = _lib.load()
    stream = _lib.stream_handle(dev)
    f16, f32 = torch.float16, torch.float32
    kdim = 3 * cfg.patch_size**2

    def packed(rows: int, cols: int) -> Tensor:
        # rows past `rows` in the last tile are never read into a result that is kept; they only have to be finite
        # enough not to matter, and every consumer masks them, so the buffer is left uninitialised
        return torch.empty(packed_elems(rows, cols), dtype=f16, device=dev)

    patches = packed(b * (t - 1), kdim)
    _lib.check(lib.isc_patchify_f16(x.data_ptr(), b, 3, cfg.image_size, cfg.image_size, cfg.patch_size,
                                    patches.data_ptr(), 1, stream), "isc_patchify_f16")
    pe = torch.empty((b * (t - 1), d), dtype=f32, device=dev)
    _gemm(patches, b * (t - 1), net.patch, pe, stream=stream)
    del patches
    xa = torch.empty((m, d), dtype=f32, device=dev)  # residual stream (ping)
    xb = torch.empty((m, d), dtype=f32, device=dev)  # residual stream (pong)
    _lib.check(lib.isc_vit_assemble(pe.data_ptr(), net.cls_token.data_ptr(), net.pos_embed.data_ptr(), b, t, d,
                                    xa.data_ptr(), stream), "isc_vit_assemble")
    del pe
    hbuf, qkv, att, mlp = packed(m, d), packed(m, 3 * d), packed(m, d), packed(m, cfg.mlp_dim)
    for blk in net.blocks:
        _layernorm_packed(xa, m, blk.norm1, cfg.ln_eps, hbuf, stream)
        _gemm(hbuf, m, blk.qkv, qkv, stream=stream)
        _lib.check(lib.isc_attention_f16(qkv.data_ptr(), b, t, cfg.heads, d // cfg.heads, att.data_ptr(), 1, stream),
                   "isc_attention_f16")
        _gemm(att, m, blk.proj, xb, residual=xa, stream=stream)
        _layernorm_packed(xb, m, blk.norm2, cfg.ln_eps, hbuf, stream)
        _gemm(hbuf, m, blk.fc1, mlp, act=_lib.ISC_ACT_GELU, stream=stream)
        _gemm(mlp, m, blk.fc2, xa, residual=xb, stream=stream)
    out = torch.empty((b, d), dtype=f32, device=dev)
    st = lib.isc_layernorm(xa.data_ptr(), b, d, t * d, net.norm.weight.data_ptr(), net.norm.bias.data_ptr(), cfg.ln_eps,
                           out.data_ptr(), _lib.ISC_F32, d, 0, stream)  # class-token rows only (row stride T * D)
    _lib.check(st, "isc_layernorm")
    return out


__all__ = ["VIT_B16", "ViTConfig", "forward_cls", "gemm_flops", "make_state_dict", "pack_rows", "prepare", "unpack_rows"]
