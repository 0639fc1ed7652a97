"""CPU restatement of the ViT-B/16 encoder (test infrastructure, see oracle/__init__.py).

The reference has no ViT (its only concrete encoder is torchvision EfficientNetV2,
src/imagescry/models/embedding.py:133-147); BASELINE.json configs[4] names "ViT-B/16 (random weights) -> 768-d, fp16".
`vit_forward` is the build's own definition written with plain float32 `torch.nn.functional` calls on a timm-named
state dict: Conv2d patch embedding, class token + position embedding, pre-LN blocks (fused qkv Linear, softmax
attention, exact-erf GELU MLP), final LayerNorm, class-token row.  It is pinned against an independent implementation
-- `transformers.ViTModel` with the same weights -- in tests/test_oracle_vit.py.  Encoder values: parity unpinned by
the reference (nothing in it computes a ViT).
"""

from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import Tensor


def vit_forward(sd: dict[str, Tensor], x: Tensor, *, patch: int = 16, heads: int = 12, eps: float = 1e-6,
                round_operands_fp16: bool = False) -> Tensor:
    """float32 [B, 3, S, S] -> float32 [B, D] (class token after the final LayerNorm).

    `round_operands_fp16` rounds every matrix-product operand to fp16 (the values, not the arithmetic: products are
    then exact in float32 and sums are float32), which is what the HIP path feeds the matrix cores; the difference to
    the plain float32 result is the precision cost of the fp16 configuration, not an implementation error."""
    r = (lambda t: t.half().float()) if round_operands_fp16 else (lambda t: t)
    d = sd["cls_token"].shape[-1]
    b = x.shape[0]
    tok = F.conv2d(r(x), r(sd["patch_embed.proj.weight"]), sd["patch_embed.proj.bias"], stride=patch)
    tok = tok.flatten(2).transpose(1, 2)  # [B, T-1, D], patches row-major
    tok = torch.cat([sd["cls_token"].expand(b, -1, -1), tok], dim=1) + sd["pos_embed"]
    t = tok.shape[1]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(depth):
        p = f"blocks.{i}"
        h = F.layer_norm(tok, (d,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], eps)
        qkv = F.linear(r(h), r(sd[f"{p}.attn.qkv.weight"]), sd[f"{p}.attn.qkv.bias"])
        qkv = r(qkv).reshape(b, t, 3, heads, d // heads).permute(2, 0, 3, 1, 4)  # [3, B, H, T, 64]
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = torch.softmax((q @ k.transpose(-1, -2)) * (d // heads) ** -0.5, dim=-1)
        a = (r(att) @ v).transpose(1, 2).reshape(b, t, d)
        tok = tok + F.linear(r(a), r(sd[f"{p}.attn.proj.weight"]), sd[f"{p}.attn.proj.bias"])
        h = F.layer_norm(tok, (d,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], eps)
        h = F.gelu(F.linear(r(h), r(sd[f"{p}.mlp.fc1.weight"]), sd[f"{p}.mlp.fc1.bias"]))
        tok = tok + F.linear(r(h), r(sd[f"{p}.mlp.fc2.weight"]), sd[f"{p}.mlp.fc2.bias"])
    return F.layer_norm(tok[:, 0], (d,), sd["norm.weight"], sd["norm.bias"], eps)


def to_transformers_state_dict(sd: dict[str, Tensor], hf_keys: list[str]) -> dict[str, Tensor]:
    """Rename a timm-style ViT state dict to the parameter names of the installed `transformers.ViTModel`
    (whose naming changed between releases: both the classic `encoder.layer.N.attention.attention.query` and the
    newer `layers.N.attention.q_proj` schemes are handled)."""
    d = sd["cls_token"].shape[-1]
    out: dict[str, Tensor] = {}
    new_style = any(k.startswith("layers.") for k in hf_keys)
    out["embeddings.cls_token"] = sd["cls_token"]
    out["embeddings.position_embeddings"] = sd["pos_embed"]
    out["embeddings.patch_embeddings.projection.weight"] = sd["patch_embed.proj.weight"]
    out["embeddings.patch_embeddings.projection.bias"] = sd["patch_embed.proj.bias"]
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(depth):
        p = f"blocks.{i}"
        qw, kw, vw = sd[f"{p}.attn.qkv.weight"].split(d)
        qb, kb, vb = sd[f"{p}.attn.qkv.bias"].split(d)
        if new_style:
            h = f"layers.{i}"
            names = {"q": f"{h}.attention.q_proj", "k": f"{h}.attention.k_proj", "v": f"{h}.attention.v_proj",
                     "o": f"{h}.attention.o_proj", "fc1": f"{h}.mlp.fc1", "fc2": f"{h}.mlp.fc2"}
        else:
            h = f"encoder.layer.{i}"
            names = {"q": f"{h}.attention.attention.query", "k": f"{h}.attention.attention.key",
                     "v": f"{h}.attention.attention.value", "o": f"{h}.attention.output.dense",
                     "fc1": f"{h}.intermediate.dense", "fc2": f"{h}.output.dense"}
        for nm, w_, b_ in (("q", qw, qb), ("k", kw, kb), ("v", vw, vb)):
            out[names[nm] + ".weight"], out[names[nm] + ".bias"] = w_, b_
        out[names["o"] + ".weight"], out[names["o"] + ".bias"] = sd[f"{p}.attn.proj.weight"], sd[f"{p}.attn.proj.bias"]
        out[names["fc1"] + ".weight"], out[names["fc1"] + ".bias"] = sd[f"{p}.mlp.fc1.weight"], sd[f"{p}.mlp.fc1.bias"]
        out[names["fc2"] + ".weight"], out[names["fc2"] + ".bias"] = sd[f"{p}.mlp.fc2.weight"], sd[f"{p}.mlp.fc2.bias"]
        out[f"{h}.layernorm_before.weight"], out[f"{h}.layernorm_before.bias"] = sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"]
        out[f"{h}.layernorm_after.weight"], out[f"{h}.layernorm_after.bias"] = sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"]
    out["layernorm.weight"], out["layernorm.bias"] = sd["norm.weight"], sd["norm.bias"]
    return out
