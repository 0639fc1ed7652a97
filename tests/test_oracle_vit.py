"""The ViT-B/16 restatement (oracle/vit_oracle.py) against an independent implementation: `transformers.ViTModel`
loaded with the same weights.  CPU only."""

import pytest
import torch

from imagescry_amd import vit
from oracle.vit_oracle import to_transformers_state_dict, vit_forward


def test_vit_oracle_matches_transformers():
    transformers = pytest.importorskip("transformers")
    cfg = vit.ViTConfig(depth=3)  # three blocks exercise everything a twelve-block model does
    sd = vit.make_state_dict(cfg, seed=5, randomize_affine=True)
    hf_cfg = transformers.ViTConfig(num_hidden_layers=cfg.depth, layer_norm_eps=cfg.ln_eps)
    model = transformers.ViTModel(hf_cfg, add_pooling_layer=False).eval()
    hf_sd = to_transformers_state_dict(sd, list(model.state_dict().keys()))
    assert set(hf_sd) == set(model.state_dict().keys())
    model.load_state_dict(hf_sd)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1)).clamp(-3, 3)
    with torch.no_grad():
        want = model(pixel_values=x).last_hidden_state[:, 0]
        got = vit_forward(sd, x, eps=cfg.ln_eps)
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4)


def test_vit_state_dict_shapes_and_parameter_count():
    sd = vit.make_state_dict()
    assert sum(v.numel() for v in sd.values()) == 85_798_656  # ViT-B/16 without pooler / classifier head
    assert vit.VIT_B16.tokens == 197
    net = vit.prepare(sd)
    qkv = net.blocks[0].qkv
    assert qkv.weight.dtype == torch.float16 and (qkv.out_features, qkv.in_features) == (2304, 768)
    assert torch.equal(vit.unpack_rows(qkv.weight, 2304, 768), sd["blocks.0.attn.qkv.weight"].half())  # packed storage
    assert (net.patch.out_features, net.patch.in_features) == (768, 768)
    x = torch.arange(300 * 128, dtype=torch.float32).reshape(300, 128)
    assert torch.equal(vit.unpack_rows(vit.pack_rows(x), 300, 128), x) and vit.pack_rows(x).numel() == 512 * 128
    with pytest.raises(ValueError):
        vit.prepare(sd, vit.ViTConfig(image_size=256))


def test_fp16_operand_rounding_is_small():
    """The fp16 configuration's precision cost on the L2-normalised embedding stays far inside the 1e-2 tolerance."""
    cfg = vit.ViTConfig(depth=2)
    sd = vit.make_state_dict(cfg, seed=2, randomize_affine=True)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(3)).clamp(-3, 3)
    with torch.no_grad():
        a = torch.nn.functional.normalize(vit_forward(sd, x), dim=1)
        b = torch.nn.functional.normalize(vit_forward(sd, x, round_operands_fp16=True), dim=1)
    assert (a - b).abs().max().item() < 2e-3
