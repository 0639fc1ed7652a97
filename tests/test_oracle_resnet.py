"""The ResNet-50 restatement (oracle/encoder_oracle.py::resnet50_forward) against an independent implementation:
`transformers.ResNetModel` (v1.5: stride on the 3 x 3) built from a CONFIG OBJECT -- no download, no pretrained name -- and
loaded with the same seeded weights.  The reference has no ResNet (its encoder is torchvision EfficientNetV2,
src/imagescry/models/embedding.py:133-147) and torchvision is not installed here, so this is the pin the encoder VALUES of
BASELINE config 2 have: two implementations that share nothing but the weights agree to float32 rounding.  CPU only."""

import pytest
import torch

from imagescry_amd import resnet50
from oracle import encoder_oracle


def _to_transformers_state_dict(sd: dict[str, torch.Tensor], want_keys: list[str]) -> dict[str, torch.Tensor]:
    """torchvision parameter names -> transformers.ResNetModel names."""
    bn_fields = ("weight", "bias", "running_mean", "running_var")
    out: dict[str, torch.Tensor] = {"embedder.embedder.convolution.weight": sd["conv1.weight"]}
    for f in bn_fields:
        out[f"embedder.embedder.normalization.{f}"] = sd[f"bn1.{f}"]
    for li, (_planes, blocks, _stride) in enumerate(resnet50.STAGES, start=1):
        for bi in range(blocks):
            src, dst = f"layer{li}.{bi}", f"encoder.stages.{li - 1}.layers.{bi}"
            for j in (1, 2, 3):
                out[f"{dst}.layer.{j - 1}.convolution.weight"] = sd[f"{src}.conv{j}.weight"]
                for f in bn_fields:
                    out[f"{dst}.layer.{j - 1}.normalization.{f}"] = sd[f"{src}.bn{j}.{f}"]
            if f"{src}.downsample.0.weight" in sd:
                out[f"{dst}.shortcut.convolution.weight"] = sd[f"{src}.downsample.0.weight"]
                for f in bn_fields:
                    out[f"{dst}.shortcut.normalization.{f}"] = sd[f"{src}.downsample.1.{f}"]
    for k in want_keys:
        if k.endswith("num_batches_tracked"):
            out[k] = torch.tensor(0)
    return out


def test_resnet50_oracle_matches_transformers():
    transformers = pytest.importorskip("transformers")
    sd = resnet50.make_state_dict(embedding_dim=768, seed=3, randomize_bn=True)  # random affine + running statistics
    cfg = transformers.ResNetConfig(
        num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
        layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False,
    )
    model = transformers.ResNetModel(cfg).eval()
    want_keys = list(model.state_dict().keys())
    hf_sd = _to_transformers_state_dict(sd, want_keys)
    assert set(hf_sd) == set(want_keys)
    model.load_state_dict(hf_sd)
    x = torch.randn(2, 3, 96, 80, generator=torch.Generator().manual_seed(4)).clamp(-3, 3)
    with torch.no_grad():
        pooled = model(pixel_values=x).pooler_output.flatten(1)  # [B, 2048]: trunk + global average pool
        want = torch.nn.functional.linear(pooled, sd["fc.weight"], sd["fc.bias"])[:, :, None, None]
        got = encoder_oracle.resnet50_forward(x, sd)
    assert got.shape == want.shape == (2, 768, 1, 1)
    scale = float(want.abs().max())
    torch.testing.assert_close(got, want, rtol=0, atol=2e-5 * scale)


def test_resnet50_parameter_count_is_the_published_one():
    """23 508 032 parameters in the trunk (torchvision's resnet50 without `fc`: 25 557 032 - 2 049 000)."""
    sd = resnet50.make_state_dict(embedding_dim=768)
    trunk = sum(v.numel() for k, v in sd.items() if not k.startswith("fc.") and "running" not in k)
    assert trunk == 23_508_032
