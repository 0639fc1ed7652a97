"""Host-side logic that needs no GPU: sharding arithmetic, argument validation, the loud failure when tensors
are not on a HIP device (there is no CPU fallback), ResNet-50 BatchNorm folding and weight layout."""

from __future__ import annotations

import pytest
import torch
import torch.nn.functional as F

from imagescry_amd import (
    EmbeddingBank,
    ImageBatch,
    ResNet50Embedder,
    normalize_per_channel,
    resize,
    resnet50,
    shard_bounds,
    to_4d,
)
from imagescry_amd._lib import HipLibraryError
from oracle import encoder_oracle


def test_shard_bounds_partition_rows() -> None:
    for n in (0, 1, 7, 1000, 10_000_000):
        for g in (1, 2, 3, 8):
            spans = [shard_bounds(n, g, r) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(10_000_000, 8, 3) == (3_750_000, 5_000_000)
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_no_cpu_fallback() -> None:
    """Every product entry point refuses CPU tensors loudly instead of computing on the host."""
    img = torch.zeros((2, 3, 8, 8), dtype=torch.uint8)
    with pytest.raises(HipLibraryError, match="no CPU fallback"):
        normalize_per_channel(img)
    with pytest.raises(HipLibraryError, match="no CPU fallback"):
        resize(img, 4)
    with pytest.raises(HipLibraryError, match="no CPU fallback"):
        EmbeddingBank(torch.randn(10, 64))
    model = ResNet50Embedder(seed=0)
    with pytest.raises(HipLibraryError, match="no CPU fallback"):
        model.predict_step(ImageBatch(indices=torch.arange(2), images=img))


def test_transform_argument_validation() -> None:
    with pytest.raises(ValueError):
        normalize_per_channel(torch.zeros((3, 8, 8), dtype=torch.uint8))  # needs B C H W
    with pytest.raises(TypeError):
        normalize_per_channel([1, 2, 3])  # type: ignore[arg-type]
    with pytest.raises(ValueError):
        resize(torch.zeros((1, 1, 1, 3, 8, 8)), 4)
    with pytest.raises(ValueError):
        to_4d(torch.zeros(2))
    assert to_4d(torch.randn(3, 4)).shape == (1, 1, 3, 4)
    assert to_4d(torch.randn(3, 5, 7)).shape == (1, 3, 5, 7)
    assert to_4d(torch.randn(16, 3, 5, 7)).shape == (16, 3, 5, 7)


def test_bank_argument_validation() -> None:
    with pytest.raises(TypeError):
        EmbeddingBank(torch.zeros((4, 8), dtype=torch.int64))
    with pytest.raises(ValueError):
        EmbeddingBank(torch.zeros((4, 8, 1)))
    with pytest.raises(ValueError):
        EmbeddingBank(torch.zeros((4, 8)), dtype=torch.bfloat16)


def test_embedder_constructor_validation() -> None:
    with pytest.raises(ValueError):
        ResNet50Embedder(embedding_dim=770)
    with pytest.raises(ValueError):
        ResNet50Embedder(max_side_length=0)
    sd = resnet50.make_state_dict(embedding_dim=128, seed=1)
    with pytest.raises(ValueError):
        ResNet50Embedder(embedding_dim=768, state_dict=sd)
    m = ResNet50Embedder(embedding_dim=128, state_dict=sd)
    assert m.embedding_dim == 128 and m.device == torch.device("cpu")
    assert m.hparams == {"embedding_dim": 128, "max_side_length": 640}
    with pytest.raises(TypeError):
        m.preprocess(torch.zeros((1, 3, 4, 4)))  # uint8 only, as the reference's UInt8 annotation
    with pytest.raises(TypeError):
        m.predict_step(torch.zeros((1, 3, 4, 4), dtype=torch.uint8))  # type: ignore[arg-type]


def test_state_dict_layout() -> None:
    sd = resnet50.make_state_dict(seed=0)
    assert sd["conv1.weight"].shape == (64, 3, 7, 7)
    assert sd["layer1.0.downsample.0.weight"].shape == (256, 64, 1, 1)
    assert sd["layer2.0.conv2.weight"].shape == (128, 128, 3, 3)
    assert sd["layer4.2.conv3.weight"].shape == (2048, 512, 1, 1)
    assert "layer1.1.downsample.0.weight" not in sd
    assert sd["fc.weight"].shape == (768, 2048)
    n_params = sum(v.numel() for k, v in sd.items() if "running" not in k)
    assert n_params == 23_508_032 + 768 * 2048 + 768  # torchvision trunk + the 768-d head
    assert torch.equal(resnet50.make_state_dict(seed=0)["layer3.4.conv2.weight"], sd["layer3.4.conv2.weight"])
    assert abs(resnet50.conv_flops(1, 224, 224) / 1e9 - 8.18) < 0.01


def test_bn_folding_and_krsc_layout_match_the_unfused_oracle() -> None:
    """Folded (conv + BN) weights, applied with plain torch, reproduce the oracle's conv -> batch_norm."""
    sd = resnet50.make_state_dict(seed=5, randomize_bn=True)
    net = resnet50.fold_state_dict(sd)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 9, 9, generator=g)
    blk = net.blocks[0]
    w = blk.conv2.weight.permute(0, 3, 1, 2)  # KRSC -> OIHW
    got = F.conv2d(x, w, blk.conv2.bias, stride=blk.conv2.stride, padding=blk.conv2.pad)
    exp = encoder_oracle._bn(F.conv2d(x, sd["layer1.0.conv2.weight"], padding=1), sd, "layer1.0.bn2")
    assert torch.allclose(got, exp, rtol=1e-4, atol=1e-5)
    # stem: [64, 56 taps, 4 channels], taps ordered (r, s); channel 3 and taps 49..55 are zero padding
    img = torch.randn(1, 3, 20, 20, generator=g)
    cols = F.unfold(img, 7, stride=2, padding=3).reshape(1, 3, 49, -1).permute(0, 3, 2, 1).reshape(-1, 147)
    assert net.stem.weight.shape == (64, resnet50.STEM_TAPS, 4)
    w147 = net.stem.weight[:, :49, :3].reshape(64, 147)
    got = (cols @ w147.T + net.stem.bias).T.reshape(1, 64, 10, 10)
    exp = encoder_oracle._bn(F.conv2d(img, sd["conv1.weight"], stride=2, padding=3), sd, "bn1")
    assert torch.allclose(got, exp, rtol=1e-4, atol=1e-5)
    assert float(net.stem.weight[:, 49:].abs().max()) == 0.0 and float(net.stem.weight[:, :, 3].abs().max()) == 0.0
    assert len(net.blocks) == 16 and sum(b.downsample is not None for b in net.blocks) == 4


def test_efficientnet_v2_parameter_counts_match_torchvision() -> None:
    """torchvision is absent, so the restated architecture is pinned by the published parameter totals of
    efficientnet_v2_{s,m,l} (21,458,488 / 54,139,356 / 118,515,272) minus the 1280 -> 1000 classifier the reference
    drops (`.features` only, embedding.py:147)."""
    from imagescry_amd import efficientnet

    for size, total in (("s", 21_458_488), ("m", 54_139_356), ("l", 118_515_272)):
        sd = efficientnet.make_state_dict(size)
        n = sum(v.numel() for k, v in sd.items() if "running" not in k)
        assert n == total - (1280 * 1000 + 1000)
    sd = efficientnet.make_state_dict("s")
    assert sd["features.0.0.weight"].shape == (24, 3, 3, 3)
    assert sd["features.2.0.block.0.0.weight"].shape == (96, 24, 3, 3)
    assert sd["features.4.0.block.1.0.weight"].shape == (256, 1, 3, 3)  # depthwise
    assert sd["features.4.0.block.2.fc1.weight"].shape == (16, 256, 1, 1)  # squeeze = in_channels // 4
    assert sd["features.7.0.weight"].shape == (1280, 256, 1, 1)
