"""Boundary containers: same contract as the reference's `ImageBatch` / `EmbeddingBatch`
(src/imagescry/data.py:29-144; reference tests: tests/test_data.py:49-68, tests/test_models/test_embedding.py:29-75,
tests/test_typechecking.py:19-35 for the "wrong type / shape / dtype raises" behaviour)."""

from __future__ import annotations

import dataclasses

import pytest
import torch

from imagescry_amd import EmbeddingBatch, ImageBatch


def test_image_batch_device_and_len() -> None:
    batch = ImageBatch(indices=torch.tensor([0, 1, 2]), images=torch.randint(0, 255, (3, 3, 3, 3)).to(torch.uint8))
    assert batch.device == torch.device("cpu")
    assert len(batch) == 3
    assert batch.cpu().device == torch.device("cpu")
    moved = batch.to("cpu")
    assert isinstance(moved, ImageBatch) and torch.equal(moved.images, batch.images)


def test_batches_are_frozen() -> None:
    batch = ImageBatch(indices=torch.tensor([0]), images=torch.zeros((1, 3, 2, 2), dtype=torch.uint8))
    with pytest.raises(dataclasses.FrozenInstanceError):
        batch.images = batch.images  # type: ignore[misc]


@pytest.mark.parametrize(
    "indices,images,exc",
    [
        (torch.tensor([0, 1]), torch.zeros((2, 3, 4, 4), dtype=torch.float32), TypeError),  # images must be uint8
        (torch.tensor([0.0, 1.0]), torch.zeros((2, 3, 4, 4), dtype=torch.uint8), TypeError),  # indices must be int64
        (torch.tensor([0, 1]), torch.zeros((2, 4, 4, 4), dtype=torch.uint8), ValueError),  # C must be 3
        (torch.tensor([0, 1]), torch.zeros((3, 4, 4), dtype=torch.uint8), ValueError),  # rank 4
        (torch.tensor([0, 1, 2]), torch.zeros((2, 3, 4, 4), dtype=torch.uint8), ValueError),  # B mismatch
        ([0, 1], torch.zeros((2, 3, 4, 4), dtype=torch.uint8), TypeError),  # not a tensor
    ],
)
def test_image_batch_rejects_bad_inputs(indices, images, exc) -> None:
    with pytest.raises(exc):
        ImageBatch(indices=indices, images=images)


def test_device_mismatch_raises_value_error() -> None:
    """reference data.py:46-52 -- exercised with the meta device, which needs no GPU."""
    with pytest.raises(ValueError, match="same device"):
        ImageBatch(indices=torch.tensor([0]), images=torch.zeros((1, 3, 2, 2), dtype=torch.uint8, device="meta"))
    with pytest.raises(ValueError, match="same device"):
        EmbeddingBatch(indices=torch.tensor([0]), embeddings=torch.zeros((1, 4, 2, 2), device="meta"))


def test_embedding_batch_properties_and_flat_vectors() -> None:
    """reference tests/test_models/test_embedding.py:29-75."""
    torch.manual_seed(1234)
    batch_size, embedding_dim, spatial_dims = 3, 128, (7, 10)
    batch = EmbeddingBatch(indices=torch.arange(batch_size), embeddings=torch.randn(batch_size, embedding_dim, *spatial_dims))
    assert len(batch) == batch_size
    assert batch.embedding_dim == embedding_dim
    assert batch.spatial_dims == spatial_dims
    flat = batch.get_flat_vectors()
    assert flat.shape == (batch_size * spatial_dims[0] * spatial_dims[1], embedding_dim)
    assert torch.equal(flat, batch.embeddings.permute(0, 2, 3, 1).reshape(-1, embedding_dim))
    # row order is (b, h, w)
    assert torch.equal(flat[1 * 70 + 2 * 10 + 3], batch.embeddings[1, :, 2, 3])


def test_embedding_batch_rejects_bad_inputs() -> None:
    with pytest.raises(TypeError):
        EmbeddingBatch(indices=torch.tensor([0]), embeddings=torch.zeros((1, 4, 2, 2), dtype=torch.int32))
    with pytest.raises(ValueError):
        EmbeddingBatch(indices=torch.tensor([0]), embeddings=torch.zeros((1, 4, 2)))
    with pytest.raises(ValueError):
        EmbeddingBatch(indices=torch.tensor([0, 1]), embeddings=torch.zeros((1, 4, 2, 2)))
    half = EmbeddingBatch(indices=torch.tensor([0]), embeddings=torch.zeros((1, 4, 2, 2), dtype=torch.float16))
    assert half.embedding_dim == 4
