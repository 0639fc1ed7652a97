"""Boundary types of the embed-and-search hot path.

`ImageBatch` is what the encoder consumes and `EmbeddingBatch` is what it
produces; both mirror the reference containers field-for-field so callers can
switch without edits (reference: src/imagescry/data.py:29-76 for `ImageBatch`,
src/imagescry/data.py:79-144 for `EmbeddingBatch`).

The reference validates dtype / rank / shape through jaxtyping + beartype and
raises `TypeCheckError` (tests/test_typechecking.py:19-35).  Neither package is
available here, so the same contract is enforced by hand in `__post_init__`:
wrong container type or dtype -> `TypeError`, wrong rank / shape or mismatched
devices -> `ValueError`.
"""

from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import Tensor

__all__ = ["EmbeddingBatch", "ImageBatch"]


def _require_tensor(name: str, value: object) -> Tensor:
    if not isinstance(value, Tensor):
        raise TypeError(f"{name} must be a torch.Tensor, got {type(value).__name__}")
    return value


def _check_indices(indices: Tensor, batch: int) -> None:
    if indices.dtype != torch.int64:
        raise TypeError(f"indices must be int64, got {indices.dtype}")
    if indices.ndim != 1:
        raise ValueError(f"indices must be 1-D [B], got shape {tuple(indices.shape)}")
    if indices.shape[0] != batch:
        raise ValueError(f"indices has {indices.shape[0]} entries but the batch holds {batch} items")


@dataclass(frozen=True, slots=True)
class ImageBatch:
    """Batch of RGB uint8 images `[B, 3, H, W]` (NCHW) and their dataset indices `[B]`.

    Reference: src/imagescry/data.py:29-76.
    """

    indices: Tensor
    images: Tensor

    def __post_init__(self) -> None:
        indices = _require_tensor("indices", self.indices)
        images = _require_tensor("images", self.images)
        if images.dtype != torch.uint8:
            raise TypeError(f"images must be uint8, got {images.dtype}")
        if images.ndim != 4 or images.shape[1] != 3:
            raise ValueError(f"images must have shape [B, 3, H, W], got {tuple(images.shape)}")
        _check_indices(indices, images.shape[0])
        # reference: data.py:46-52
        if indices.device != images.device:
            raise ValueError(
                "Tensors must be on the same device. "
                f"Got indices on {indices.device} and images on {images.device}"
            )

    def __len__(self) -> int:
        return len(self.indices)

    def to(self, device: str | torch.device) -> "ImageBatch":
        """New batch with both tensors on `device` (reference: data.py:62-71)."""
        return ImageBatch(indices=self.indices.to(device), images=self.images.to(device))

    def cpu(self) -> "ImageBatch":
        return self.to("cpu")

    @property
    def device(self) -> torch.device:
        return self.indices.device


@dataclass(frozen=True, slots=True)
class EmbeddingBatch:
    """Batch of embedding feature maps `[B, E, H, W]` (floating) and their dataset indices `[B]`.

    Reference: src/imagescry/data.py:79-144.
    """

    indices: Tensor
    embeddings: Tensor

    def __post_init__(self) -> None:
        indices = _require_tensor("indices", self.indices)
        embeddings = _require_tensor("embeddings", self.embeddings)
        if not embeddings.dtype.is_floating_point:
            raise TypeError(f"embeddings must be floating point, got {embeddings.dtype}")
        if embeddings.ndim != 4:
            raise ValueError(f"embeddings must have shape [B, E, H, W], got {tuple(embeddings.shape)}")
        _check_indices(indices, embeddings.shape[0])
        # reference: data.py:96-102
        if indices.device != embeddings.device:
            raise ValueError(
                "Tensors must be on the same device. "
                f"Got indices on {indices.device} and embeddings on {embeddings.device}"
            )

    def __len__(self) -> int:
        return len(self.indices)

    def get_flat_vectors(self) -> Tensor:
        """`[B, E, H, W]` -> `[B*H*W, E]`, rows ordered (b, h, w) (reference: data.py:112-118)."""
        return self.embeddings.permute(0, 2, 3, 1).reshape(-1, self.embedding_dim)

    def to(self, device: str | torch.device) -> "EmbeddingBatch":
        return EmbeddingBatch(indices=self.indices.to(device), embeddings=self.embeddings.to(device))

    def cpu(self) -> "EmbeddingBatch":
        return self.to("cpu")

    @property
    def device(self) -> torch.device:
        return self.indices.device

    @property
    def embedding_dim(self) -> int:
        return self.embeddings.size(1)

    @property
    def spatial_dims(self) -> tuple[int, int]:
        return self.embeddings.size(2), self.embeddings.size(3)
