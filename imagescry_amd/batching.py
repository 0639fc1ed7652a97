"""Same-shape batching harness: the caller-side loop that feeds variable-size images to the hot path.

`SimilarShapeBatcher` reproduces the reference sampler (src/imagescry/data.py:403-452): index the image shapes,
sort them, group equal shapes, cut every group into chunks of at most `max_batch_size`.  `ImageTensorDataset` is an
in-memory stand-in for the reference's `ImageFilesDataset` (data.py:148-250) -- decoding image files is host I/O
outside the hot path -- whose `get_loader` yields `ImageBatch` objects exactly like the reference's loader does
(`_collate_image_batch`, data.py:455-459).
"""

from __future__ import annotations

from typing import Iterable, Iterator, Sequence

import torch
from torch import Tensor

from imagescry_amd.data import ImageBatch

__all__ = ["ImageTensorDataset", "SimilarShapeBatcher"]


class SimilarShapeBatcher:
    """Batches of dataset indices in which every image has the same `(height, width)`."""

    def __init__(self, image_shapes: Iterable[tuple[int, int]], max_batch_size: int) -> None:
        if max_batch_size < 1:
            raise ValueError(f"max_batch_size must be at least 1, got {max_batch_size}")
        self.max_batch_size = max_batch_size
        indexed = sorted(enumerate(tuple(s) for s in image_shapes), key=lambda item: item[1])
        self.batched_indexes: list[list[int]] = []
        group: list[int] = []
        previous: tuple[int, ...] | None = None
        for idx, shape in indexed:
            if previous is not None and shape != previous:  # a new shape group starts
                self._flush(group)
                group = []
            group.append(idx)
            previous = shape
        self._flush(group)

    def _flush(self, group: list[int]) -> None:
        for i in range(0, len(group), self.max_batch_size):
            self.batched_indexes.append(group[i : i + self.max_batch_size])

    def __iter__(self) -> Iterator[list[int]]:
        yield from self.batched_indexes

    def __len__(self) -> int:
        return len(self.batched_indexes)


class ImageTensorDataset:
    """uint8 RGB images `[3, H, W]` of arbitrary sizes held in memory."""

    def __init__(self, images: Sequence[Tensor]) -> None:
        for img in images:
            if not isinstance(img, Tensor) or img.dtype != torch.uint8 or img.ndim != 3 or img.shape[0] != 3:
                raise TypeError("every image must be a uint8 [3, H, W] tensor")
        self.images = list(images)

    def __len__(self) -> int:
        return len(self.images)

    def __getitem__(self, idx: int) -> tuple[Tensor, Tensor]:
        return torch.tensor(idx), self.images[idx]

    @property
    def shapes(self) -> list[tuple[int, int]]:
        return [(int(img.shape[1]), int(img.shape[2])) for img in self.images]

    def get_loader(self, max_batch_size: int) -> Iterator[ImageBatch]:
        """`ImageBatch` objects, images grouped by shape (reference: `ImageFilesDataset.get_loader`, data.py:213-250)."""
        for indexes in SimilarShapeBatcher(self.shapes, max_batch_size):
            yield ImageBatch(
                indices=torch.tensor(indexes, dtype=torch.int64),
                images=torch.stack([self.images[i] for i in indexes]),
            )
