"""Embed-then-compress pipeline (the caller of the hot path).

Mirrors the arithmetic of the reference's `EmbeddingPCAPipeline.predict_step` / `predict`
(src/imagescry/models/pipelines.py:22-131): embedder.predict_step -> flat vectors -> PCA.transform -> back to
`[B, K, H, W]`, and optionally the write-back of every compressed map into the reference's SQLite `embeddings`
table (`imagescry_amd.storage.write_embeddings`, the format of storage/models.py:104-129) instead of returning it.
"""

from __future__ import annotations

from os import PathLike
from typing import Iterable, Sequence

import torch

from imagescry_amd import storage
from imagescry_amd.data import EmbeddingBatch, ImageBatch
from imagescry_amd.decomposition import PCA
from imagescry_amd.embedding import EmbeddingModule

__all__ = ["EmbeddingPCAPipeline"]


class EmbeddingPCAPipeline:
    """Embeds images and projects every embedding vector onto the fitted principal components."""

    def __init__(
        self,
        *,
        embedding_model: EmbeddingModule,
        pca: PCA,
        db: "str | PathLike | None" = None,
        image_ids: Sequence[int] | None = None,
        pca_checkpoint_id: int | None = None,
    ) -> None:
        if not pca.fitted:  # reference: pipelines.py:49-50
            raise ValueError("PCA model must be fitted before it can be used in the pipeline.")
        if db is not None and (image_ids is None or pca_checkpoint_id is None):  # reference: pipelines.py:52-55
            raise ValueError("If a database is provided, both `image_ids` and `pca_checkpoint_id` must be provided.")
        self.embedding_model = embedding_model
        self.pca = pca
        self.db = db
        self.image_ids = torch.tensor(list(image_ids or []), dtype=torch.int64)
        self.pca_checkpoint_id = pca_checkpoint_id

    def predict_step(self, batch: ImageBatch) -> "EmbeddingBatch | list[int]":
        """Compressed embedding maps of the batch -- or, with a database, the ids of the rows they were stored in
        (reference: pipelines.py:63-97)."""
        batch_size = len(batch)
        full = self.embedding_model.predict_step(batch)
        flat = self.pca.transform(full.get_flat_vectors())
        compressed = flat.reshape(batch_size, *full.spatial_dims, self.pca.num_components).permute(0, 3, 1, 2)
        if self.db is None:
            return EmbeddingBatch(indices=batch.indices, embeddings=compressed)
        batch_image_ids = self.image_ids[batch.indices.cpu()].tolist()
        host = compressed.float().cpu()  # device -> host boundary, as reference pipelines.py:91-93
        return storage.write_embeddings(
            self.db, [(image_id, host[i].contiguous()) for i, image_id in enumerate(batch_image_ids)],
            checkpoint_id=self.pca_checkpoint_id,
        )

    def predict(self, dataloader: Iterable[ImageBatch]) -> "list[EmbeddingBatch] | list[int]":
        """One compressed `EmbeddingBatch` per input batch in loader order, or the flattened list of stored row ids
        (reference: pipelines.py:99-131)."""
        device = self.embedding_model.device
        results = [self.predict_step(batch.to(device)) for batch in dataloader]
        if self.db is None:
            return results  # type: ignore[return-value]
        return [row_id for ids in results for row_id in ids]  # type: ignore[union-attr]
