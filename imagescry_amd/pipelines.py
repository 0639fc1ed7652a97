"""Embed-then-compress pipeline (the caller of the hot path).

Mirrors the arithmetic of the reference's `EmbeddingPCAPipeline.predict_step` / `predict`
(src/imagescry/models/pipelines.py:22-131): embedder.predict_step -> flat vectors -> PCA.transform -> back to
`[B, K, H, W]`.  The reference's optional write-back into its SQLite store is out of scope (SURVEY.md section 2
rows 7-9); passing a database manager is therefore not supported here.
"""

from __future__ import annotations

from typing import Iterable

from imagescry_amd.data import EmbeddingBatch, ImageBatch
from imagescry_amd.decomposition import PCA
from imagescry_amd.embedding import EmbeddingModule

__all__ = ["EmbeddingPCAPipeline"]


class EmbeddingPCAPipeline:
    """Embeds images and projects every embedding vector onto the fitted principal components."""

    def __init__(self, *, embedding_model: EmbeddingModule, pca: PCA) -> None:
        if not pca.fitted:  # reference: pipelines.py:49-50
            raise ValueError("PCA model must be fitted before it can be used in the pipeline.")
        self.embedding_model = embedding_model
        self.pca = pca

    def predict_step(self, batch: ImageBatch) -> EmbeddingBatch:
        """reference: pipelines.py:63-86."""
        batch_size = len(batch)
        full = self.embedding_model.predict_step(batch)
        flat = self.pca.transform(full.get_flat_vectors())
        compressed = flat.reshape(batch_size, *full.spatial_dims, self.pca.num_components).permute(0, 3, 1, 2)
        return EmbeddingBatch(indices=batch.indices, embeddings=compressed)

    def predict(self, dataloader: Iterable[ImageBatch]) -> list[EmbeddingBatch]:
        """One compressed `EmbeddingBatch` per input batch, in loader order (reference: pipelines.py:99-131)."""
        device = self.embedding_model.device
        return [self.predict_step(batch.to(device)) for batch in dataloader]
