"""imagescry_amd -- the embed-and-search hot path of libertininick/imagescry on AMD MI355X (gfx950).

Public surface (same names as the reference where the reference has them):

* `ImageBatch`, `EmbeddingBatch`                 -- reference src/imagescry/data.py:29-144
* `resize`, `normalize_per_channel`, `to_4d`     -- reference src/imagescry/image/transforms.py
* `EmbeddingModule`, `EfficientNetEmbedder`,
  `ResNet50Embedder`, `ViTB16Embedder`           -- reference src/imagescry/models/embedding.py:27-183
* `PCA`, `EmbeddingPCAPipeline`                  -- reference src/imagescry/models/decomposition.py, pipelines.py
* `EmbedSearchPipeline`                          -- encode -> search on two HIP streams (BASELINE config 5; new)
* `EmbeddingBank`                                -- cosine top-k search (new; see search.py)

All arithmetic runs in hand-written HIP kernels behind the C ABI of include/imagescry_hip.h;
PyTorch is used for device memory, streams and `torch.distributed` only.
"""

from imagescry_amd.batching import ImageTensorDataset, SimilarShapeBatcher
from imagescry_amd.data import EmbeddingBatch, ImageBatch
from imagescry_amd.decomposition import PCA
from imagescry_amd.embedding import (
    EfficientNetEmbedder,
    EmbeddingModule,
    ResNet50Embedder,
    ViTB16Embedder,
    l2_normalize_channels,
)
from imagescry_amd.pipelines import EmbeddingPCAPipeline, EmbedSearchPipeline, SearchResult
from imagescry_amd.search import EmbeddingBank, SearchHandle, shard_bounds
from imagescry_amd.transforms import normalize_per_channel, resize, to_4d

__all__ = [
    "EfficientNetEmbedder",
    "EmbeddingBank",
    "EmbeddingBatch",
    "EmbeddingModule",
    "EmbeddingPCAPipeline",
    "EmbedSearchPipeline",
    "SearchHandle",
    "SearchResult",
    "PCA",
    "ResNet50Embedder",
    "ViTB16Embedder",
    "l2_normalize_channels",
    "ImageBatch",
    "ImageTensorDataset",
    "SimilarShapeBatcher",
    "normalize_per_channel",
    "resize",
    "shard_bounds",
    "to_4d",
]
__version__ = "0.1.0"
