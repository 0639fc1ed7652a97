"""Callers of the hot path: embed-then-compress (the reference's pipeline) and embed-then-search (BASELINE config 5).

Mirrors the arithmetic of the reference's `EmbeddingPCAPipeline.predict_step` / `predict`
(src/imagescry/models/pipelines.py:22-131): embedder.predict_step -> flat vectors -> PCA.transform -> back to
`[B, K, H, W]`, and optionally the write-back of every compressed map into the reference's SQLite `embeddings`
table (`imagescry_amd.storage.write_embeddings`, the format of storage/models.py:104-129) instead of returning it.
"""

from __future__ import annotations

from dataclasses import dataclass
from os import PathLike
from typing import Iterable, Sequence

import torch
import torch.distributed as dist
from torch import Tensor

from imagescry_amd import storage
from imagescry_amd.data import EmbeddingBatch, ImageBatch
from imagescry_amd.decomposition import PCA
from imagescry_amd.embedding import EmbeddingModule
from imagescry_amd.search import EmbeddingBank, SearchHandle

__all__ = ["EmbedSearchPipeline", "EmbeddingPCAPipeline", "SearchResult"]


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


@dataclass(frozen=True)
class SearchResult:
    """Top-k neighbours of the images of one batch: `indices` are the batch's own `ImageBatch.indices`,
    `scores` float32 `[R, k]` / `neighbours` int64 `[R, k]` one row per embedding vector (R = B for a `[B, E, 1, 1]`
    embedder, B * h * w for a spatial one, in `get_flat_vectors()` order)."""

    indices: Tensor
    scores: Tensor
    neighbours: Tensor


class EmbedSearchPipeline:
    """Encode image batches and search their embeddings in a bank, the encode of batch i + 1 overlapped with the
    search of batch i on two HIP streams (BASELINE.json configs[4]: "ViT-B/16 fp16 encode pipelined into 50M x 768
    sharded search, encode/search overlapped on HIP streams").

    The reference has neither step combined (its caller of the embedder is `EmbeddingPCAPipeline`, pipelines.py:22-131);
    the loop below is that class's `predict` with the search in place of the PCA.

    One process per GPU.  With a row-sharded bank (`bank.process_group` set) every rank encodes its OWN batches
    (replicas -- a batch is never split, its normalisation statistics are batch-wide), the embeddings of all ranks are
    all-gathered, every rank searches all of them against its shard, and the partial results are exchanged and merged
    by `EmbeddingBank.search`; each rank returns the rows of its own batch.  All ranks must feed the same number of
    batches with the same number of embedding rows.

    Ordering between the two streams is by events only; the host never waits inside the loop, and there is nothing to
    check afterwards: `EmbeddingBank.search` is final on the device (queries the float32 filter cannot prove are redone
    exactly by the same call).
    """

    def __init__(self, *, embedding_model: EmbeddingModule, bank: EmbeddingBank, k: int = 10, overlap: bool = True) -> None:
        if not isinstance(k, int) or isinstance(k, bool) or k < 1:
            raise ValueError(f"k must be a positive int, got {k!r}")
        if embedding_model.embedding_dim != bank.dim:
            raise ValueError(f"embedder produces {embedding_model.embedding_dim}-d vectors, the bank holds {bank.dim}-d rows")
        self.embedding_model = embedding_model
        self.bank = bank
        self.k = k
        self.overlap = overlap
        self.exact_pass_queries: Tensor | None = None

    def _queries(self, emb: EmbeddingBatch) -> Tensor:
        """This rank's flat vectors -- gathered over the ranks of a sharded bank.  One GPU: exactly what the reference's
        `get_flat_vectors` returns (float32, data.py:112-118); `isc_cosine_topk` rounds them to the bank dtype while it
        packs them (`q_dtype`), no cast kernel runs.  Sharded: they are cast to the bank dtype BEFORE the all-gather, which
        halves the bytes every rank sends over xGMI for an fp16 bank (the search would round them to it anyway)."""
        group = self.bank.process_group
        if group is None:
            return emb.get_flat_vectors()
        q = emb.get_flat_vectors().to(self.bank.dtype).contiguous()
        world = dist.get_world_size(group)
        on_host = dist.get_backend(group) == "gloo" and q.device.type != "cpu"
        src = q.cpu() if on_host else q
        gathered = torch.empty((world * src.shape[0], src.shape[1]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(gathered, src, group=group)
        return gathered.to(q.device)

    def _own_rows(self, t: Tensor, rows: int) -> Tensor:
        group = self.bank.process_group
        if group is None:
            return t
        r = dist.get_rank(group)
        return t[r * rows : (r + 1) * rows]

    def run(self, dataloader: Iterable[ImageBatch]) -> list[SearchResult]:
        """One `SearchResult` per input batch, in loader order.

        Streams (overlap=True): batch i is encoded, and its embeddings all-gathered, on the encode stream; its local
        search runs on the search stream behind an event; for a sharded bank the exchange of the partial results
        (all-gather + merge) runs on the bank's own exchange stream (`EmbeddingBank.search_async`), i.e. under the
        local search of batch i + 1 and the encode of batch i + 2.  Handle i - 1 is resolved right after search i has been
        enqueued (an event wait on the caller's stream, no host synchronisation), so at most two searches are unresolved
        and device memory does not grow with the number of batches (a handle pins its gathered exchange buffers)."""
        device = self.embedding_model.device
        use_streams = self.overlap and device.type == "cuda"
        if use_streams:
            enc_stream, search_stream = torch.cuda.Stream(device), torch.cuda.Stream(device)
            enc_stream.wait_stream(torch.cuda.current_stream(device))
            search_stream.wait_stream(torch.cuda.current_stream(device))
        results: list[SearchResult] = []
        pending: tuple[Tensor, int, SearchHandle] | None = None

        def resolve(item: tuple[Tensor, int, SearchHandle]) -> None:
            indices, rows, handle = item
            scores, neighbours = handle.result()  # orders the current stream behind the search / exchange stream
            results.append(SearchResult(indices=indices, scores=self._own_rows(scores, rows),
                                        neighbours=self._own_rows(neighbours, rows)))

        for batch in dataloader:
            batch = batch.to(device)
            if use_streams:
                enc_stream.wait_stream(torch.cuda.current_stream(device))  # the host-to-device copy of this batch
                batch.images.record_stream(enc_stream)
                with torch.cuda.stream(enc_stream):
                    emb = self.embedding_model.predict_step(batch)
                    q = self._queries(emb)
                    ready = torch.cuda.Event()
                    ready.record(enc_stream)
                    rows = emb.get_flat_vectors().shape[0]
                with torch.cuda.stream(search_stream):
                    search_stream.wait_event(ready)
                    q.record_stream(search_stream)
                    handle = self.bank.search_async(q, self.k)
            else:
                emb = self.embedding_model.predict_step(batch)
                q = self._queries(emb)
                rows = emb.get_flat_vectors().shape[0]
                handle = self.bank.search_async(q, self.k)
                if self.bank.process_group is not None:
                    handle.result()  # one stream: exchange i before search i + 1
            if pending is not None:
                resolve(pending)
            pending = (batch.indices, rows, handle)
        if use_streams:
            torch.cuda.current_stream(device).wait_stream(enc_stream)
            torch.cuda.current_stream(device).wait_stream(search_stream)
        if pending is not None:
            resolve(pending)
        self.exact_pass_queries = self._exact_pass_counter()
        return results

    def _exact_pass_counter(self) -> Tensor | None:
        """Diagnostics of the LAST search of the run, still on the device (reading it is the caller's synchronisation):
        how many of its queries the float32 filter could not prove and the exact pass answered -- summed over the
        shards of a sharded bank.  A duplicate-heavy bank sends many queries there."""
        bank = self.bank
        if bank.process_group is not None and bank.last_gathered_status is not None:
            return bank.last_gathered_status[:, 1].sum()
        return None if bank.last_status is None else bank.last_status[1]
