"""Brute-force cosine top-k over an embedding bank resident in HBM.

The reference stores embeddings as SQLite BLOB rows (src/imagescry/storage/models.py:73-129) and reads them
back with `get_embeddings_by_image_id` (src/imagescry/storage/operations.py:108-144); it has no search step
(SURVEY.md section 0 fact 2).  `EmbeddingBank` is the storage-flavoured object BASELINE.json's `north_star`
asks for: it keeps the `[N, D]` bank on the GPU and answers `search(queries, k)`.

Semantics (pinned by oracle/search_oracle.py, not by the reference):
`score = float32(dot_f64(q, b) / max(||q||, 1e-12))`, bank rows used as stored (L2-normalised once at build
time with the `F.normalize` formula of src/imagescry/models/embedding.py:74), results ordered by
(score descending, row index ascending).  Row ids are positions in the bank, the analogue of the reference's
DB row order (operations.py:135-144).

The matrix-core pass is a float32 filter; the candidates are re-scored in float64, a rounding-error guard proves per
query that the filter lost nothing, and what it cannot prove is searched again exactly on the device -- one call, no
host synchronisation, the result is final (csrc/cosine_topk.hip).  The bank is stored in a fixed pseudo-random row
order (`isc_bank_permutation`), so banks whose rows arrive sorted or clustered -- the reference's store returns all
cells of one image adjacent, operations.py:135-144 -- behave like shuffled ones; indices are always original rows.

Multi-GPU: one process per GPU.  The bank is row-sharded -- rank r holds rows `[r*N//G, (r+1)*N//G)` -- every
rank searches its shard with the replicated queries, one all-gather (RCCL over xGMI) exchanges the
`Q x k x 12 B` partial results and every rank merges them; the merge order is total, so the answer does not
depend on G.  The exchange is software-pipelined: the local search runs on the caller's stream, the all-gather and
the merge on the bank's own exchange stream (ordered by events, two exchange buffers), and `search_async` returns as
soon as the merge is enqueued -- in a stream of searches the exchange of search i runs under the local kernels of
search i + 1.
"""

from __future__ import annotations

import math
from os import PathLike
from typing import Sequence

import torch
import torch.distributed as dist
from torch import Tensor

from imagescry_amd import _lib
from imagescry_amd.data import EmbeddingBatch

__all__ = ["EmbeddingBank", "SearchHandle", "shard_bounds"]

_PAD_INDEX = torch.iinfo(torch.int64).max


def shard_bounds(n_rows: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous row range `[lo, hi)` of `rank` when `n_rows` are split over `world_size` ranks."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError(f"invalid rank {rank} for world size {world_size}")
    return rank * n_rows // world_size, (rank + 1) * n_rows // world_size


# The two streams asynchronous searches alternate between, per DEVICE and shared by every bank on it: the runtime maps
# streams onto a handful of hardware queues, and streams that share a queue wait for each other -- with a pair per bank, a
# process holding two banks ran its searches slower asynchronously than serially (0.366 against 0.321 ms at a 1.25 M-row
# shard).
_LANE_STREAMS: dict[int, tuple["torch.cuda.Stream", "torch.cuda.Stream"]] = {}


def _lane_streams(device: torch.device) -> tuple["torch.cuda.Stream", "torch.cuda.Stream"]:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pair = _LANE_STREAMS.get(idx)
    if pair is None:
        pair = (torch.cuda.Stream(device), torch.cuda.Stream(device))
        _LANE_STREAMS[idx] = pair
    return pair


class SearchHandle:
    """Result of `EmbeddingBank.search_async`.  The tensors exist at once; their CONTENTS are final when the event
    recorded behind the merge has fired.  `result()` orders the caller's current stream behind that event (no host
    synchronisation) and returns `(scores, indices)`."""

    __slots__ = ("_scores", "_indices", "_event", "gathered_status")

    def __init__(self, scores: Tensor, indices: Tensor, event: "torch.cuda.Event | None" = None,
                 gathered_status: Tensor | None = None) -> None:
        self._scores = scores
        self._indices = indices
        self._event = event
        # [G, 4] diagnostics of every shard: a view into the gathered exchange buffers, which it keeps alive (they were
        # allocated on the exchange stream and are read only there, so nothing else has to hold them)
        self.gathered_status = gathered_status

    def result(self) -> tuple[Tensor, Tensor]:
        """May be called more than once and from different streams: EVERY call orders the then-current stream behind the
        answer and tells the allocator about that stream (events are cheap; a stream already ordered waits for nothing)."""
        if self._event is not None:
            cur = torch.cuda.current_stream(self._scores.device)
            cur.wait_event(self._event)
            for t in (self._scores, self._indices, self.gathered_status):  # allocated on another stream, used on this one
                if t is not None:
                    t.record_stream(cur)
        return self._scores, self._indices


class _ExchangeSlot:
    """One of the two exchange buffers of a sharded bank and the event behind the last exchange that read it."""

    __slots__ = ("buf", "done")

    def __init__(self) -> None:
        self.buf: Tensor | None = None
        self.done: "torch.cuda.Event | None" = None


class EmbeddingBank:
    """`[N, D]` embedding bank on one GPU (or one row shard of it per rank) with cosine top-k search.

    Args:
        embeddings: floating `[N, D]` tensor on a HIP device.  With `process_group` set and
            `presharded=False` it is the FULL bank (identical on every rank) and this rank keeps only its
            rows; with `presharded=True` it is this rank's shard and `index_base` is the global index of
            its first row.
        dtype: storage dtype of the bank, `torch.float16` (default) or `torch.float32`.
        normalize: L2-normalise every row before storing it (skip for rows that are already unit length,
            e.g. the output of `Embedder.predict_step`).
        index_base: global row index of local row 0 (only with `presharded=True`).
        process_group: the `torch.distributed` group the bank is sharded over (`None` = single GPU).
    """

    def __init__(
        self,
        embeddings: Tensor,
        *,
        dtype: torch.dtype = torch.float16,
        normalize: bool = True,
        index_base: int = 0,
        process_group: dist.ProcessGroup | None = None,
        presharded: bool = False,
    ) -> None:
        if not isinstance(embeddings, Tensor) or not embeddings.dtype.is_floating_point:
            raise TypeError("embeddings must be a floating point torch.Tensor")
        if embeddings.ndim != 2:
            raise ValueError(f"embeddings must have shape [N, D], got {tuple(embeddings.shape)}")
        if dtype not in (torch.float16, torch.float32):
            raise ValueError(f"bank dtype must be float16 or float32, got {dtype}")
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group) if process_group is not None else 1
        self.rank = dist.get_rank(process_group) if process_group is not None else 0
        if process_group is not None and not presharded:
            lo, hi = shard_bounds(embeddings.shape[0], self.world_size, self.rank)
            embeddings = embeddings[lo:hi]
            index_base = lo
        elif process_group is None and index_base != 0 and not presharded:
            raise ValueError("index_base is only meaningful for a presharded bank")
        self.index_base = int(index_base)
        self.dtype = dtype
        self.dim = int(embeddings.shape[1])
        self.num_local_rows = int(embeddings.shape[0])
        if self.dim == 0:
            raise ValueError("embedding dimension must be positive")
        self._bank = self._store(embeddings, normalize)
        # search workspaces per LANE: -1 = the caller's stream (`search`), 0 / 1 = the two streams `search_async` alternates
        # between -- two searches in flight must not share a workspace
        self._workspaces: dict[int, dict[tuple[int, int], Tensor]] = {}
        self._lane_next = 0
        self._slots = (_ExchangeSlot(), _ExchangeSlot())
        self._slot_next = 0
        self._xstream: "torch.cuda.Stream | None" = None
        self.last_status: Tensor | None = None
        self.last_gathered_status: Tensor | None = None
        self.row_origin: Tensor | None = None  # set by from_database: (image_id, h, w) of every row

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_batches(cls, batches: Sequence[EmbeddingBatch], **kwargs: object) -> "EmbeddingBank":
        """Bank whose rows are `get_flat_vectors()` of every batch, in order (reference: data.py:112-118).

        `predict_step` output is already L2-normalised per location, so `normalize` defaults to False here.
        """
        if len(batches) == 0:
            raise ValueError("need at least one EmbeddingBatch")
        rows = torch.cat([b.get_flat_vectors() for b in batches], dim=0)
        kwargs.setdefault("normalize", False)
        return cls(rows, **kwargs)  # type: ignore[arg-type]

    @classmethod
    def from_database(
        cls,
        db: "str | PathLike",
        *,
        device: str | torch.device = "cuda",
        image_ids: Sequence[int] | None = None,
        **kwargs: object,
    ) -> "EmbeddingBank":
        """Bank built from the reference's SQLite store (`<dir>/imagescry.db`, table `embeddings`): every spatial
        location of every stored `[C, H, W]` map becomes one row, in (record, h, w) order.  `row_origin`
        (`int64 [N, 3]` = image_id, h, w) maps result indices back to images.  Stored maps are PCA-compressed,
        i.e. not unit length, so rows are L2-normalised unless `normalize=False` is passed.
        (reference format: storage/models.py:73-129; read order: storage/operations.py:108-144)"""
        from imagescry_amd import storage

        rows, origin = storage.flat_rows(storage.read_embeddings(db, image_ids=image_ids))
        bank = cls(rows.to(device), **kwargs)  # type: ignore[arg-type]
        bank.row_origin = origin
        return bank

    def _store(self, embeddings: Tensor, normalize: bool) -> Tensor:
        """Row-normalise / cast the rows and write them into the PACKED bank image (`isc_bank_pack`):
        `[tile of 256 rows][K step][row][128 B]`, rows permuted, the layout the search kernels stream
        (include/imagescry_hip.h).  `_norm_bound` receives an upper bound of the stored rows' norms (the search's
        rounding guard needs it)."""
        _lib.require_device(embeddings, "embeddings")
        n, d = embeddings.shape
        lib = _lib.load()
        code = _lib.dtype_code(self.dtype)
        self._norm_bound = torch.zeros(1, dtype=torch.float32, device=embeddings.device)
        if n == 0:
            return torch.empty(0, dtype=torch.uint8, device=embeddings.device)
        need = _lib.c_size_t()
        _lib.check(lib.isc_bank_packed_bytes(code, n, d, need), "isc_bank_packed_bytes")
        packed = torch.empty(need.value, dtype=torch.uint8, device=embeddings.device)
        tile_bytes = need.value // ((n + 255) // 256)
        packed[-tile_bytes:].zero_()  # padding rows of the last tile (isc_bank_pack writes real rows only)
        if embeddings.dtype not in (torch.float16, torch.float32):
            embeddings = embeddings.float()
        block = 1 << 20
        with torch.cuda.device(embeddings.device):
            for r0 in range(0, n, block):
                rows = embeddings[r0 : r0 + block]
                if rows.stride(1) != 1:
                    rows = rows.contiguous()
                st = lib.isc_bank_pack(
                    rows.data_ptr(), _lib.dtype_code(rows.dtype), rows.shape[0], d, rows.stride(0), r0, n,
                    int(normalize), 1e-12, packed.data_ptr(), code, self._norm_bound.data_ptr(),
                    _lib.stream_handle(embeddings.device),
                )
                _lib.check(st, "isc_bank_pack")
        return packed

    # ------------------------------------------------------------------ properties
    @property
    def device(self) -> torch.device:
        return self._bank.device

    @property
    def bank(self) -> Tensor:
        """The stored rows as a row-major `[N_local, D]` tensor of the bank dtype (unpacked copy, `isc_bank_unpack`)."""
        out = torch.empty((self.num_local_rows, self.dim), dtype=self.dtype, device=self.device)
        if self.num_local_rows:
            lib = _lib.load()
            with torch.cuda.device(self.device):
                st = lib.isc_bank_unpack(
                    self._bank.data_ptr(), _lib.dtype_code(self.dtype), self.dim, self.num_local_rows, 0,
                    self.num_local_rows, out.data_ptr(), self.dim, _lib.stream_handle(self.device),
                )
            _lib.check(st, "isc_bank_unpack")
        return out

    def __len__(self) -> int:
        return self.num_local_rows

    # ------------------------------------------------------------------ search
    def _prepare_queries(self, queries: Tensor) -> Tensor:
        if not isinstance(queries, Tensor) or not queries.dtype.is_floating_point:
            raise TypeError("queries must be a floating point torch.Tensor")
        if queries.ndim != 2 or queries.shape[1] != self.dim:
            raise ValueError(f"queries must have shape [Q, {self.dim}], got {tuple(queries.shape)}")
        if queries.device != self.device:
            raise ValueError(f"queries are on {queries.device} but the bank is on {self.device}")
        # float16 and float32 queries go to the library as they are: `isc_cosine_topk` rounds them to the bank dtype while
        # it packs them (`q_dtype`; float32 -> fp16 round to nearest even, the arithmetic of `Tensor.to(float16)`), so the
        # reference-shaped call `bank.search(predict_step(batch).get_flat_vectors())` -- float32 vectors, data.py:112-118 --
        # runs no cast kernel.  Other floating types are converted here, directly to the bank dtype.
        if queries.dtype not in (torch.float16, torch.float32):
            queries = queries.to(self.dtype)
        return queries if queries.stride(1) == 1 or queries.shape[0] == 0 else queries.contiguous()

    def _workspace(self, n_queries: int, k: int, lane: int = -1) -> Tensor:
        """The search workspace.  The C side runs a call as passes of at most `ISC_SEARCH_PASS_QUERIES` queries over
        one workspace and cuts the queries into ONE tile of 64 (Q <= 64) or 128 (Q <= 128), or tiles of 256, so the size depends on
        (padded queries of a pass, k) only: alternating batch sizes inside one bucket -- a pipeline's short last
        batch -- reuse one allocation instead of reallocating 150-300 MB per call.  One buffer per bucket and lane is kept."""
        nq = min(n_queries, _lib.ISC_SEARCH_PASS_QUERIES)
        key = (-(-nq // 64) * 64 if nq <= 128 else -(-nq // 256) * 256, k)
        cache = self._workspaces.setdefault(lane, {})
        ws = cache.get(key)
        if ws is None:
            lib = _lib.load()
            need = _lib.c_size_t()
            st = lib.isc_cosine_topk_workspace_bytes(
                _lib.dtype_code(self.dtype), self.num_local_rows, self.dim, min(key[0], _lib.ISC_SEARCH_PASS_QUERIES), k, need
            )
            _lib.check(st, "isc_cosine_topk_workspace_bytes")
            ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
            if len(cache) >= 4:  # bound what a bank pins: drop the oldest bucket
                cache.pop(next(iter(cache)))
            cache[key] = ws
        return ws

    def _local_topk(
        self, queries: Tensor, k: int, out: tuple[Tensor, Tensor, Tensor] | None = None, lane: int = -1,
        stream: "torch.cuda.Stream | None" = None,
    ) -> tuple[Tensor, Tensor]:
        """Top-k of this rank's rows: `(float32 [Q, k], int64 [Q, k])` with GLOBAL row indices, final when the stream
        has run the call.  `out` optionally supplies the (scores, indices, status int32[4]) tensors to write into (the
        exchange buffer of a sharded search); `lane` / `stream`: the workspace set and the stream of an asynchronous search
        (default: the caller's current stream).  Tensors are allocated on the caller's stream whichever stream computes."""
        nq = queries.shape[0]
        if out is None:
            scores = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            indices = torch.empty((nq, k), dtype=torch.int64, device=self.device)
            status = torch.empty(4, dtype=torch.int32, device=self.device)
        else:
            scores, indices, status = out
        ws = self._workspace(nq, k, lane)
        if stream is not None:
            # every tensor this call touches was allocated on some other stream: tell the allocator the lane uses it, so
            # that nothing handed back early (a dropped handle, a dropped bank, a workspace bucket pushed out of the cache)
            # is given to somebody else before the lane's kernels have run
            for t in (scores, indices, status, ws, self._bank, self._norm_bound):
                t.record_stream(stream)
        lib = _lib.load()
        with torch.cuda.device(self.device):
            st = lib.isc_cosine_topk(
                self._bank.data_ptr(), _lib.dtype_code(self.dtype), self.num_local_rows, self.dim, queries.data_ptr(),
                _lib.dtype_code(queries.dtype), nq, queries.stride(0), k, self.index_base, self._norm_bound.data_ptr(),
                scores.data_ptr(),
                indices.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(),
                stream.cuda_stream if stream is not None else _lib.stream_handle(self.device),
            )
            _lib.check(st, "isc_cosine_topk")
        self.last_status = status
        return scores, indices

    def search_exhaustive(self, queries: Tensor, k: int = 10) -> tuple[Tensor, Tensor]:
        """The same answer from the data-independent float64 kernel (`isc_cosine_topk_exhaustive`): every score of
        every query evaluated exactly.  Slow; the on-device reference the fast path is tested against."""
        q = self._prepare_queries(queries)
        nq = q.shape[0]
        if self.process_group is not None:
            raise ValueError("search_exhaustive answers for one shard; merge the shards with search()")
        if not 1 <= k <= self.num_local_rows:
            raise ValueError(f"k={k} must be in [1, {self.num_local_rows}]")
        lib = _lib.load()
        code = _lib.dtype_code(self.dtype)
        need = _lib.c_size_t()
        _lib.check(
            lib.isc_cosine_topk_exhaustive_workspace_bytes(code, self.num_local_rows, self.dim, nq, k, need),
            "isc_cosine_topk_exhaustive_workspace_bytes",
        )
        ews = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        scores = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        indices = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            st = lib.isc_cosine_topk_exhaustive(
                self._bank.data_ptr(), code, self.num_local_rows, self.dim, q.data_ptr(), _lib.dtype_code(q.dtype), nq,
                q.stride(0), k, self.index_base, scores.data_ptr(), indices.data_ptr(), ews.data_ptr(), ews.numel(),
                _lib.stream_handle(self.device),
            )
        _lib.check(st, "isc_cosine_topk_exhaustive")
        return scores, indices

    def _merge_topk(self, scores: Tensor, indices: Tensor, k: int) -> tuple[Tensor, Tensor]:
        """Merge `[G, Q, kin]` partial results into `[Q, k]` by (score desc, index asc) (`isc_topk_merge`).  The two
        inputs may be strided along G (views into the all-gathered exchange buffer); their `[Q, kin]` blocks are dense."""
        g, nq, kin = scores.shape
        if scores.stride(1) != kin or scores.stride(2) != 1 or indices.stride(1) != kin or indices.stride(2) != 1:
            scores, indices = scores.contiguous(), indices.contiguous()
        out_s = torch.empty((nq, k), dtype=torch.float32, device=scores.device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=scores.device)
        lib = _lib.load()
        with torch.cuda.device(scores.device):
            st = lib.isc_topk_merge(
                scores.data_ptr(), indices.data_ptr(), g, nq, kin, k, scores.stride(0) if g > 1 else 0,
                indices.stride(0) if g > 1 else 0, out_s.data_ptr(), out_i.data_ptr(), _lib.stream_handle(scores.device),
            )
        _lib.check(st, "isc_topk_merge")
        return out_s, out_i

    def _total_rows(self) -> int:
        if self.process_group is None:
            return self.num_local_rows
        counts = [0] * self.world_size
        dist.all_gather_object(counts, self.num_local_rows, group=self.process_group)
        return int(sum(counts))

    def search(self, queries: Tensor, k: int = 10, *, check: bool = True) -> tuple[Tensor, Tensor]:
        """Cosine top-k of every query against the whole (possibly sharded) bank.

        Returns `(scores float32 [Q, k], indices int64 [Q, k])`, best first, ties by lower row index.  The call
        only enqueues work: no host synchronisation, and the result is final -- queries the float32 filter cannot
        prove are redone exactly on the device.  `last_status` (int32[4], device) holds diagnostics: [0] overflowed
        candidate buffers, [1] queries the first pass could not prove (searched again: one more matrix-core pass over the
        bank for all of them together), [3] queries answered by the exhaustive float64 sweep (about one bank sweep per
        four such queries: a bank with thousands of exact copies of a row pays this for queries that hit them).  `check` is accepted for compatibility with the
        first version of this API and ignored.  Everything runs on the caller's current stream (a sharded bank's
        exchange on the bank's exchange stream, ordered behind it).
        """
        del check
        return self._search(queries, k, lanes=False).result()

    def search_async(self, queries: Tensor, k: int = 10) -> SearchHandle:
        """`search` that returns as soon as everything is ENQUEUED; `handle.result()` orders the caller's current stream
        behind the answer.  The local kernels of a search of up to 128 queries run on one of TWO library-owned streams
        of the device, alternately -- the bank keeps a workspace for each -- ordered behind the caller's stream as it
        stands at the call (larger searches stay on the caller's stream); a sharded bank's exchange (all-gather + merge)
        runs on the bank's exchange stream.  A caller that issues search i + 1 before it resolves handle i
        therefore has the short kernels at the end of search i (selection, exact re-score, the empty redo launches) and
        its exchange running beside the first kernels of search i + 1 -- at a 1.25 M-row shard 20 - 25 us of a 320 us
        search (`scripts/two_stream_probe.py`).  At most two searches of one bank should be unresolved at a time (a
        third waits, on the device, for the first one).

        Ownership until `handle.result()`: the search reads `queries` on a library-owned stream (float16 / float32
        queries with unit inner stride are NOT copied), and writes `last_status` / `last_gathered_status` there -- so the
        caller must not overwrite the query tensor in place, nor read those status tensors, on its own stream before it
        has resolved the handle.  `search()` has no such window: everything it does is ordered on the caller's stream."""
        return self._search(queries, k, lanes=True)

    def _lane(self, cur: "torch.cuda.Stream", q: Tensor) -> tuple[int, "torch.cuda.Stream"]:
        """The next of the two search streams, ordered behind everything the caller's stream holds so far."""
        lane = self._lane_next
        self._lane_next ^= 1
        ls = _lane_streams(self.device)[lane]
        ready = torch.cuda.Event()
        ready.record(cur)
        ls.wait_event(ready)
        q.record_stream(ls)  # (possibly a copy made on the caller's stream: keep it until the lane has read it)
        return lane, ls

    # Searches of more queries than this run on the caller's stream even when asynchronous: their kernels fill the GPU for
    # milliseconds, there is nothing at their ends worth overlapping (measured: 1 757 -> 1 749 us at Q = 1024 on a 1.25 M-row
    # shard), and two of them sharing the GPU would stretch each other's launches.
    _LANE_MAX_QUERIES = 128

    def _search(self, queries: Tensor, k: int, lanes: bool) -> SearchHandle:
        if not isinstance(k, int) or isinstance(k, bool):
            raise TypeError(f"k must be an int, got {type(k).__name__}")
        if k < 1:
            raise ValueError(f"k must be >= 1, got {k}")
        if k > _lib.ISC_TOPK_MAX_K:
            raise ValueError(f"k must be <= {_lib.ISC_TOPK_MAX_K}, got {k}")
        q = self._prepare_queries(queries)
        nq = q.shape[0]
        lanes = lanes and nq <= self._LANE_MAX_QUERIES
        if self.process_group is None:
            if k > self.num_local_rows:
                raise ValueError(f"k={k} exceeds the bank size {self.num_local_rows}")
            if nq == 0:
                return SearchHandle(torch.empty((0, k), dtype=torch.float32, device=self.device),
                                    torch.empty((0, k), dtype=torch.int64, device=self.device))
            if not (lanes and self.device.type == "cuda"):
                return SearchHandle(*self._local_topk(q, k))
            lane, ls = self._lane(torch.cuda.current_stream(self.device), q)
            out_s, out_i = self._local_topk(q, k, lane=lane, stream=ls)
            done = torch.cuda.Event()
            done.record(ls)
            return SearchHandle(out_s, out_i, done)

        # ---- sharded: local partial top-k -> ONE all-gather -> merge on every rank.  Every rank issues exactly one
        # collective per search whatever its shard holds, so the ranks cannot fall out of step.
        if not hasattr(self, "_n_total"):
            self._n_total = self._total_rows()
        if k > self._n_total:
            raise ValueError(f"k={k} exceeds the bank size {self._n_total}")
        if nq == 0:
            return SearchHandle(torch.empty((0, k), dtype=torch.float32, device=self.device),
                                torch.empty((0, k), dtype=torch.int64, device=self.device))
        # exchange buffer of this rank: [scores f32 Q*k | indices i64 Q*k | status i32 x4], written in place by the
        # search kernels, gathered with ONE collective and read in place by the merge kernel
        off_i = (4 * nq * k + 7) // 8 * 8
        off_s = off_i + 8 * nq * k
        nbytes = off_s + 16
        on_gpu = self.device.type == "cuda"
        slot = self._slots[self._slot_next]
        self._slot_next ^= 1
        lane, ls = -1, None
        kl = min(k, self.num_local_rows)
        if on_gpu:
            cur = torch.cuda.current_stream(self.device)
            if slot.buf is None or slot.buf.numel() < nbytes:
                if slot.done is not None:  # the old buffer goes back to this stream's pool: nobody may still read it
                    cur.wait_event(slot.done)
                slot.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            if lanes and kl == k:  # (a shard with fewer rows than k is padded with tensor ops on the caller's stream)
                lane, ls = self._lane(cur, q)
                cur = ls  # the stream the local kernels run on
            if slot.done is not None:  # the exchange that last read this buffer (two searches ago)
                cur.wait_event(slot.done)
        elif slot.buf is None or slot.buf.numel() < nbytes:
            slot.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        xbuf = slot.buf[:nbytes]
        part_s = xbuf[: 4 * nq * k].view(torch.float32).view(nq, k)
        part_i = xbuf[off_i:off_s].view(torch.int64).view(nq, k)
        status = xbuf[off_s:].view(torch.int32)
        if kl < k:  # a shard with fewer rows than k: pad with entries that rank after every real candidate
            part_s.fill_(-math.inf)
            part_i.fill_(_PAD_INDEX)
            status.zero_()
            if kl > 0:
                s, i = self._local_topk(q, kl)
                part_s[:, :kl] = s
                part_i[:, :kl] = i
        else:
            self._local_topk(q, k, out=(part_s, part_i, status), lane=lane, stream=ls)

        def exchange() -> tuple[Tensor, Tensor, Tensor, Tensor]:
            gathered = self._all_gather_bytes(xbuf)
            all_s = gathered[:, : 4 * nq * k].view(torch.float32).view(self.world_size, nq, k)
            all_i = gathered[:, off_i:off_s].view(torch.int64).view(self.world_size, nq, k)
            out_s, out_i = self._merge_topk(all_s, all_i, k)
            return out_s, out_i, gathered, gathered[:, off_s:].view(torch.int32)  # [G, 4]: every shard's diagnostics

        if not on_gpu:  # CPU tensors (the gloo rehearsal of the host logic): nothing to overlap
            out_s, out_i, gathered, gstatus = exchange()
            self.last_gathered_status = gstatus
            return SearchHandle(out_s, out_i, None, gstatus)
        if self._xstream is None:
            self._xstream = torch.cuda.Stream(self.device)
        local_done = torch.cuda.Event()
        local_done.record(cur)
        with torch.cuda.stream(self._xstream):
            self._xstream.wait_event(local_done)
            xbuf.record_stream(self._xstream)  # (allocated on the caller's stream, read by the all-gather on this one)
            out_s, out_i, gathered, gstatus = exchange()
            done = torch.cuda.Event()
            done.record(self._xstream)
        slot.done = done
        self.last_gathered_status = gstatus  # valid once the handle has been resolved
        return SearchHandle(out_s, out_i, done, gstatus)

    def _all_gather_bytes(self, xbuf: Tensor) -> Tensor:
        """`[G, nbytes]` uint8: every rank's exchange buffer (one all-gather; RCCL over xGMI on the GPUs)."""
        # a gloo group exchanges host copies (used to rehearse the multi-rank path without RCCL)
        on_host = dist.get_backend(self.process_group) == "gloo" and xbuf.device.type != "cpu"
        src = xbuf.cpu() if on_host else xbuf
        gathered = torch.empty(self.world_size * src.numel(), dtype=torch.uint8, device=src.device)
        dist.all_gather_into_tensor(gathered, src, group=self.process_group)
        return gathered.to(xbuf.device).view(self.world_size, src.numel())
