"""Reading (and writing) the reference's on-disk embedding format, to connect the search to real data.

The reference keeps embeddings in the `embeddings` table of `<dir>/imagescry.db` (SQLite;
src/imagescry/storage/database.py:38,54): one row per image with the columns `id`, `checkpoint_id`, `image_id`,
`embedding_dim`, `embedding_height`, `embedding_width` and `embedding_data`, the raw little-endian float32 bytes of
the `[C, H, W]` tensor (`tensor.numpy().tobytes()`, src/imagescry/storage/models.py:73-129).  This module reads
that table with the standard library's `sqlite3` -- none of the reference's ORM stack is needed -- and hands the
rows to `EmbeddingBank`.  Host-side I/O only; the arithmetic stays in the HIP kernels.
"""

from __future__ import annotations

import sqlite3
from dataclasses import dataclass
from os import PathLike
from pathlib import Path
from typing import Iterable, Sequence

import numpy as np
import torch
from torch import Tensor

DATABASE_NAME = "imagescry.db"  # reference: database.py:38

_SCHEMA = """
CREATE TABLE IF NOT EXISTS embeddings (
    id INTEGER NOT NULL PRIMARY KEY,
    checkpoint_id INTEGER,
    image_id INTEGER NOT NULL,
    embedding_dim INTEGER NOT NULL,
    embedding_height INTEGER NOT NULL,
    embedding_width INTEGER NOT NULL,
    embedding_data BLOB
)
"""


@dataclass(frozen=True)
class StoredEmbedding:
    """One row of the `embeddings` table (reference: storage/models.py:73-102)."""

    id: int
    image_id: int
    checkpoint_id: int | None
    tensor: Tensor  # float32 [C, H, W]


def database_path(db: str | PathLike) -> Path:
    """A directory (the reference's `db_dir`) or the SQLite file itself."""
    p = Path(db)
    return p / DATABASE_NAME if p.is_dir() else p


def _decode(dim: int, height: int, width: int, blob: bytes) -> Tensor:
    """reference: `Embedding.embedding_tensor`, storage/models.py:94-102."""
    if dim <= 0 or height <= 0 or width <= 0:
        raise ValueError(f"invalid embedding dimensions ({dim}, {height}, {width})")
    arr = np.frombuffer(blob, dtype="<f4")
    if arr.size != dim * height * width:
        raise ValueError(f"embedding_data holds {arr.size} floats, expected {dim}x{height}x{width}")
    return torch.from_numpy(arr.reshape(dim, height, width).astype(np.float32, copy=True))


def read_embeddings(
    db: str | PathLike, *, image_ids: Sequence[int] | None = None, embedding_ids: Sequence[int] | None = None
) -> list[StoredEmbedding]:
    """Rows of the `embeddings` table.

    With `image_ids` the result follows the order of the ids and skips images without an embedding, as
    `get_embeddings_by_image_id` does (reference: storage/operations.py:108-144); otherwise table (`id`) order.
    """
    path = database_path(db)
    if not path.exists():
        raise FileNotFoundError(path)
    if image_ids is not None and len(image_ids) == 0:
        raise ValueError("image_ids cannot be empty")
    con = sqlite3.connect(f"file:{path}?mode=ro", uri=True)
    try:
        rows = con.execute(
            "SELECT id, image_id, checkpoint_id, embedding_dim, embedding_height, embedding_width, embedding_data "
            "FROM embeddings ORDER BY id"
        ).fetchall()
    finally:
        con.close()
    records = [StoredEmbedding(r[0], r[1], r[2], _decode(r[3], r[4], r[5], r[6])) for r in rows]
    if embedding_ids is not None:
        wanted = set(embedding_ids)
        records = [r for r in records if r.id in wanted]
    if image_ids is not None:
        by_image = {r.image_id: r for r in records}
        records = [by_image[i] for i in image_ids if i in by_image]
        if not records:
            raise RuntimeError(f"No embeddings found for image IDs: {list(image_ids)}")
    return records


def write_embeddings(
    db: str | PathLike, items: Iterable[tuple[int, Tensor]], *, checkpoint_id: int | None = None
) -> list[int]:
    """Append `(image_id, float32 [C, H, W] tensor)` pairs in the reference's format (`Embedding.create`,
    storage/models.py:104-129); returns the new row ids.  Creates the table if the file is new."""
    path = database_path(db)
    con = sqlite3.connect(path)
    ids: list[int] = []
    try:
        con.execute(_SCHEMA)
        for image_id, tensor in items:
            if tensor.ndim != 3 or tensor.dtype != torch.float32:
                raise TypeError("embedding tensors must be float32 [C, H, W]")
            c, h, w = tensor.shape
            blob = tensor.detach().cpu().contiguous().numpy().astype("<f4", copy=False).tobytes()
            cur = con.execute(
                "INSERT INTO embeddings (checkpoint_id, image_id, embedding_dim, embedding_height, embedding_width, "
                "embedding_data) VALUES (?, ?, ?, ?, ?, ?)",
                (checkpoint_id, int(image_id), c, h, w, blob),
            )
            ids.append(int(cur.lastrowid))
        con.commit()
    finally:
        con.close()
    return ids


def stack_padded(records: Sequence[StoredEmbedding]) -> tuple[Tensor, Tensor]:
    """`(ids int64 [N], embeddings float32 [N, E, Hmax, Wmax])`, every map zero-padded at the bottom / right to the
    largest height and width -- the collation `StoredEmbeddingsDataset` performs (reference: data.py:351-399)."""
    if not records:
        raise ValueError("no embeddings")
    dims = {r.tensor.shape[0] for r in records}
    if len(dims) != 1:
        raise ValueError(f"embeddings have different channel counts: {sorted(dims)}")
    hmax = max(r.tensor.shape[1] for r in records)
    wmax = max(r.tensor.shape[2] for r in records)
    out = torch.zeros((len(records), dims.pop(), hmax, wmax), dtype=torch.float32)
    for i, r in enumerate(records):
        out[i, :, : r.tensor.shape[1], : r.tensor.shape[2]] = r.tensor
    return torch.tensor([r.id for r in records], dtype=torch.int64), out


def flat_rows(records: Sequence[StoredEmbedding]) -> tuple[Tensor, Tensor]:
    """Every spatial location of every stored map as one bank row, in (record, h, w) order -- the
    `get_flat_vectors` order of reference data.py:112-118 -- plus its origin `int64 [N, 3]` = (image_id, h, w)."""
    if not records:
        raise ValueError("no embeddings")
    rows, origin = [], []
    for r in records:
        c, h, w = r.tensor.shape
        rows.append(r.tensor.permute(1, 2, 0).reshape(-1, c))
        hh, ww = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        origin.append(torch.stack([torch.full((h * w,), r.image_id), hh.reshape(-1), ww.reshape(-1)], dim=1))
    dims = {t.shape[1] for t in rows}
    if len(dims) != 1:
        raise ValueError(f"embeddings have different channel counts: {sorted(dims)}")
    return torch.cat(rows), torch.cat(origin).to(torch.int64)
