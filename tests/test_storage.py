"""The reference's on-disk embedding format: float32 `[C, H, W]` blobs in the
`embeddings` table of `imagescry.db` (storage/models.py:73-129).  The reference's own round-trip test
(tests/test_storage/test_models.py:96-136: create -> store -> read back equals the tensor) is mirrored with the
standard-library reader / writer; no GPU needed."""

from __future__ import annotations

import sqlite3
from pathlib import Path

import numpy as np
import pytest
import torch

from imagescry_amd import storage


def _make_db(tmp_path: Path) -> tuple[Path, list[torch.Tensor]]:
    g = torch.Generator().manual_seed(1234)
    tensors = [torch.randn(8, 2, 3, generator=g), torch.randn(8, 1, 1, generator=g), torch.randn(8, 3, 2, generator=g)]
    ids = storage.write_embeddings(tmp_path, [(10, tensors[0]), (11, tensors[1]), (12, tensors[2])], checkpoint_id=7)
    assert ids == [1, 2, 3]
    return tmp_path, tensors


def test_round_trip_and_blob_layout(tmp_path: Path) -> None:
    db, tensors = _make_db(tmp_path)
    assert (db / "imagescry.db").exists()
    records = storage.read_embeddings(db)
    assert [r.id for r in records] == [1, 2, 3] and [r.image_id for r in records] == [10, 11, 12]
    assert all(r.checkpoint_id == 7 for r in records)
    for r, t in zip(records, tensors):
        assert r.tensor.dtype == torch.float32 and torch.equal(r.tensor, t)
    # the blob is exactly `tensor.numpy().tobytes()` (models.py:128): a reader of the reference sees the same bytes
    con = sqlite3.connect(db / "imagescry.db")
    dim, h, w, blob = con.execute(
        "SELECT embedding_dim, embedding_height, embedding_width, embedding_data FROM embeddings WHERE id = 1"
    ).fetchone()
    con.close()
    assert (dim, h, w) == (8, 2, 3) and blob == tensors[0].numpy().tobytes()
    assert np.array_equal(np.frombuffer(blob, dtype=np.float32).reshape(8, 2, 3), tensors[0].numpy())


def test_image_id_order_and_missing_ids(tmp_path: Path) -> None:
    """get_embeddings_by_image_id semantics (operations.py:108-144): query order kept, missing images skipped."""
    db, tensors = _make_db(tmp_path)
    got = storage.read_embeddings(db, image_ids=[12, 99, 10])
    assert [r.image_id for r in got] == [12, 10]
    assert torch.equal(got[0].tensor, tensors[2])
    with pytest.raises(ValueError):
        storage.read_embeddings(db, image_ids=[])
    with pytest.raises(RuntimeError):
        storage.read_embeddings(db, image_ids=[404])
    assert [r.id for r in storage.read_embeddings(db, embedding_ids=[3, 1])] == [1, 3]
    with pytest.raises(FileNotFoundError):
        storage.read_embeddings(tmp_path / "nowhere")


def test_stack_padded_matches_stored_embeddings_dataset(tmp_path: Path) -> None:
    """StoredEmbeddingsDataset zero-pads every map to the largest H and W (data.py:378-399)."""
    db, tensors = _make_db(tmp_path)
    ids, stacked = storage.stack_padded(storage.read_embeddings(db))
    assert ids.tolist() == [1, 2, 3] and stacked.shape == (3, 8, 3, 3)
    assert torch.equal(stacked[0, :, :2, :3], tensors[0]) and float(stacked[0, :, 2, :].abs().max()) == 0.0
    assert torch.equal(stacked[1, :, :1, :1], tensors[1]) and float(stacked[1, :, 1:, :].abs().max()) == 0.0
    exp = torch.nn.functional.pad(tensors[2], (0, 1, 0, 0))
    assert torch.equal(stacked[2], exp)


def test_flat_rows_order(tmp_path: Path) -> None:
    db, tensors = _make_db(tmp_path)
    rows, origin = storage.flat_rows(storage.read_embeddings(db))
    assert rows.shape == (6 + 1 + 6, 8) and origin.shape == (13, 3)
    exp = torch.cat([t.permute(1, 2, 0).reshape(-1, 8) for t in tensors])  # get_flat_vectors order (data.py:118)
    assert torch.equal(rows, exp)
    assert origin[0].tolist() == [10, 0, 0] and origin[5].tolist() == [10, 1, 2] and origin[6].tolist() == [11, 0, 0]
    assert origin[-1].tolist() == [12, 2, 1]


def test_corrupt_blob_is_rejected(tmp_path: Path) -> None:
    storage.write_embeddings(tmp_path, [(1, torch.zeros(4, 1, 1))])
    con = sqlite3.connect(tmp_path / "imagescry.db")
    con.execute("UPDATE embeddings SET embedding_dim = 5")
    con.commit()
    con.close()
    with pytest.raises(ValueError):
        storage.read_embeddings(tmp_path)
    with pytest.raises(TypeError):
        storage.write_embeddings(tmp_path, [(1, torch.zeros(4, 1, 1, dtype=torch.float64))])
