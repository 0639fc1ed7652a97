"""Same-shape batching: the reference's sampler test
(tests/test_data.py:141-170, shapes of tests/test_data.py:25-40) against this build's `SimilarShapeBatcher`."""

from __future__ import annotations

import pytest
import torch

from imagescry_amd import ImageTensorDataset, SimilarShapeBatcher

SHAPES = [(7, 7), (7, 8), (8, 8), (8, 8), (7, 7), (5, 7), (8, 8), (8, 7), (8, 7), (7, 7), (7, 7), (2, 2), (3, 2), (2, 2)]


def _dataset() -> ImageTensorDataset:
    g = torch.Generator().manual_seed(1234)
    return ImageTensorDataset([torch.randint(0, 255, (3, *s), dtype=torch.uint8, generator=g) for s in SHAPES])


@pytest.mark.parametrize("max_batch_size", [1, 2, 3, 4])
def test_dataloader_similar_shape_batcher(max_batch_size: int) -> None:
    dataset = _dataset()
    observed: set[int] = set()
    for batch in dataset.get_loader(max_batch_size=max_batch_size):
        assert len(batch) <= max_batch_size
        shapes = {tuple(img.shape[-2:]) for img in batch.images}
        assert len(shapes) == 1
        index_list = batch.indices.tolist()
        assert [SHAPES[i] for i in index_list] == [tuple(img.shape[-2:]) for img in batch.images]
        for i, img in zip(index_list, batch.images):
            assert torch.equal(img, dataset.images[i])
        observed.update(index_list)
    assert observed == set(range(len(SHAPES)))


def test_batches_follow_sorted_shape_groups() -> None:
    batches = list(SimilarShapeBatcher(SHAPES, 2))
    assert batches[0] == [11, 13] and batches[1] == [12] and batches[2] == [5]  # (2,2) x2, (3,2), (5,7)
    assert batches[3:5] == [[0, 4], [9, 10]]  # the four (7,7) images, in dataset order, two per batch
    assert sum(len(b) for b in batches) == len(SHAPES) and len(SimilarShapeBatcher(SHAPES, 2)) == len(batches)
    with pytest.raises(ValueError):
        SimilarShapeBatcher(SHAPES, 0)
    with pytest.raises(TypeError):
        ImageTensorDataset([torch.zeros(3, 4, 4)])
