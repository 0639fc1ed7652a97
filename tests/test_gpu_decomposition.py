"""GPU parity of `PCA.transform` (C ABI: isc_linear_centered) and of `EmbeddingPCAPipeline.predict_step` against
the oracle's restatement of the reference (decomposition.py:78-91, pipelines.py:63-86)."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import cases  # noqa: E402

from oracle import decomposition_oracle, encoder_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,f,kmax", [(1000, 4, None), (777, 768, 64), (300, 1280, 37), (5, 96, 3)])
def test_transform_matches_oracle(n: int, f: int, kmax, device: torch.device) -> None:
    from imagescry_amd import PCA

    g = cases.gen(n + f)
    # correlated features with a large common offset: centring BEFORE the product matters for the rounding
    x = torch.randn(n, f, generator=g) @ torch.randn(f, f, generator=g) * 0.1 + 50.0
    pca = PCA(max_num_components=kmax, min_explained_variance=1.0).fit(x).to(device)
    ref = decomposition_oracle.fit(x, max_num_components=kmax, min_explained_variance=1.0)
    assert pca.num_components == ref.num_components
    got = pca.transform(x.to(device)).cpu()
    exp = ref.transform(x)
    assert got.shape == exp.shape and got.dtype == torch.float32
    scale = float(exp.abs().max())
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=0, atol=2e-5 * max(scale, 1.0))
    with pytest.raises(ValueError):
        pca.transform(x[:, :-1].to(device))
    with pytest.raises(Exception):
        pca.transform(x)  # CPU tensor: no fallback


def test_reference_decorrelation_property_on_gpu(device: torch.device) -> None:
    """tests/test_models/test_decomposition.py:78-82 with the projection done by the HIP kernel."""
    from imagescry_amd import PCA
    from torch.distributions import MultivariateNormal

    torch.manual_seed(1234)
    cov = torch.tensor([[1.0, 0.5, 0.0, 0.0], [0.5, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, -0.5], [0.0, 0.0, -0.5, 1.0]])
    x = MultivariateNormal(loc=torch.tensor([0.0, 1.0, -1.0, 0.0]), covariance_matrix=cov).sample((1000,))
    pca = PCA(min_explained_variance=0.8).fit(x.to(device))  # fit on a device tensor moves the model there
    projected = pca.transform(x.to(device)).cpu()
    assert projected.shape == (1000, 3)
    corr = torch.abs(torch.corrcoef(projected.T))
    assert torch.all(torch.tril(corr, diagonal=-1) <= 1e-4)


def test_pipeline_predict_step(device: torch.device) -> None:
    from imagescry_amd import EmbeddingPCAPipeline, ImageBatch, PCA, ResNet50Embedder, resnet50

    sd = resnet50.make_state_dict(seed=2, randomize_bn=True)
    model = ResNet50Embedder(state_dict=sd).to(device)
    fit_images = cases.images_u8((24, 3, 64, 64), seed=31)
    emb = model.predict_step(ImageBatch(indices=torch.arange(24), images=fit_images).to(device))
    with pytest.raises(ValueError):
        EmbeddingPCAPipeline(embedding_model=model, pca=PCA())
    pca = PCA(max_num_components=8, min_explained_variance=1.0).fit(emb.get_flat_vectors())
    pipe = EmbeddingPCAPipeline(embedding_model=model, pca=pca)
    images = cases.images_u8((3, 3, 64, 64), seed=32)
    batch = ImageBatch(indices=torch.tensor([4, 2, 9]), images=images)
    out = pipe.predict_step(batch.to(device))
    assert out.embeddings.shape == (3, 8, 1, 1) and out.indices.cpu().tolist() == [4, 2, 9]
    full = encoder_oracle.predict_step_embeddings(images, sd)
    ref = decomposition_oracle.FittedPCA(pca.feature_means.cpu(), pca.explained_variance.cpu(), pca.component_vectors.cpu())
    exp = ref.transform(full.permute(0, 2, 3, 1).reshape(-1, 768)).reshape(3, 1, 1, 8).permute(0, 3, 1, 2)
    np.testing.assert_allclose(out.embeddings.cpu().numpy(), exp.numpy(), rtol=0, atol=2e-5)
    results = pipe.predict([batch, batch])
    assert len(results) == 2 and torch.equal(results[0].embeddings, out.embeddings)


def test_variable_size_images_through_the_pipeline_with_write_back(tmp_path, device: torch.device) -> None:
    """N4: images of three different sizes -> same-shape batches -> resize branch (long side > max_side_length) ->
    embed -> PCA -> rows in the reference's SQLite format; read back, they equal the direct computation."""
    from imagescry_amd import EmbeddingPCAPipeline, ImageTensorDataset, PCA, ResNet50Embedder, resnet50, storage

    sd = resnet50.make_state_dict(seed=4, randomize_bn=True)
    model = ResNet50Embedder(state_dict=sd, max_side_length=64).to(device)
    sizes = [(70, 50), (40, 40), (70, 50), (100, 60), (40, 40), (70, 50)]
    images = [cases.images_u8((3, *s), seed=40 + i) for i, s in enumerate(sizes)]
    dataset = ImageTensorDataset(images)
    fit_rows = torch.cat([model.predict_step(b.to(device)).get_flat_vectors() for b in dataset.get_loader(4)])
    pca = PCA(max_num_components=4, min_explained_variance=1.0).fit(fit_rows)
    with pytest.raises(ValueError):
        EmbeddingPCAPipeline(embedding_model=model, pca=pca, db=tmp_path)
    image_ids = [100 + i for i in range(len(images))]
    pipe = EmbeddingPCAPipeline(embedding_model=model, pca=pca, db=tmp_path, image_ids=image_ids, pca_checkpoint_id=3)
    row_ids = pipe.predict(dataset.get_loader(2))
    assert sorted(row_ids) == list(range(1, 7))
    stored = {r.image_id: r for r in storage.read_embeddings(tmp_path)}
    assert set(stored) == set(image_ids) and all(r.checkpoint_id == 3 for r in stored.values())
    plain = EmbeddingPCAPipeline(embedding_model=model, pca=pca)
    for batch in dataset.get_loader(2):
        out = plain.predict_step(batch.to(device))
        for j, idx in enumerate(batch.indices.tolist()):
            assert torch.equal(stored[100 + idx].tensor, out.embeddings[j].cpu())
            assert stored[100 + idx].tensor.shape == (4, 1, 1)
    # oracle check of one resized batch (70x50 > 64 takes the resize branch, batch statistics over the pair)
    pair = torch.stack([images[0], images[2]])
    full = encoder_oracle.predict_step_embeddings(pair, sd, 64)
    ref = decomposition_oracle.FittedPCA(pca.feature_means.cpu(), pca.explained_variance.cpu(), pca.component_vectors.cpu())
    exp = ref.transform(full.flatten(1))
    got = torch.stack([stored[100].tensor, stored[102].tensor]).flatten(1)
    np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=0, atol=2e-5)
