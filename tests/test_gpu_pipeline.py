"""`EmbedSearchPipeline` on the GPU (BASELINE config 5 shape of work, small sizes): the two-stream overlapped run, the
single-stream run and the step-by-step path (`predict_step` then a checked `EmbeddingBank.search`) must agree
bit for bit, and the neighbours must equal the oracle's for the embeddings the GPU produced."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import search_oracle

pytestmark = pytest.mark.gpu


def test_overlapped_pipeline_equals_sequential(device):
    from imagescry_amd import EmbeddingBank, EmbedSearchPipeline, ImageBatch, ViTB16Embedder, vit

    cfg = vit.ViTConfig(depth=1)
    model = ViTB16Embedder(config=cfg, state_dict=vit.make_state_dict(cfg, seed=1, randomize_affine=True)).to(device)
    g = torch.Generator().manual_seed(9)
    rows = torch.nn.functional.normalize(torch.randn(30_000, 768, generator=g), dim=1)
    bank = EmbeddingBank(rows.to(device), dtype=torch.float16, normalize=False)
    batches = [ImageBatch(indices=torch.arange(6) + 6 * b,
                          images=torch.randint(0, 256, (6, 3, 224, 224), dtype=torch.uint8, generator=g)) for b in range(4)]
    overlapped = EmbedSearchPipeline(embedding_model=model, bank=bank, k=10).run(batches)
    serial = EmbedSearchPipeline(embedding_model=model, bank=bank, k=10, overlap=False).run(batches)
    assert len(overlapped) == len(serial) == 4
    stored = bank.bank.cpu()
    for b, a, s in zip(batches, overlapped, serial):
        emb = model.predict_step(b.to(device))
        q = emb.get_flat_vectors().half()
        ref_s, ref_i = bank.search(q, 10)
        for r in (a, s):
            assert torch.equal(r.indices.cpu(), b.indices)
            assert torch.equal(r.neighbours, ref_i) and torch.equal(r.scores, ref_s)
        exp_s, exp_i = search_oracle.cosine_topk(stored, q.cpu(), 10)
        np.testing.assert_array_equal(a.neighbours.cpu().numpy(), exp_i)
        np.testing.assert_allclose(a.scores.cpu().numpy(), exp_s, rtol=0, atol=1e-5)


def test_pipeline_argument_checks(device):
    from imagescry_amd import EmbeddingBank, EmbedSearchPipeline, ViTB16Embedder, vit

    model = ViTB16Embedder(config=vit.ViTConfig(depth=1)).to(device)
    bank = EmbeddingBank(torch.randn(64, 32, device=device))
    with pytest.raises(ValueError):
        EmbedSearchPipeline(embedding_model=model, bank=bank)  # 768-d embedder, 32-d bank
    bank = EmbeddingBank(torch.randn(64, 768, device=device))
    with pytest.raises(ValueError):
        EmbedSearchPipeline(embedding_model=model, bank=bank, k=0)
    assert EmbedSearchPipeline(embedding_model=model, bank=bank).run([]) == []
