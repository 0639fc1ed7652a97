"""`EmbedSearchPipeline` on the GPU (BASELINE config 5 shape of work, small sizes): the two-stream overlapped run, the
single-stream run and the step-by-step path (`predict_step` then a checked `EmbeddingBank.search`) must agree
bit for bit, and the neighbours must equal the oracle's for the embeddings the GPU produced."""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import search_oracle
from props import assert_topk_properties

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


def test_config5_at_its_own_size(device):
    """BASELINE config 5 at ITS OWN size on one GPU: the ViT-B/16 embedder at depth 12 (fp16 operands), three DIFFERENT
    batches of 512 images, pipelined on two HIP streams (`EmbedSearchPipeline(overlap=True)`) into a 50 000 000 x 768 fp16
    bank (76.8 GB packed beside the 76.8 GB of row-major rows the proof reads: 154 of the 288 GB).  No CPU oracle finishes
    this size, so every batch's answer is PROVEN on the device with the three size-independent properties -- sorted under
    the total order, scores = float32(float64 cosine) of the returned rows, exactly k - 1 rows of the 50 M rank before the
    k-th entry -- for the embeddings the GPU produced; the bank is `bench.make_shard`'s (one seed per 2^20-row block, the
    bank `bench.py --workload pipeline` searches).  `last_status[:2] == [0, 0]`: no candidate buffer overflowed and no
    query needed the second pass on this iid bank.  Falls back to the largest bank that fits when the device has less
    memory free than the full size needs (the size used is asserted to be the configuration's on an MI355X)."""
    import bench
    from imagescry_amd import EmbeddingBank, EmbedSearchPipeline, ImageBatch, ViTB16Embedder

    n, d, b, k = 50_000_000, 768, 512, 10
    free = torch.cuda.mem_get_info(device)[0]
    need = 2 * n * d * 2 + (24 << 30)  # rows + packed bank + encoder activations / workspaces / proof blocks
    if free < need:  # not an MI355X-sized device: the largest multiple of 2^20 rows that fits
        n = max(1 << 20, int((free - (24 << 30)) // (2 * d * 2)) >> 20 << 20)
    rows = bench.make_shard(0, n, d, device)
    bank = EmbeddingBank(rows, dtype=torch.float16, normalize=False)
    model = ViTB16Embedder(seed=0).to(device)
    assert model.config.depth == 12 and model.embedding_dim == d
    g = torch.Generator().manual_seed(4321)
    batches = [ImageBatch(indices=torch.arange(b) + b * j,
                          images=torch.randint(0, 256, (b, 3, 224, 224), dtype=torch.uint8, generator=g)) for j in range(3)]
    pipe = EmbedSearchPipeline(embedding_model=model, bank=bank, k=k, overlap=True)
    results = pipe.run(batches)
    assert len(results) == 3
    assert bank.last_status.cpu().tolist()[:2] == [0, 0]
    assert int(pipe.exact_pass_queries.item()) == 0
    seen = []
    for batch, res in zip(batches, results):
        assert torch.equal(res.indices.cpu(), batch.indices)
        assert res.scores.shape == (b, k) and res.neighbours.shape == (b, k)
        # the queries the search saw: the embedder's float32 vectors rounded to the bank dtype (isc_cosine_topk's q_dtype)
        q = model.predict_step(batch.to(device)).get_flat_vectors().to(torch.float16)
        assert_topk_properties(rows, q, res.scores, res.neighbours, k, block=1 << 18)
        seen.append(res.neighbours[:, 0].clone())
    assert not torch.equal(seen[0], seen[1])  # different images, different neighbours
    if torch.cuda.get_device_properties(device).total_memory >= 250 << 30:
        assert n == 50_000_000
