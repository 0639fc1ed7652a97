"""The multi-rank search path on CPU: world_size 2 (and 3), gloo backend, 127.0.0.1 rendezvous.

The HIP kernels cannot run here, so the two device hooks of `EmbeddingBank` (`_store`, `_local_topk`,
`_merge_topk`) are replaced by the oracle in a test-only subclass; everything else -- the row sharding, the
index_base arithmetic, padding of short shards, the packed all-gather and the merge call -- is the product code.
"""

from __future__ import annotations

import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n: int, k: int, out_dir: str) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cases
        from imagescry_amd import EmbeddingBank
        from oracle import search_oracle

        class OracleBank(EmbeddingBank):
            def _store(self, embeddings, normalize):  # keep the rows on the CPU
                return embeddings.contiguous()

            def _local_topk(self, queries, kk, out=None, lane=-1, stream=None):  # `out`: the product's exchange buffer, filled in place
                s, i = search_oracle.cosine_topk(self._bank, queries, kk, index_base=self.index_base)
                s, i = torch.from_numpy(s), torch.from_numpy(i)
                if out is not None:
                    out[0].copy_(s), out[1].copy_(i), out[2].zero_()
                return s, i

            def _merge_topk(self, scores, indices, kk):
                s, i = search_oracle.topk_merge(scores.numpy(), indices.numpy(), kk)
                return torch.from_numpy(s), torch.from_numpy(i)

        bank, queries = cases.search_case(n, 64, 9, torch.float16, seed=5)
        queries[2] = 0  # all-tie row
        eb = OracleBank(bank, dtype=torch.float16, normalize=False, process_group=dist.group.WORLD)
        lo, hi = rank * n // world, (rank + 1) * n // world
        assert eb.index_base == lo and len(eb) == hi - lo
        scores, indices = eb.search(queries, k)
        # presharded construction gives the same answer
        eb2 = OracleBank(bank[lo:hi], dtype=torch.float16, normalize=False, process_group=dist.group.WORLD,
                         presharded=True, index_base=lo)
        s2, i2 = eb2.search(queries, k)
        assert torch.equal(indices, i2) and torch.equal(scores, s2)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), scores=scores.numpy(), indices=indices.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,k", [(2, 1001, 10), (3, 7, 5)])
def test_sharded_search_equals_unsharded(world: int, n: int, k: int, tmp_path: Path) -> None:
    import cases
    from oracle import search_oracle

    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, k, str(tmp_path)), nprocs=world, join=True)
    bank, queries = cases.search_case(n, 64, 9, torch.float16, seed=5)
    queries[2] = 0
    exp_s, exp_i = search_oracle.cosine_topk(bank, queries, k)
    for rank in range(world):  # every rank holds the full merged answer
        got = np.load(tmp_path / f"rank{rank}.npz")
        np.testing.assert_array_equal(got["indices"], exp_i)
        np.testing.assert_array_equal(got["scores"], exp_s)


def _uneven_worker(rank: int, world: int, port: int, k: int, out_dir: str) -> None:
    """Presharded, very uneven shards: rank 0 holds 3 rows (fewer than k), rank 1 the rest.  Every rank must issue
    the same single collective (the first version of this path let the short shard return after one all-gather while
    a full shard could enter a second one)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cases
        from imagescry_amd import EmbeddingBank
        from oracle import search_oracle

        calls = {"gather": 0}

        class OracleBank(EmbeddingBank):
            def _store(self, embeddings, normalize):
                return embeddings.contiguous()

            def _local_topk(self, queries, kk, out=None, lane=-1, stream=None):  # `out`: the product's exchange buffer, filled in place
                s, i = search_oracle.cosine_topk(self._bank, queries, kk, index_base=self.index_base)
                s, i = torch.from_numpy(s), torch.from_numpy(i)
                if out is not None:
                    out[0].copy_(s), out[1].copy_(i), out[2].zero_()
                return s, i

            def _merge_topk(self, scores, indices, kk):
                s, i = search_oracle.topk_merge(scores.numpy(), indices.numpy(), kk)
                return torch.from_numpy(s), torch.from_numpy(i)

            def _all_gather_bytes(self, xbuf):
                calls["gather"] += 1
                return super()._all_gather_bytes(xbuf)

        n = 3000
        bank, queries = cases.search_case(n, 32, 6, torch.float16, seed=17)
        t = torch.linspace(0.0, 1.0, n)[:, None]  # ordered: later rows are closer to query 0
        bank = torch.nn.functional.normalize(bank.float() * (1 - t) + queries[0].float()[None, :] * t, dim=1).half()
        lo, hi = (0, 3) if rank == 0 else (3, n)
        eb = OracleBank(bank[lo:hi], dtype=torch.float16, normalize=False, process_group=dist.group.WORLD,
                        presharded=True, index_base=lo)
        for _ in range(2):
            scores, indices = eb.search(queries, k)
        assert calls["gather"] == 2  # one collective per search on every rank
        handles = [eb.search_async(queries, k) for _ in range(3)]  # the pipelined form: still one collective each
        assert calls["gather"] == 5
        for h in handles:
            s2, i2 = h.result()
            assert torch.equal(s2, scores) and torch.equal(i2, indices)
            assert h.gathered_status.shape == (world, 4)
        exp_s, exp_i = search_oracle.cosine_topk(bank, queries, k)
        np.testing.assert_array_equal(indices.numpy(), exp_i)
        np.testing.assert_array_equal(scores.numpy(), exp_s)
    finally:
        dist.destroy_process_group()


def test_uneven_presharded_shards_stay_in_step(tmp_path: Path) -> None:
    mp.spawn(_uneven_worker, args=(2, _free_port(), 5, str(tmp_path)), nprocs=2, join=True)


def _pipeline_worker(rank: int, world: int, port: int, n: int, k: int, out_dir: str) -> None:
    """`EmbedSearchPipeline` over a 2-way sharded bank: each rank 'encodes' its own batches (a test double stands in
    for the HIP embedder: fixed random projection of the mean pixel rows), the embeddings are all-gathered, searched
    against every shard and each rank keeps the rows of its own batches."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cases
        from imagescry_amd import EmbeddingBank, EmbeddingBatch, EmbeddingModule, EmbedSearchPipeline, ImageBatch
        from oracle import search_oracle

        class OracleBank(EmbeddingBank):
            def _store(self, embeddings, normalize):
                return embeddings.contiguous()

            def _local_topk(self, queries, kk, out=None, lane=-1, stream=None):  # `out`: the product's exchange buffer, filled in place
                s, i = search_oracle.cosine_topk(self._bank, queries, kk, index_base=self.index_base)
                s, i = torch.from_numpy(s), torch.from_numpy(i)
                if out is not None:
                    out[0].copy_(s), out[1].copy_(i), out[2].zero_()
                return s, i

            def _merge_topk(self, scores, indices, kk):
                s, i = search_oracle.topk_merge(scores.numpy(), indices.numpy(), kk)
                return torch.from_numpy(s), torch.from_numpy(i)

        class StubEmbedder(EmbeddingModule):
            proj = torch.randn(3 * 8, 64, generator=torch.Generator().manual_seed(3))

            def preprocess(self, images):
                return images.float()

            def forward(self, x):
                return (x.mean(dim=3).flatten(1) @ self.proj)[:, :, None, None]

            @property
            def embedding_dim(self):
                return 64

            def predict_step(self, batch):
                e = torch.nn.functional.normalize(self.forward(self.preprocess(batch.images)), dim=1)
                return EmbeddingBatch(indices=batch.indices, embeddings=e)

        bank, _ = cases.search_case(n, 64, 4, torch.float16, seed=8)
        eb = OracleBank(bank, dtype=torch.float16, normalize=False, process_group=dist.group.WORLD)
        g = torch.Generator().manual_seed(100 + rank)
        batches = [ImageBatch(indices=torch.arange(5) + 10 * b + 100 * rank,
                              images=torch.randint(0, 256, (5, 3, 8, 6), dtype=torch.uint8, generator=g)) for b in range(3)]
        model = StubEmbedder()
        results = EmbedSearchPipeline(embedding_model=model, bank=eb, k=k).run(batches)
        assert len(results) == 3
        for b, r in zip(batches, results):
            assert torch.equal(r.indices, b.indices) and r.scores.shape == (5, k)
            q = model.predict_step(b).get_flat_vectors().half()
            exp_s, exp_i = search_oracle.cosine_topk(bank, q, k)  # unsharded answer for this rank's own images
            np.testing.assert_array_equal(r.neighbours.numpy(), exp_i)
            np.testing.assert_array_equal(r.scores.numpy(), exp_s)
    finally:
        dist.destroy_process_group()


def test_embed_search_pipeline_two_ranks(tmp_path: Path) -> None:
    mp.spawn(_pipeline_worker, args=(2, _free_port(), 333, 7, str(tmp_path)), nprocs=2, join=True)
