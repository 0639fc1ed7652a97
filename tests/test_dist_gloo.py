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

            def _local_topk(self, queries, kk, check, out=None):  # `out`: the product's exchange buffer (unused here)
                s, i = search_oracle.cosine_topk(self._bank, queries, kk, index_base=self.index_base)
                return torch.from_numpy(s), torch.from_numpy(i)

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
