"""The sharded search end to end on the GPU: two ranks (gloo rendezvous on 127.0.0.1, both on cuda:0 -- the
single-GPU box has no second device; the exchange goes through host copies, everything else is the product path:
isc_cosine_topk per shard with its index_base, one all-gather of the exchange buffers, isc_topk_merge reading
them in place)."""

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

pytestmark = pytest.mark.gpu


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import cases
        from imagescry_amd import EmbeddingBank

        device = torch.device("cuda:0")
        results = {}
        for name, n, q, k, dtype in (("big", 30011, 40, 10, torch.float16), ("tiny", 5, 3, 4, torch.float32)):
            bank, queries = cases.search_case(n, 128, q, dtype, seed=21)
            eb = EmbeddingBank(bank.to(device), dtype=dtype, normalize=False, process_group=dist.group.WORLD)
            s, i = eb.search(queries.to(device), k)
            results[f"{name}_s"], results[f"{name}_i"] = s.cpu().numpy(), i.cpu().numpy()
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **results)
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_search_matches_oracle(tmp_path: Path) -> None:
    import cases
    from oracle import search_oracle

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for name, n, q, k, dtype in (("big", 30011, 40, 10, torch.float16), ("tiny", 5, 3, 4, torch.float32)):
        bank, queries = cases.search_case(n, 128, q, dtype, seed=21)
        exp_s, exp_i = search_oracle.cosine_topk(bank, queries, k)
        for rank in range(world):
            got = np.load(tmp_path / f"rank{rank}.npz")
            np.testing.assert_array_equal(got[f"{name}_i"], exp_i)
            np.testing.assert_allclose(got[f"{name}_s"], exp_s, rtol=0, atol=1e-6)
