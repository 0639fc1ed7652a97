"""The sharded search end to end on the GPU, two ranks (rendezvous on 127.0.0.1).

The backend follows the box: with at least as many devices as ranks it is `nccl` -- RCCL over xGMI, ONE RANK PER DEVICE,
the collectives of the product path (`all_gather_into_tensor` on device buffers from the bank's exchange stream) as the
8-GPU run issues them; on the single-GPU test box both ranks share cuda:0 and the exchange goes through `gloo` with host
copies.  Everything else is the product path either way: isc_cosine_topk per shard with its index_base, one all-gather
of the exchange buffers, isc_topk_merge reading them in place.  The assertions are the same for both backends; the
backend that ran is recorded in every rank's result file and checked against what the box offers.  The parent never
touches the GPU (`device_count` does not initialise it); the ranks are `mp.spawn` children."""

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


def _expected_backend(world: int) -> str:
    return "nccl" if torch.cuda.device_count() >= world else "gloo"


def _join(rank: int, world: int, port: int) -> torch.device:
    """Rendezvous of one rank; returns its device.  nccl (RCCL): rank r owns device r; gloo: the ranks share cuda:0."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    backend = _expected_backend(world)
    device = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(device)
    dist.init_process_group(backend, rank=rank, world_size=world)
    return device


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    device = _join(rank, world, port)
    try:
        import cases
        from imagescry_amd import EmbeddingBank

        results = {"backend": np.array(dist.get_backend())}
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
            assert str(got["backend"]) == _expected_backend(world)
            np.testing.assert_array_equal(got[f"{name}_i"], exp_i)
            np.testing.assert_allclose(got[f"{name}_s"], exp_s, rtol=0, atol=1e-6)


def _pipeline_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    """BASELINE config 5 at test size, the product path end to end on the HIP device: every rank encodes ITS OWN batches
    with the ViT-B/16 kernels (depth 1), the embeddings are all-gathered, every rank searches all of them against its
    shard of a 30 011-row fp16 bank (isc_cosine_topk with index_base), the partial results are exchanged on the bank's
    exchange stream and merged (isc_topk_merge).  Two streams + two collectives per step, three steps, so the
    collectives of consecutive steps interleave."""
    device = _join(rank, world, port)
    try:
        from imagescry_amd import EmbeddingBank, EmbedSearchPipeline, ImageBatch, ViTB16Embedder, vit

        cfg = vit.ViTConfig(depth=1)
        model = ViTB16Embedder(config=cfg, state_dict=vit.make_state_dict(cfg, seed=1, randomize_affine=True)).to(device)
        rows = torch.nn.functional.normalize(torch.randn(30_011, 768, generator=torch.Generator().manual_seed(9)), dim=1)
        sharded = EmbeddingBank(rows.to(device), dtype=torch.float16, normalize=False, process_group=dist.group.WORLD)
        whole = EmbeddingBank(rows.to(device), dtype=torch.float16, normalize=False)  # the unsharded answer, same process
        g = torch.Generator().manual_seed(100 + rank)  # different images on every rank
        batches = [ImageBatch(indices=torch.arange(6) + 6 * b + 1000 * rank,
                              images=torch.randint(0, 256, (6, 3, 224, 224), dtype=torch.uint8, generator=g))
                   for b in range(3)]
        runs = {}
        for name, overlap in (("overlap", True), ("serial", False)):
            pipe = EmbedSearchPipeline(embedding_model=model, bank=sharded, k=10, overlap=overlap)
            runs[name] = pipe.run(batches)
            assert len(runs[name]) == 3
            assert int(pipe.exact_pass_queries.item()) >= 0
        out = {}
        for bi, batch in enumerate(batches):
            q = model.predict_step(batch.to(device)).get_flat_vectors().half()
            ref_s, ref_i = whole.search(q, 10)
            for name in runs:
                r = runs[name][bi]
                assert torch.equal(r.indices.cpu(), batch.indices), (name, bi)
                assert r.scores.shape == (6, 10)
                assert torch.equal(r.neighbours, ref_i) and torch.equal(r.scores, ref_s), (name, bi)
            out[f"q{bi}"] = q.cpu().numpy()
            out[f"s{bi}"] = runs["overlap"][bi].scores.cpu().numpy()
            out[f"i{bi}"] = runs["overlap"][bi].neighbours.cpu().numpy()
        out["bank"] = whole.bank.cpu().numpy()
        out["backend"] = np.array(dist.get_backend())
        np.savez(os.path.join(out_dir, f"pipe{rank}.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_two_rank_embed_search_pipeline(tmp_path: Path) -> None:
    from oracle import search_oracle

    world = 2
    mp.spawn(_pipeline_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        got = np.load(tmp_path / f"pipe{rank}.npz")
        assert str(got["backend"]) == _expected_backend(world)
        bank = torch.from_numpy(got["bank"])
        for bi in range(3):
            exp_s, exp_i = search_oracle.cosine_topk(bank, torch.from_numpy(got[f"q{bi}"]), 10)
            np.testing.assert_array_equal(got[f"i{bi}"], exp_i)
            np.testing.assert_allclose(got[f"s{bi}"], exp_s, rtol=0, atol=1e-5)
    # the two ranks encoded different images
    assert not np.array_equal(np.load(tmp_path / "pipe0.npz")["q0"], np.load(tmp_path / "pipe1.npz")["q0"])


def _stream_worker(rank: int, world: int, port: int, out_dir: str) -> None:
    """A stream of `search_async` calls with the handles resolved late (three in flight over two exchange buffers): the
    answers must equal the synchronous ones, and every search must issue exactly one collective."""
    device = _join(rank, world, port)
    try:
        import cases
        from imagescry_amd import EmbeddingBank

        bank, _ = cases.search_case(20_003, 128, 1, torch.float16, seed=31)
        calls = {"gather": 0}

        class Counting(EmbeddingBank):
            def _all_gather_bytes(self, xbuf):
                calls["gather"] += 1
                return super()._all_gather_bytes(xbuf)

        eb = Counting(bank.to(device), dtype=torch.float16, normalize=False, process_group=dist.group.WORLD)
        qs = [torch.randn(q, 128, generator=torch.Generator().manual_seed(50 + j)).half().to(device)
              for j, q in enumerate((40, 7, 40, 130, 7))]
        want = [eb.search(q, 10) for q in qs]
        assert calls["gather"] == len(qs)
        handles = [eb.search_async(q, 10) for q in qs]  # nothing resolved until all are enqueued
        assert calls["gather"] == 2 * len(qs)
        for (ws, wi), h in zip(want, handles):
            s, i = h.result()
            assert torch.equal(s, ws) and torch.equal(i, wi)
            assert h.gathered_status.shape == (world, 4)
        np.savez(os.path.join(out_dir, f"stream{rank}.npz"), ok=np.ones(1), backend=np.array(dist.get_backend()))
    finally:
        dist.destroy_process_group()


def test_two_rank_search_stream_with_late_handles(tmp_path: Path) -> None:
    mp.spawn(_stream_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        assert str(np.load(tmp_path / f"stream{rank}.npz")["backend"]) == _expected_backend(2)
