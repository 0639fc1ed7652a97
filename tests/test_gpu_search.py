"""GPU parity of `EmbeddingBank.search` (C ABI: isc_cosine_topk / isc_topk_merge / isc_cosine_topk_exhaustive).

Bar: top-k indices bit-exact against the oracle's total order (score desc, index asc);
scores are the float32 rounding of a float64 evaluation on both sides, compared to 1e-6 -- well inside the
1e-5 (fp32) / 1e-2 (fp16) tolerance `north_star` states.
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
import cases  # noqa: E402

from oracle import search_oracle  # noqa: E402

pytestmark = pytest.mark.gpu

SCORE_ATOL = 1e-6
GOLDEN = np.load(Path(__file__).resolve().parent / "golden" / "search.npz")


def _bank(bank: torch.Tensor, device: torch.device, **kw):
    from imagescry_amd import EmbeddingBank

    return EmbeddingBank(bank.to(device), dtype=bank.dtype, normalize=False, **kw)


def _check(scores: torch.Tensor, indices: torch.Tensor, exp_s: np.ndarray, exp_i: np.ndarray) -> None:
    assert indices.dtype == torch.int64 and scores.dtype == torch.float32
    np.testing.assert_array_equal(indices.cpu().numpy(), exp_i)
    np.testing.assert_allclose(scores.cpu().numpy(), exp_s, rtol=0, atol=SCORE_ATOL)


@pytest.mark.parametrize("name", list(cases.SEARCH_CASES))
def test_search_matches_golden(name: str, device: torch.device) -> None:
    n, d, q, k, dtype = cases.SEARCH_CASES[name]
    bank, queries = cases.search_case(n, d, q, dtype)
    scores, indices = _bank(bank, device).search(queries.to(device), k)
    _check(scores, indices, GOLDEN[f"{name}_scores"], GOLDEN[f"{name}_indices"])


@pytest.mark.parametrize("dtype,tag", [(torch.float16, "f16"), (torch.float32, "f32")])
def test_exact_ties_go_to_the_lowest_index(dtype: torch.dtype, tag: str, device: torch.device) -> None:
    bank, queries = cases.tie_case(dtype)
    scores, indices = _bank(bank, device).search(queries.to(device), 50)
    _check(scores, indices, GOLDEN[f"tie_{tag}_scores"], GOLDEN[f"tie_{tag}_indices"])
    # query 0 is 3 * base[0]: its 40 exact copies (rows 0, 24, 48, ...) lead, in index order
    assert indices[0, :40].cpu().tolist() == list(range(0, 24 * 40, 24))


@pytest.mark.parametrize(
    "n,d,q,k,dtype",
    [
        (256, 768, 256, 10, torch.float32),  # BASELINE config 0 shape: 256 embeddings searched against themselves
        (257, 64, 1, 1, torch.float16),  # one query, k = 1, ragged bank
        (4097, 128, 5, 120, torch.float16),  # largest k, one row past the first level
        (266241, 64, 9, 10, torch.float16),  # three levels (4096 | 262144 | rest)
        (70000, 768, 300, 10, torch.float16),  # two query tiles
        (9000, 1536, 17, 10, torch.float32),
    ],
)
def test_search_matches_oracle(n: int, d: int, q: int, k: int, dtype: torch.dtype, device: torch.device) -> None:
    bank, queries = cases.search_case(n, d, q, dtype, seed=n + q)
    exp_s, exp_i = search_oracle.cosine_topk(bank, queries, k)
    scores, indices = _bank(bank, device).search(queries.to(device), k)
    _check(scores, indices, exp_s, exp_i)


def test_self_search_config0(device: torch.device) -> None:
    """BASELINE config 0: every embedding finds itself at rank 0 with score 1 +- 1e-5."""
    bank, _ = cases.search_case(256, 768, 1, torch.float32)
    eb = _bank(bank, device)
    scores, indices = eb.search(bank.to(device), 10)
    assert indices[:, 0].cpu().tolist() == list(range(256))
    assert torch.allclose(scores[:, 0].cpu(), torch.ones(256), atol=1e-5)


def test_zero_query_and_unpadded_dim(device: torch.device) -> None:
    """A zero query scores 0 against every row -> rows 0..k-1; D = 100 is zero-padded to the kernel's K step."""
    bank, queries = cases.search_case(1000, 100, 4, torch.float16)
    queries[1] = 0
    exp_s, exp_i = search_oracle.cosine_topk(bank, queries, 8)
    from imagescry_amd import EmbeddingBank

    eb = EmbeddingBank(bank.to(device), dtype=torch.float16, normalize=False)
    scores, indices = eb.search(queries.to(device), 8)
    _check(scores, indices, exp_s, exp_i)
    assert indices[1].cpu().tolist() == list(range(8))
    assert scores[1].abs().max().item() == 0.0


def test_bank_normalisation_and_fp32_queries(device: torch.device) -> None:
    """`normalize=True` applies the F.normalize formula before the cast; float32 queries are cast to the bank dtype."""
    g = cases.gen(5)
    raw = torch.randn(3000, 192, generator=g) * 4.0
    queries = torch.randn(6, 192, generator=g)
    from imagescry_amd import EmbeddingBank

    eb = EmbeddingBank(raw.to(device), dtype=torch.float16, normalize=True)
    stored = eb.bank.cpu()
    expect = search_oracle.l2_normalize_rows(raw).half()
    assert (stored.float() - expect.float()).abs().max().item() <= 2.0 ** -11  # at most one fp16 ulp below 1.0
    exp_s, exp_i = search_oracle.cosine_topk(stored, queries.half(), 10)
    scores, indices = eb.search(queries.to(device), 10)
    _check(scores, indices, exp_s, exp_i)


def test_index_base_and_merge_equal_unsharded(device: torch.device) -> None:
    """Searching 4 row shards with their index_base and merging equals the unsharded search (G-independence)."""
    bank, queries = cases.search_case(10000, 256, 20, torch.float16, seed=3)
    k = 10
    exp_s, exp_i = search_oracle.cosine_topk(bank, queries, k)
    from imagescry_amd import EmbeddingBank, shard_bounds

    parts_s, parts_i = [], []
    for r in range(4):
        lo, hi = shard_bounds(10000, 4, r)
        eb = EmbeddingBank(bank[lo:hi].to(device), dtype=torch.float16, normalize=False, index_base=lo, presharded=True)
        s, i = eb.search(queries.to(device), k)
        parts_s.append(s)
        parts_i.append(i)
    ms, mi = eb._merge_topk(torch.stack(parts_s), torch.stack(parts_i), k)
    _check(ms, mi, exp_s, exp_i)


def test_exhaustive_kernel_matches_oracle(device: torch.device) -> None:
    from imagescry_amd import _lib

    for dtype in (torch.float16, torch.float32):
        bank, queries = cases.search_case(5000, 160, 11, dtype, seed=9)
        k = 10
        exp_s, exp_i = search_oracle.cosine_topk(bank, queries, k, index_base=1000)
        eb = _bank(bank, device)
        q = queries.to(device)
        lib = _lib.load()
        need = _lib.c_size_t()
        code = _lib.dtype_code(dtype)
        _lib.check(lib.isc_cosine_topk_exhaustive_workspace_bytes(code, 5000, 160, 11, k, need), "ws")
        ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        s = torch.empty((11, k), dtype=torch.float32, device=device)
        i = torch.empty((11, k), dtype=torch.int64, device=device)
        st = lib.isc_cosine_topk_exhaustive(
            eb._bank.data_ptr(), code, 5000, 160, q.data_ptr(), 11, 160, k, 1000, s.data_ptr(), i.data_ptr(),
            ws.data_ptr(), ws.numel(), _lib.stream_handle(device),
        )
        _lib.check(st, "isc_cosine_topk_exhaustive")
        _check(s, i, exp_s, exp_i)


def test_packed_bank_round_trip(device: torch.device) -> None:
    """isc_bank_pack -> isc_bank_unpack returns the rows bit for bit (ragged N, D not a multiple of the K step)."""
    for dtype in (torch.float16, torch.float32):
        bank, _ = cases.search_case(1234, 100, 1, dtype, seed=4)
        eb = _bank(bank, device)
        assert torch.equal(eb.bank.cpu(), bank)
        assert eb._bank.numel() == 5 * (2 if dtype == torch.float16 else 4) * 256 * 128


def test_overflow_falls_back_to_exhaustive(device: torch.device) -> None:
    """A bank sorted by ascending similarity to the query makes every later row beat the threshold: the
    candidate buffers overflow, status[0] becomes non-zero and `search` reruns on the exhaustive kernel."""
    d = 64
    g = cases.gen(11)
    q = torch.nn.functional.normalize(torch.randn(1, d, generator=g), dim=1)
    noise = torch.nn.functional.normalize(torch.randn(40000, d, generator=g), dim=1)
    t = torch.linspace(0.0, 0.9, 40000)[:, None]
    bank = torch.nn.functional.normalize(t * q + (1 - t) * noise * 0.2, dim=1).half()
    queries = q.half()
    exp_s, exp_i = search_oracle.cosine_topk(bank, queries, 10)
    eb = _bank(bank, device)
    scores, indices = eb.search(queries.to(device), 10)
    _check(scores, indices, exp_s, exp_i)


def test_argument_errors(device: torch.device) -> None:
    bank, queries = cases.search_case(300, 64, 2, torch.float16)
    eb = _bank(bank, device)
    with pytest.raises(ValueError):
        eb.search(queries.to(device), 301)
    with pytest.raises(ValueError):
        eb.search(queries.to(device), 0)
    with pytest.raises(ValueError):
        eb.search(queries[:, :32].to(device), 5)
    with pytest.raises(TypeError):
        eb.search(queries.to(device).to(torch.int32), 5)
    with pytest.raises(ValueError):
        eb.search(queries, 5)  # CPU queries against a GPU bank


@pytest.mark.parametrize("dtype", [torch.float16])
def test_full_size_properties(dtype: torch.dtype, device: torch.device) -> None:
    """BASELINE config 3 shape (1M x 768, 1024 queries, k = 10) checked without a CPU oracle:
    the expected answer is rebuilt on the GPU with torch float64 matmuls (test-side only)."""
    n, d, q, k = 1_000_000, 768, 1024, 10
    g = torch.Generator(device=device).manual_seed(1234)
    bank = torch.nn.functional.normalize(torch.randn(n, d, generator=g, device=device), dim=1).to(dtype)
    queries = torch.randn(q, d, generator=g, device=device).to(dtype)
    from imagescry_amd import EmbeddingBank

    eb = EmbeddingBank(bank, dtype=dtype, normalize=False)
    scores, indices = eb.search(queries, k)
    assert int(eb.last_status[0].item()) == 0
    # sortedness under the total order
    s, i = scores.double(), indices
    assert bool(((s[:, :-1] > s[:, 1:]) | ((s[:, :-1] == s[:, 1:]) & (i[:, :-1] < i[:, 1:]))).all())
    # returned scores are the exact cosine of the returned rows
    q64 = queries.double()
    denom = q64.norm(dim=1).clamp_min(1e-12)
    rows = bank[indices.reshape(-1)].double().reshape(q, k, d)
    exact = (torch.einsum("qkd,qd->qk", rows, q64) / denom[:, None]).float()
    assert torch.allclose(scores, exact, rtol=0, atol=1e-7)
    # nothing outside the returned set beats the k-th score
    kth = scores[:, -1].double()
    better = torch.zeros(q, dtype=torch.int64, device=device)
    for r0 in range(0, n, 65536):
        blk = (q64 @ bank[r0 : r0 + 65536].double().T / denom[:, None]).float().double()
        better += (blk > kth[:, None]).sum(dim=1)
    assert bool((better <= k - 1).all())
    assert bool((better == (scores.double() > kth[:, None]).sum(dim=1)).all())


def test_bank_from_reference_database(tmp_path, device: torch.device) -> None:
    """N2: a bank built from the reference's SQLite `embeddings` table answers like the oracle on the same rows."""
    from imagescry_amd import EmbeddingBank, storage

    g = cases.gen(12)
    maps = [torch.randn(96, 2, 3, generator=g), torch.randn(96, 1, 2, generator=g), torch.randn(96, 3, 3, generator=g)]
    storage.write_embeddings(tmp_path, [(5, maps[0]), (6, maps[1]), (9, maps[2])])
    bank = EmbeddingBank.from_database(tmp_path, device=device, dtype=torch.float32)
    rows = torch.cat([m.permute(1, 2, 0).reshape(-1, 96) for m in maps])
    assert len(bank) == 17 and bank.row_origin.shape == (17, 3)
    stored = bank.bank.cpu()
    assert torch.allclose(stored, search_oracle.l2_normalize_rows(rows), atol=1e-6)
    queries = rows[[3, 7, 16]] + 0.01 * torch.randn(3, 96, generator=g)
    exp_s, exp_i = search_oracle.cosine_topk(stored, queries, 5)
    scores, indices = bank.search(queries.to(device), 5)
    _check(scores, indices, exp_s, exp_i)
    assert indices[:, 0].cpu().tolist() == [3, 7, 16]
    assert bank.row_origin[indices[2, 0].item()].tolist() == [9, 2, 2]


@pytest.mark.parametrize("n", [513, 520, 575, 1025, 4097 + 512 + 7])
def test_candidate_counts_just_past_a_multiple_of_512(n: int, device: torch.device) -> None:
    """Regression: k_select scans its candidate list 512 entries per trip; with 512 t + (1..63) candidates only some
    lanes ran the last trip, their ballot-counted survivor total went stale and the candidates of that trip were
    dropped (found by scripts/fuzz_search.py: row 512 of a 513-row bank never came back)."""
    from oracle import c_oracle

    bank, queries = cases.search_case(n, 33, 256, torch.float32, seed=n)
    eb = _bank(bank, device)
    scores, indices = eb.search(queries.to(device), 58)
    exp_s, exp_i = c_oracle.cosine_topk(bank.numpy(), queries.numpy(), 58)
    _check(scores, indices, exp_s, exp_i)
    assert int(eb.last_status[0].item()) == 0


def test_randomised_shapes_against_the_c_oracle(device: torch.device) -> None:
    """A fixed-seed slice of scripts/fuzz_search.py: shapes around the tile, level and padding boundaries, fp16 / fp32,
    duplicated rows, zero queries, index_base.  Indices exact, scores to 1e-6."""
    from imagescry_amd import EmbeddingBank
    from oracle import c_oracle

    rng = np.random.default_rng(7)
    ns = [1, 2, 15, 16, 17, 255, 256, 257, 511, 513, 4095, 4096, 4097, 5000]
    ds = [1, 3, 31, 32, 33, 63, 64, 65, 100, 128, 384]
    qs = [1, 2, 15, 63, 64, 65, 127, 128, 129, 255, 256, 257]
    ks = [1, 2, 9, 10, 16, 17, 58, 120]
    for _ in range(60):
        n, d, q = int(rng.choice(ns)), int(rng.choice(ds)), int(rng.choice(qs))
        k = int(rng.choice([kk for kk in ks if kk <= n]))
        dtype = torch.float16 if rng.random() < 0.5 else torch.float32
        g = torch.Generator().manual_seed(int(rng.integers(1 << 31)))
        bank, queries = torch.randn(n, d, generator=g), torch.randn(q, d, generator=g)
        if rng.random() < 0.3 and n > 4:
            bank[torch.randint(0, n, (n // 3,), generator=g)] = bank[torch.randint(0, n, (n // 3,), generator=g)]
        if rng.random() < 0.2:
            queries[int(rng.integers(q))] = 0
        base = int(rng.choice([0, 7, 1 << 33]))
        eb = EmbeddingBank(bank.to(device), dtype=dtype, normalize=bool(rng.random() < 0.5), index_base=base,
                           presharded=base != 0)
        scores, indices = eb.search(queries.to(device), k)
        exp_s, exp_i = c_oracle.cosine_topk(eb.bank.cpu().float().numpy(), queries.to(dtype).float().numpy(), k,
                                            index_base=base)
        _check(scores, indices, exp_s, exp_i)


@pytest.mark.parametrize("k,dtype", [(100, torch.float16), (120, torch.float32), (58, torch.float16)])
def test_large_k_on_a_multi_level_bank_stays_on_the_fast_path(k: int, dtype: torch.dtype, device: torch.device) -> None:
    """Regression (scripts/fuzz_search.py deep): with a fixed 64x level growth every search with k > 58 over more than
    4096 rows overflowed the per-query candidate list and silently took the exhaustive kernel.  Also a zero query and
    duplicated rows (exact ties with the threshold) must not push the call there."""
    from oracle import c_oracle

    g = torch.Generator().manual_seed(k)
    n, d, q = 300_000, 64, 64
    bank = torch.randn(n, d, generator=g)
    bank[torch.randint(0, n, (n // 3,), generator=g)] = bank[torch.randint(0, n, (n // 3,), generator=g)]
    queries = torch.randn(q, d, generator=g)
    queries[5] = 0
    from imagescry_amd import EmbeddingBank

    eb = EmbeddingBank(bank.to(device), dtype=dtype, normalize=True)
    scores, indices = eb.search(queries.to(device), k)
    assert int(eb.last_status[0].item()) == 0  # no candidate buffer overflowed
    exp_s, exp_i = c_oracle.cosine_topk(eb.bank.cpu().float().numpy(), queries.to(dtype).float().numpy(), k)
    _check(scores, indices, exp_s, exp_i)
    assert indices[5].cpu().tolist() == list(range(k))  # all-zero query: every score ties at 0, index order
