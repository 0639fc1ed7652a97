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
from props import assert_topk_properties as _assert_topk_properties  # noqa: E402

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
        # several 256-query tiles (the half-major K step of k_dots_filter): an odd number of K steps and a ragged last
        # tile; an fp32 bank with a zero-padded last K step; four full tiles at the benchmark's D
        (40000, 192, 513, 10, torch.float16),
        (30000, 100, 700, 7, torch.float32),
        (16000, 768, 1024, 10, torch.float16),
        # the 128-query tile (64 < Q <= 128: waves 4 x 2, 3-deep rings): full and ragged query tiles, three levels, an fp32
        # bank with a padded K step, k past the 64-candidate lists; and the query counts around it -- 192 / 384 / 512
        (70000, 768, 128, 10, torch.float16),
        (266241, 64, 100, 10, torch.float16),
        (30000, 100, 65, 7, torch.float32),
        (50000, 256, 127, 70, torch.float16),
        (60000, 768, 192, 10, torch.float16),
        (30000, 384, 384, 10, torch.float16),
        (20000, 768, 512, 10, torch.float16),
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


def test_signed_zero_scores_are_one_score(device: torch.device) -> None:
    """A zero query scores -0.0 against a bank row without a positive entry and +0.0 against the others; both are the
    score 0 (to the oracle's comparison and to torch.topk alike), so the answer is rows 0..k-1 -- in the fast path, in the
    exact pass and in the merge of partial lists."""
    from imagescry_amd import EmbeddingBank, _lib

    bank, queries = cases.search_case(3000, 64, 3, torch.float16)
    bank[::3] = -bank[::3].abs()  # every third row: no positive entry
    queries[1] = 0
    exp_s, exp_i = search_oracle.cosine_topk(bank, queries, 12)
    eb = EmbeddingBank(bank.to(device), dtype=torch.float16, normalize=False)
    for search in (eb.search, eb.search_exhaustive):
        scores, indices = search(queries.to(device), 12)
        _check(scores, indices, exp_s, exp_i)
        assert indices[1].cpu().tolist() == list(range(12))
    # the merge kernel on its own: partial lists holding both zeros
    sc = torch.tensor([[[0.0, -0.0, -1.0]], [[-0.0, 0.0, -2.0]]])  # [G = 2, Q = 1, kin = 3]
    ix = torch.tensor([[[7, 3, 100]], [[5, 9, 101]]], dtype=torch.int64)
    out_s = torch.empty((1, 5), device=device)
    out_i = torch.empty((1, 5), dtype=torch.int64, device=device)
    sd, idd = sc.to(device), ix.to(device)
    _lib.check(_lib.load().isc_topk_merge(sd.data_ptr(), idd.data_ptr(), 2, 1, 3, 5, 0, 0, out_s.data_ptr(),
                                          out_i.data_ptr(), _lib.stream_handle(device)), "isc_topk_merge")
    assert out_i.cpu().tolist() == [[3, 5, 7, 9, 100]]
    assert out_s.cpu().tolist() == [[0.0, 0.0, 0.0, 0.0, -1.0]]


def test_bank_normalisation_and_fp32_queries(device: torch.device) -> None:
    """`normalize=True` applies the F.normalize formula before the cast; float32 queries are rounded to the bank dtype
    (inside `isc_cosine_topk`, ABI 4)."""
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


@pytest.mark.parametrize("bank_dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("n,d,nq", [(3000, 192, 6), (70_000, 768, 200), (20_000, 100, 1030)])
def test_query_dtype_is_independent_of_the_bank_dtype(bank_dtype: torch.dtype, n: int, d: int, nq: int,
                                                      device: torch.device) -> None:
    """ABI 4: `isc_cosine_topk` takes the queries in their own element type (`q_dtype`) and rounds them to the bank dtype
    while it packs them -- float32 -> fp16 by round to nearest even, the arithmetic of `Tensor.to(float16)`; fp16 -> float32
    exactly.  So float32 queries against an fp16 bank (the reference-shaped call: `get_flat_vectors()` is float32,
    data.py:112-118) give BIT-IDENTICAL (scores, indices) to cast-then-search, with no cast kernel in front; likewise fp16
    queries against a float32 bank; the exhaustive kernel rounds the same way; and the result equals the oracle's for the
    cast queries.  Queries include values that round across an fp16 binade, fp16 denormals, overflow to inf and a
    non-unit leading dimension."""
    from imagescry_amd import EmbeddingBank, _lib

    g = cases.gen(n + nq)
    bank = search_oracle.l2_normalize_rows(torch.randn(n, d, generator=g)).to(bank_dtype)
    eb = EmbeddingBank(bank.to(device), dtype=bank_dtype, normalize=False)
    k = 10
    for q_dtype in (torch.float32, torch.float16):
        wide = torch.randn(nq, d + 8, generator=g)
        wide[0, :4] = torch.tensor([2049.0, 1.0 + 2.0 ** -11, 3.0e-8, 6.0e-6])  # ties-to-even, fp16 denormals
        if nq > 3:
            wide[3, 0] = 1.0e5  # float32 -> fp16: inf
        wide = wide.to(q_dtype).to(device)
        queries = wide[:, :d]  # ldq = d + 8: the library reads the rows in place
        assert queries.stride(0) == d + 8
        cast = queries.to(bank_dtype).contiguous()  # what the host used to do in front of the call (an ATen cast kernel)
        got_s, got_i = eb.search(queries, k)
        ref_s, ref_i = eb.search(cast, k)
        assert torch.equal(got_i, ref_i)
        assert torch.equal(got_s.view(torch.int32), ref_s.view(torch.int32))  # bit for bit, NaN rows included
        if n <= 20_000:
            ex_s, ex_i = eb.search_exhaustive(queries, k)
            ex2_s, ex2_i = eb.search_exhaustive(cast, k)
            assert torch.equal(ex_i, ex2_i) and torch.equal(ex_s.view(torch.int32), ex2_s.view(torch.int32))
        finite = torch.isfinite(cast.float()).all(dim=1).cpu()  # (the numpy oracle itself only takes finite queries)
        good = finite.nonzero().flatten().tolist()
        exp_s, exp_i = search_oracle.cosine_topk(bank, cast.cpu()[good], k)
        _check(got_s[good], got_i[good], exp_s, exp_i)
        for bad in (~finite).nonzero().flatten().tolist():  # overflowed to inf in fp16: NaN against every row
            assert got_i[bad].cpu().tolist() == list(range(k)) and bool(torch.isnan(got_s[bad]).all())
        if nq > 3:
            assert bool(finite[3]) == (bank_dtype == torch.float32 and q_dtype == torch.float32)
    # the C entry point itself: a q_dtype that is no float type is refused before anything is launched
    need = _lib.c_size_t()
    code = _lib.dtype_code(bank_dtype)
    lib = _lib.load()
    _lib.check(lib.isc_cosine_topk_workspace_bytes(code, n, d, 2, k, need), "ws")
    ws = torch.empty(need.value, dtype=torch.uint8, device=device)
    s = torch.empty((2, k), dtype=torch.float32, device=device)
    i = torch.empty((2, k), dtype=torch.int64, device=device)
    st4 = torch.empty(4, dtype=torch.int32, device=device)
    bad = lib.isc_cosine_topk(eb._bank.data_ptr(), code, n, d, cast.data_ptr(), _lib.ISC_U8, 2, d, k, 0, None,
                              s.data_ptr(), i.data_ptr(), st4.data_ptr(), ws.data_ptr(), ws.numel(),
                              _lib.stream_handle(device))
    assert bad == _lib.ISC_ERR_INVALID_ARG


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
            eb._bank.data_ptr(), code, 5000, 160, q.data_ptr(), code, 11, 160, k, 1000, s.data_ptr(), i.data_ptr(),
            ws.data_ptr(), ws.numel(), _lib.stream_handle(device),
        )
        _lib.check(st, "isc_cosine_topk_exhaustive")
        _check(s, i, exp_s, exp_i)


def test_exhaustive_kernel_many_queries_and_wide_rows(device: torch.device) -> None:
    """k_exact with every query listed: more queries than one pass holds (1024), D = 5000 (one query per sweep),
    k = 120 (few, long partial lists) -- against the C oracle."""
    from oracle import c_oracle

    for n, d, q, k, dtype in ((3000, 40, 1100, 3, torch.float16), (700, 5000, 3, 120, torch.float32),
                              (70000, 200, 9, 10, torch.float16)):
        bank, queries = cases.search_case(n, d, q, dtype, seed=n)
        eb = _bank(bank, device)
        s, i = eb.search_exhaustive(queries.to(device), k)
        exp_s, exp_i = c_oracle.cosine_topk(bank.float().numpy(), queries.float().numpy(), k)
        _check(s, i, exp_s, exp_i)


def test_packed_bank_round_trip(device: torch.device) -> None:
    """isc_bank_pack -> isc_bank_unpack returns the rows bit for bit (ragged N, D not a multiple of the K step)."""
    for dtype in (torch.float16, torch.float32):
        bank, _ = cases.search_case(1234, 100, 1, dtype, seed=4)
        eb = _bank(bank, device)
        assert torch.equal(eb.bank.cpu(), bank)
        assert eb._bank.numel() == 5 * (2 if dtype == torch.float16 else 4) * 256 * 128


def test_bank_sorted_by_similarity_stays_on_the_fast_path(device: torch.device) -> None:
    """A bank sorted by ascending similarity to the query used to make every later row beat the threshold (levels were
    prefixes of the rows as they arrived): candidate buffers overflowed and the whole call fell back.  The packed bank
    now stores rows in a pseudo-random order, so the same bank is answered by the fast path."""
    d = 64
    g = cases.gen(11)
    q = torch.nn.functional.normalize(torch.randn(1, d, generator=g), dim=1)
    noise = torch.nn.functional.normalize(torch.randn(400000, d, generator=g), dim=1)
    t = torch.linspace(0.0, 0.9, 400000)[:, None]
    bank = torch.nn.functional.normalize(t * q + (1 - t) * noise * 0.2, dim=1).half()
    queries = torch.cat([q, torch.randn(70, d, generator=g)]).half()
    from oracle import c_oracle

    exp_s, exp_i = c_oracle.cosine_topk(bank.float().numpy(), queries.float().numpy(), 10)
    eb = _bank(bank, device)
    scores, indices = eb.search(queries.to(device), 10)
    _check(scores, indices, exp_s, exp_i)
    st = eb.last_status.cpu().tolist()
    # no overflow.  Query 0 itself may go through the exact pass, legitimately: the rows next to it are 2e-6 apart in
    # t, i.e. their scores lie inside float32 accumulation noise of each other, which is what the guard is for.
    assert st[0] == 0 and st[1] <= 1, st


def test_database_ordered_bank_of_near_duplicate_cells(device: torch.device) -> None:
    """Rows in the reference's store order (record, h, w) -- src/imagescry/storage/operations.py:135-144,
    src/imagescry/data.py:112-118: the 49 cells of one 7 x 7 map are adjacent and nearly identical.  Queries are cells of
    stored images, so ~49 rows crowd the top of every result.  Exact answer, fast path."""
    from imagescry_amd import EmbeddingBank
    from oracle import c_oracle

    g = cases.gen(31)
    images, cells, d = 6000, 49, 96
    centres = torch.nn.functional.normalize(torch.randn(images, d, generator=g), dim=1)
    rows = centres[:, None, :] + 0.05 * torch.randn(images, cells, d, generator=g)
    rows = rows.reshape(images * cells, d)
    eb = EmbeddingBank(rows.to(device), dtype=torch.float16, normalize=True)
    stored = eb.bank.cpu()
    queries = stored[torch.randint(0, images * cells, (96,), generator=g)].float()
    queries += 0.01 * torch.randn(queries.shape, generator=g)
    scores, indices = eb.search(queries.to(device), 10)
    exp_s, exp_i = c_oracle.cosine_topk(stored.float().numpy(), queries.half().float().numpy(), 10)
    _check(scores, indices, exp_s, exp_i)
    st = eb.last_status.cpu().tolist()
    assert st[0] == 0 and st[1] <= 2, st


def _ulp_cluster_case(dtype: torch.dtype):
    """30 rows that differ from each other in the last ulp of a few components, all next to the query: more rows
    than the filter carries (kp = 16 at k = 10) lie within float32 accumulation noise of the 10th score."""
    g = cases.gen(41)
    d, n = 768, 20000
    bank = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1)
    q = torch.nn.functional.normalize(torch.randn(1, d, generator=g), dim=1)
    base = torch.nn.functional.normalize(q + 0.3 * torch.nn.functional.normalize(torch.randn(1, d, generator=g), dim=1), dim=1)
    base = base.to(dtype)
    where = torch.randperm(n, generator=g)[:30]
    for j, r in enumerate(where.tolist()):
        row = base[0].clone()
        comp = torch.randint(0, d, (3,), generator=g)
        bits = row[comp].view(torch.int16 if dtype == torch.float16 else torch.int32)
        row[comp] = (bits + (1 if j % 2 else -1)).view(dtype)  # one ulp up or down
        bank[r] = row.float()
    return bank.to(dtype), torch.cat([q, torch.randn(3, d, generator=g)]).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_rounding_guard_catches_near_duplicates_of_the_kth_neighbour(dtype: torch.dtype, device: torch.device) -> None:
    """The float32 filter ranks the candidates it carries; rows that differ from the k-th neighbour by one ulp can swap
    places with it inside float32 accumulation noise.  k_final's guard must notice that it cannot prove the filter's
    choice and hand the query to the exact pass; the answer is bit-exact either way."""
    from oracle import c_oracle

    bank, queries = _ulp_cluster_case(dtype)
    eb = _bank(bank, device)
    scores, indices = eb.search(queries.to(device), 10)
    exp_s, exp_i = c_oracle.cosine_topk(bank.float().numpy(), queries.float().numpy(), 10)
    _check(scores, indices, exp_s, exp_i)
    st = eb.last_status.cpu()
    assert int(st[0]) == 0 and int(st[1]) >= 1  # query 0 went through the exact pass
    ratio = st[2:3].view(torch.float32).item()
    assert 0.0 < ratio < 0.5, ratio  # observed filter error, in units of the guard's bound


def test_every_query_through_the_exact_pass(device: torch.device) -> None:
    """A bank of 2 000 distinct vectors each stored 100 times: the ten best rows of every query are exact copies of one
    vector, more copies than the filter carries -- its choice among them cannot be proven, so EVERY query is searched
    again: ONE more matrix-core pass over the bank for all 64 together with the threshold fixed just below the k-th
    exact score, all ~100 survivors per query re-scored in float64 (k_final2); none needs the exhaustive sweep.  The
    answer must be the ten LOWEST original indices of the best vector, exactly as the oracle orders ties."""
    from oracle import c_oracle

    g = cases.gen(51)
    base = torch.nn.functional.normalize(torch.randn(2000, 128, generator=g), dim=1)
    bank = base.repeat(100, 1)[torch.randperm(200_000, generator=g)].half()
    queries = torch.randn(64, 128, generator=g).half()
    eb = _bank(bank, device)
    scores, indices = eb.search(queries.to(device), 10)
    exp_s, exp_i = c_oracle.cosine_topk(bank.float().numpy(), queries.float().numpy(), 10)
    _check(scores, indices, exp_s, exp_i)
    st = eb.last_status.cpu().tolist()
    assert st[1] == 64 and st[3] == 0, st
    # the same through the explicit exhaustive entry point, and a second search on the same workspace stays correct
    es, ei = eb.search_exhaustive(queries.to(device), 10)
    assert torch.equal(ei, indices) and torch.equal(es, scores)
    s2, i2 = eb.search(queries[:7].to(device), 10)
    assert torch.equal(i2, indices[:7]) and torch.equal(s2, scores[:7])


def test_guard_bound_holds_for_same_sign_vectors(device: torch.device) -> None:
    """Worst case for the rounding bound: every product q_i * b_i has the same sign, so the running sum is as large as
    sum |q_i b_i| all the way.  status[2] reports max |filter score - exact dot| / bound over the re-scored candidates."""
    g = cases.gen(43)
    for dtype, d in ((torch.float16, 768), (torch.float16, 4096), (torch.float32, 768)):
        bank = torch.nn.functional.normalize(torch.rand(30000, d, generator=g) + 0.05, dim=1).to(dtype)
        queries = (torch.rand(40, d, generator=g) + 0.05).to(dtype)
        eb = _bank(bank, device)
        scores, indices = eb.search(queries.to(device), 10)
        es, ei = eb.search_exhaustive(queries.to(device), 10)
        assert torch.equal(indices, ei) and torch.equal(scores, es)
        ratio = eb.last_status.cpu()[2:3].view(torch.float32).item()
        assert 0.0 <= ratio < 0.5, (dtype, d, ratio)


def test_nan_and_inf_queries(device: torch.device) -> None:
    """A query holding inf (or one that overflows fp16 to inf) scores NaN against every row: the oracle's order puts NaN
    last, i.e. rows 0..k-1 with NaN scores.  Every output slot is written (the first version left slots 1..k-1 of such
    a query uninitialised)."""
    bank, queries = cases.search_case(3000, 64, 5, torch.float16, seed=2)
    queries = queries.float()
    queries[1, 3] = float("inf")
    queries[3, 0] = 1e6  # overflows to inf in fp16
    queries[4] = float("nan")
    q16 = queries.half()
    scores, indices = _bank(bank, device).search(queries.to(device), 6)
    good = [0, 2]
    exp_s, exp_i = search_oracle.cosine_topk(bank, q16[good], 6)  # (the numpy oracle itself only takes finite queries)
    np.testing.assert_array_equal(indices[good].cpu().numpy(), exp_i)
    np.testing.assert_allclose(scores[good].cpu().numpy(), exp_s, rtol=0, atol=SCORE_ATOL)
    for bad in (1, 3, 4):
        assert indices[bad].cpu().tolist() == list(range(6)) and bool(torch.isnan(scores[bad]).all())


def test_sixteen_thousand_queries_in_one_call(device: torch.device) -> None:
    """EmbedSearchPipeline with a spatial embedder hands over Q = B * h * w queries (12 800 for 32 images at 640 px).
    Calls with more than 1024 queries run as passes of 1024, so the per-segment candidate load does not grow with Q
    (with one pass, Q = 16 384 on a 16.7 M-row bank overflowed the 32-slot segments by construction)."""
    n, d, q, k = 16_700_000, 32, 16384, 10
    g = torch.Generator(device=device).manual_seed(5)
    from imagescry_amd import EmbeddingBank

    eb = EmbeddingBank(torch.randn(n, d, generator=g, device=device), dtype=torch.float16, normalize=True)
    queries = torch.randn(q, d, generator=g, device=device).half()
    scores, indices = eb.search(queries, k)
    st = eb.last_status.cpu().tolist()
    assert st[0] == 0, st
    assert st[1] <= q // 100, st  # d = 32: scores are coarse, a few near-ties may need the exact pass
    pick = torch.tensor([0, 1023, 1024, 5000, 16383], device=device)
    es, ei = eb.search_exhaustive(queries[pick], k)
    assert torch.equal(indices[pick], ei) and torch.equal(scores[pick], es)
    s = scores.double()
    assert bool(((s[:, :-1] > s[:, 1:]) | ((s[:, :-1] == s[:, 1:]) & (indices[:, :-1] < indices[:, 1:]))).all())


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
    from imagescry_amd import EmbeddingBank, _lib

    wide = EmbeddingBank(torch.randn(40, _lib.ISC_SEARCH_MAX_D + 8, device=device), dtype=torch.float16)
    with pytest.raises(ValueError):  # rejected up front by the fast path, not by its fallback in mid-call
        wide.search(torch.randn(2, _lib.ISC_SEARCH_MAX_D + 8, device=device), 5)


def test_full_size_properties_1m(device: torch.device) -> None:
    """BASELINE config 3 shape (1M x 768, 1024 queries, k = 10) checked without a CPU oracle."""
    n, d, q, k = 1_000_000, 768, 1024, 10
    g = torch.Generator(device=device).manual_seed(1234)
    bank = torch.nn.functional.normalize(torch.randn(n, d, generator=g, device=device), dim=1).half()
    queries = torch.randn(q, d, generator=g, device=device).half()
    from imagescry_amd import EmbeddingBank

    eb = EmbeddingBank(bank, dtype=torch.float16, normalize=False)
    scores, indices = eb.search(queries, k)
    st = eb.last_status.cpu().tolist()
    assert st[0] == 0 and st[1] == 0, st
    _assert_topk_properties(bank, queries, scores, indices, k)


def test_full_size_properties_10m_and_eight_shards(device: torch.device) -> None:
    """BASELINE config 4: the 10 M x 768 fp16 bank of the headline benchmark, at Q = 1024 and Q = 1, proven with the
    same three properties; then the same bank as eight row shards of 1.25 M rows (the shard of one GPU of the 8-GPU
    run, index_base != 0), searched one by one and merged with isc_topk_merge: identical to the unsharded answer."""
    n, d, q, k = 10_000_000, 768, 1024, 10
    from imagescry_amd import EmbeddingBank, shard_bounds

    g = torch.Generator(device=device).manual_seed(1234)
    bank = torch.empty((n, d), dtype=torch.float16, device=device)
    for r0 in range(0, n, 1 << 20):
        blk = torch.randn((min(1 << 20, n - r0), d), generator=g, device=device)
        bank[r0 : r0 + blk.shape[0]] = torch.nn.functional.normalize(blk, dim=1).half()
    queries = torch.randn(q, d, generator=g, device=device).half()
    eb = EmbeddingBank(bank, dtype=torch.float16, normalize=False)
    scores, indices = eb.search(queries, k)
    st = eb.last_status.cpu().tolist()
    assert st[0] == 0 and st[1] == 0, st
    _assert_topk_properties(bank, queries, scores, indices, k, block=1 << 18)
    s1, i1 = eb.search(queries[:1], k)  # the 64-query tile shape (HBM-bound launch)
    assert torch.equal(i1, indices[:1]) and torch.equal(s1, scores[:1])
    assert eb.last_status.cpu().tolist()[:2] == [0, 0]
    del eb
    torch.cuda.empty_cache()
    parts_s, parts_i = [], []
    for r in range(8):
        lo, hi = shard_bounds(n, 8, r)
        shard = EmbeddingBank(bank[lo:hi], dtype=torch.float16, normalize=False, index_base=lo, presharded=True)
        s, i = shard.search(queries, k)
        assert shard.last_status.cpu().tolist()[:2] == [0, 0]
        parts_s.append(s)
        parts_i.append(i)
        if r == 3:  # one shard also at the HBM-bound shape
            s16, i16 = shard.search(queries[:16], k)
            assert torch.equal(i16, i[:16]) and torch.equal(s16, s[:16])
    ms, mi = shard._merge_topk(torch.stack(parts_s), torch.stack(parts_i), k)
    assert torch.equal(mi, indices) and torch.equal(ms, scores)


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
    assert eb.last_status.cpu().tolist()[:2] == [0, 0]


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
def test_large_k_on_a_multi_level_bank(k: int, dtype: torch.dtype, device: torch.device) -> None:
    """Regression (scripts/fuzz_search.py deep): with a fixed 64x level growth every search with k > 58 over more than
    4096 rows overflowed the per-query candidate list and silently took the exhaustive kernel.  Also a zero query and
    duplicated rows (exact ties with the threshold) must not push the whole call there."""
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
    st = eb.last_status.cpu().tolist()
    # the zero query is answered directly (every score ties: rows 0..k-1); duplicated rows may send a few queries to the
    # second pass -- never the whole call, and none to the exhaustive sweep
    assert st[1] <= q // 4 and st[3] == 0, st
    exp_s, exp_i = c_oracle.cosine_topk(eb.bank.cpu().float().numpy(), queries.to(dtype).float().numpy(), k)
    _check(scores, indices, exp_s, exp_i)
    assert indices[5].cpu().tolist() == list(range(k))  # all-zero query: every score ties at 0, index order
    keep = [i for i in range(q) if i != 5]
    s2, i2 = eb.search(queries[keep].to(device), k)
    assert eb.last_status.cpu().tolist()[0] == 0  # without the degenerate query no candidate buffer overflows
    assert torch.equal(i2, indices[keep]) and torch.equal(s2, scores[keep])


def test_denormal_and_huge_scale_float32_queries(device: torch.device) -> None:
    """The rounding guard's bound is relative: it holds while products and partial sums stay in float32's normal range.
    A float32 query scaled to 1e-40 (products underflow, the filter scores lose their bits) or to 1e37 (a partial sum
    can overflow while the float64 score is finite) must therefore be answered by the exact pass -- `k_final` lists them
    -- and rank exactly as the oracle ranks them (the reference normalises the query first and is unaffected by scale)."""
    bank, queries = cases.search_case(5000, 64, 4, torch.float32, seed=77)
    queries = queries.clone()
    queries[0] *= 1e-40
    queries[1] *= 1e37
    eb = _bank(bank, device)
    scores, indices = eb.search(queries.to(device), 10)
    exp_s, exp_i = search_oracle.cosine_topk(bank, queries, 10)
    np.testing.assert_array_equal(indices.cpu().numpy(), exp_i)
    np.testing.assert_allclose(scores.cpu().numpy(), exp_s, rtol=1e-6, atol=0)
    st = eb.last_status.cpu().tolist()
    assert st[1] >= 2, st  # both extreme queries went through the exact pass
    # the same directions at unit scale rank identically (cosine is scale free) and stay on the fast path
    unit = torch.nn.functional.normalize(queries.double(), dim=1).float()
    s1, i1 = eb.search(unit.to(device), 10)
    assert int(eb.last_status.cpu()[1]) == 0
    assert torch.equal(i1[2:], indices[2:])


@pytest.mark.parametrize("nq", [5, 200, 1500])
def test_redo_pass_at_every_tile_shape(nq: int, device: torch.device) -> None:
    """The matrix-core redo with the 64-query tile (5 queries), one 256-query tile (200) and a call of two passes with
    four / two 256-query tiles (1500): every third query hits a vector stored 40 times, the others are ordinary; listed
    and unlisted queries share the launch."""
    from oracle import c_oracle

    g = cases.gen(61 + nq)
    base = torch.nn.functional.normalize(torch.randn(3000, 96, generator=g), dim=1)
    dup = base[:50].repeat(40, 1)
    bank = torch.cat([base, dup])[torch.randperm(5000, generator=g)].half()
    queries = torch.randn(nq, 96, generator=g)
    hit = torch.arange(0, nq, 3)
    queries[hit] = base[hit % 50] + 0.01 * torch.randn(len(hit), 96, generator=g)
    queries = queries.half()
    eb = _bank(bank, device)
    scores, indices = eb.search(queries.to(device), 10)
    exp_s, exp_i = c_oracle.cosine_topk(bank.float().numpy(), queries.float().numpy(), 10)
    _check(scores, indices, exp_s, exp_i)
    st = eb.last_status.cpu().tolist()  # the status words of the LAST pass of the call
    assert st[1] >= 1 and st[3] == 0, st


def test_more_ties_than_the_redo_lists_hold(device: torch.device) -> None:
    """20 000 identical rows (a uniform image patch indexed over and over) among 30 000: a query that hits them ties more
    rows at its k-th score than the redo lists hold (8 192), so it -- and only it -- falls through to the exhaustive
    float64 sweep; the answer is the ten lowest indices of the copies."""
    from oracle import c_oracle

    g = cases.gen(71)
    rows = torch.nn.functional.normalize(torch.randn(30_000, 64, generator=g), dim=1)
    patch = torch.nn.functional.normalize(torch.randn(64, generator=g), dim=0)
    where = torch.randperm(30_000, generator=g)[:20_000]
    rows[where] = patch
    bank = rows.half()
    queries = torch.randn(6, 64, generator=g)
    queries[2] = patch * 3.0
    queries = queries.half()
    eb = _bank(bank, device)
    scores, indices = eb.search(queries.to(device), 10)
    exp_s, exp_i = c_oracle.cosine_topk(bank.float().numpy(), queries.float().numpy(), 10)
    _check(scores, indices, exp_s, exp_i)
    assert indices[2].cpu().tolist() == sorted(where.tolist())[:10]
    st = eb.last_status.cpu().tolist()
    assert st[3] >= 1 and st[1] >= st[3], st


def test_zero_and_nonfinite_queries_need_no_second_search(device: torch.device) -> None:
    """Every row ties for a zero query (score 0) and for a query with a non-finite norm (score NaN): the answer is the
    first k rows by definition, written directly -- such queries (padding rows of a batch are zeros) must not be listed
    for any second pass."""
    bank, queries = cases.search_case(4000, 64, 6, torch.float16, seed=81)
    queries = queries.float()
    queries[1] = 0.0
    queries[4, 7] = float("inf")
    eb = _bank(bank, device)
    scores, indices = eb.search(queries.to(device), 8)
    st = eb.last_status.cpu().tolist()
    assert st[1] == 0 and st[3] == 0, st
    assert indices[1].cpu().tolist() == list(range(8)) and bool((scores[1] == 0).all())
    assert indices[4].cpu().tolist() == list(range(8)) and bool(torch.isnan(scores[4]).all())
    good = [0, 2, 3, 5]
    exp_s, exp_i = search_oracle.cosine_topk(bank, queries.half()[good], 8)
    np.testing.assert_array_equal(indices[good].cpu().numpy(), exp_i)
    # with an index base (a shard) the trivial answer is the shard's first rows
    eb2 = _bank(bank, device, index_base=1000, presharded=True)
    s2, i2 = eb2.search(queries.to(device), 8)
    assert i2[1].cpu().tolist() == list(range(1000, 1008))


def test_level_lists_do_not_overflow_on_iid_banks(device: torch.device) -> None:
    """Regression: a level that is R times the rows seen before lets ~ (R - 1) x Gamma(kp) rows per query through, and with
    the list sized for twice the MEAN about one query in 500 overflowed it at kp = 16 -- with 512 queries per call nearly
    every search of a bank large enough to use the full ratio (> 8.4 M rows at 257 < Q <= 512) paid a second pass.  The
    ratio now leaves eight standard deviations: no candidate buffer may overflow on an iid bank, whatever the queries."""
    n, d, q = 6_500_000, 64, 512
    g = torch.Generator(device=device).manual_seed(5)
    bank = torch.nn.functional.normalize(torch.randn(n, d, generator=g, device=device), dim=1).half()
    from imagescry_amd import EmbeddingBank

    eb = EmbeddingBank(bank, dtype=torch.float16, normalize=False)
    del bank
    overflowed = exhaustive = 0
    for seed in range(6):
        queries = torch.randn(q, d, generator=torch.Generator().manual_seed(300 + seed)).half().to(device)
        scores, indices = eb.search(queries, 10)
        st = eb.last_status.cpu().tolist()
        overflowed += st[0]
        exhaustive += st[3]
    assert overflowed == 0 and exhaustive == 0, (overflowed, exhaustive)
    # the last answer against the exhaustive float64 kernel (a handful of queries: it sweeps the bank per four)
    es, ei = eb.search_exhaustive(queries[:8], 10)
    assert torch.equal(indices[:8], ei) and torch.equal(scores[:8], es)


def test_async_searches_in_flight_equal_the_serial_answers(device: torch.device) -> None:
    """`search_async` alternates between two streams of the bank (each with its own workspace) so that the tail of one
    search runs beside the head of the next: handles resolved late, in any order, must hold exactly what `search` returns
    -- with different query counts in flight (different workspace buckets and tile shapes) and the query tensors dropped
    before their searches have run."""
    bank, _ = cases.search_case(300_000, 256, 1, torch.float16, seed=77)
    eb = _bank(bank, device)
    gen = torch.Generator().manual_seed(78)
    sizes = [5, 64, 300, 1, 700, 64, 17, 256]
    expect = []
    for nq in sizes:
        qs = torch.randn((nq, 256), generator=gen).half()
        s, i = eb.search(qs.to(device), 10)
        expect.append((qs, s.cpu(), i.cpu()))
    torch.cuda.synchronize()
    for order in (0, 1):
        handles = []
        for qs, _, _ in expect:
            qd = qs.to(device)
            handles.append(eb.search_async(qd, 10))
            del qd  # the lane must keep the queries alive until it has read them
        seq = range(len(handles)) if order == 0 else reversed(range(len(handles)))
        for j in seq:
            s, i = handles[j].result()
            assert torch.equal(i.cpu(), expect[j][2]) and torch.equal(s.cpu(), expect[j][1])
