"""The search oracle: self-consistency, tie rule, sharding invariance, golden fixtures.

No reference code or test exists for the search step -- "parity unpinned" by the
reference; these tests pin the oracle's own definition (oracle/search_oracle.py docstring).
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

GOLDEN = np.load(Path(__file__).resolve().parent / "golden" / "search.npz")


def _brute(bank: torch.Tensor, queries: torch.Tensor, k: int) -> tuple[np.ndarray, np.ndarray]:
    s = search_oracle.exact_scores(bank, queries)
    idx = np.broadcast_to(np.arange(s.shape[1]), s.shape)
    order = np.lexsort((idx, -s.astype(np.float64)), axis=1)[:, :k]
    return np.take_along_axis(s, order, axis=1), order.astype(np.int64)


@pytest.mark.parametrize("name", ["f16_n4096_d768_q64", "f32_n5001_d96_q3", "f16_n300_d64_q300"])
def test_oracle_reproduces_golden(name: str) -> None:
    n, d, q, k, dtype = cases.SEARCH_CASES[name]
    bank, queries = cases.search_case(n, d, q, dtype)
    s, i = search_oracle.cosine_topk(bank, queries, k, block_rows=1000)
    np.testing.assert_array_equal(i, GOLDEN[f"{name}_indices"])
    np.testing.assert_array_equal(s, GOLDEN[f"{name}_scores"])
    bs, bi = _brute(bank, queries, k)
    np.testing.assert_array_equal(i, bi)
    np.testing.assert_array_equal(s, bs)


def test_tie_rule_lowest_index_first() -> None:
    """`torch.topk` does not guarantee this: scores [1,3,3,2,3,0,3], k=3 -> rows 1,2,4."""
    bank = torch.tensor([[1.0], [3.0], [3.0], [2.0], [3.0], [0.0], [3.0]])
    s, i = search_oracle.cosine_topk(bank, torch.tensor([[2.0]]), 3)
    assert i.tolist() == [[1, 2, 4]]
    assert s.tolist() == [[3.0, 3.0, 3.0]]
    bank, queries = cases.tie_case(torch.float16)
    s, i = search_oracle.cosine_topk(bank, queries, 50)
    np.testing.assert_array_equal(i, GOLDEN["tie_f16_indices"])
    assert i[0, :40].tolist() == list(range(0, 960, 24))


def test_zero_query_and_scores_are_cosines() -> None:
    bank, queries = cases.search_case(500, 32, 3, torch.float32)
    queries[1] = 0
    s, i = search_oracle.cosine_topk(bank, queries, 5)
    assert i[1].tolist() == [0, 1, 2, 3, 4] and float(np.abs(s[1]).max()) == 0.0
    ref = torch.nn.functional.cosine_similarity(queries[0][None].double(), bank.double())  # bank rows are unit
    assert np.allclose(s[0], np.sort(ref.numpy())[::-1][:5], atol=1e-6)


def test_sharded_merge_equals_unsharded() -> None:
    """Row-shard the bank 2 / 4 / 8 ways, search each shard with its index_base, merge: same answer
    (the merge order is total, so the result is independent of G)."""
    bank, queries = cases.search_case(3001, 48, 9, torch.float16)
    k = 10
    exp_s, exp_i = search_oracle.cosine_topk(bank, queries, k)
    for g in (2, 4, 8):
        parts = []
        for r in range(g):
            lo, hi = r * 3001 // g, (r + 1) * 3001 // g
            parts.append(search_oracle.cosine_topk(bank[lo:hi], queries, k, index_base=lo))
        s = np.stack([p[0] for p in parts])
        i = np.stack([p[1] for p in parts])
        ms, mi = search_oracle.topk_merge(s, i, k)
        np.testing.assert_array_equal(mi, exp_i)
        np.testing.assert_array_equal(ms, exp_s)


def test_reference_style_expression_agrees_up_to_near_ties() -> None:
    """The float32 `F.normalize(q) @ bank.T` expression a reference user would write returns the same rows
    except where two scores are closer than float32 accumulation noise; such swaps are flagged, not failed."""
    bank, queries = cases.search_case(4096, 768, 64, torch.float16)
    k = 10
    exp_s, exp_i = search_oracle.cosine_topk(bank, queries, k)
    ref_s, ref_i = search_oracle.cosine_topk_reference_style(bank, queries, k)
    np.testing.assert_allclose(ref_s.numpy(), exp_s, rtol=0, atol=1e-5)  # north_star: scores within 1e-5 (fp32)
    full = search_oracle.exact_scores(bank, queries)
    mism = np.argwhere(ref_i.numpy() != exp_i)
    for qi, r in mism:  # every mismatch must be a near-tie swap
        a, b = full[qi, ref_i[qi, r]], full[qi, exp_i[qi, r]]
        assert abs(float(a) - float(b)) < 2e-6
    assert len(mism) <= 0.02 * exp_i.size


def test_blocked_torch_baseline_matches() -> None:
    bank, queries = cases.search_case(5000, 64, 7, torch.float32)
    s, i = search_oracle.cosine_topk_torch_blocked(bank, queries, 10, block_rows=777)
    exp_s, exp_i = search_oracle.cosine_topk(bank, queries, 10)
    np.testing.assert_allclose(s.numpy(), exp_s, rtol=0, atol=1e-5)
    assert (i.numpy() == exp_i).mean() > 0.98


def test_argument_checks() -> None:
    bank, queries = cases.search_case(10, 8, 2, torch.float32)
    with pytest.raises(ValueError):
        search_oracle.cosine_topk(bank, queries, 11)
    with pytest.raises(ValueError):
        search_oracle.cosine_topk(bank, queries, 0)


def test_c_restatement_agrees_with_numpy_oracle() -> None:
    """oracle/c/search_oracle.c is written independently of numpy; both restatements must give the same answer
    (indices and float32 scores identical) on the golden cases, including exact ties."""
    from oracle import c_oracle

    for name in ("f16_n4096_d768_q64", "f32_n5001_d96_q3", "f16_n300_d64_q300"):
        n, d, q, k, dtype = cases.SEARCH_CASES[name]
        bank, queries = cases.search_case(n, d, q, dtype)
        s, i = c_oracle.cosine_topk(bank.float().numpy(), queries.float().numpy(), k, index_base=5)
        np.testing.assert_array_equal(i, GOLDEN[f"{name}_indices"] + 5)
        np.testing.assert_allclose(s, GOLDEN[f"{name}_scores"], rtol=0, atol=1e-7)
    bank, queries = cases.tie_case(torch.float32)
    s, i = c_oracle.cosine_topk(bank.numpy(), queries.numpy(), 50)
    np.testing.assert_array_equal(i, GOLDEN["tie_f32_indices"])
