"""Size-independent proofs used by the GPU tests at sizes no CPU oracle finishes (test infrastructure)."""

from __future__ import annotations

import torch


def assert_topk_properties(bank_rows: torch.Tensor, queries: torch.Tensor, scores: torch.Tensor, indices: torch.Tensor,
                            k: int, block: int = 65536) -> None:
    """Size-independent proof that (scores, indices) is THE cosine top-k of `queries` over `bank_rows` (row-major, on the
    GPU): sorted under the total order; the scores are the exact cosines of the returned rows; no row outside the set
    beats or ties-with-lower-index the k-th entry.  torch float64 matmuls on the device, test-side only."""
    n, d = bank_rows.shape
    q = queries.shape[0]
    s, i = scores.double(), indices
    assert bool(((s[:, :-1] > s[:, 1:]) | ((s[:, :-1] == s[:, 1:]) & (i[:, :-1] < i[:, 1:]))).all())
    q64 = queries.double()
    denom = q64.norm(dim=1).clamp_min(1e-12)
    rows = bank_rows[indices.reshape(-1)].double().reshape(q, k, d)
    exact = (torch.einsum("qkd,qd->qk", rows, q64) / denom[:, None]).float()
    assert torch.allclose(scores, exact, rtol=0, atol=1e-7)
    kth = scores[:, -1].double()
    kth_idx = indices[:, -1]
    better = torch.zeros(q, dtype=torch.int64, device=scores.device)
    for r0 in range(0, n, block):
        blk = (q64 @ bank_rows[r0 : r0 + block].double().T / denom[:, None]).float().double()
        ridx = torch.arange(r0, r0 + blk.shape[1], device=scores.device)[None, :]
        better += ((blk > kth[:, None]) | ((blk == kth[:, None]) & (ridx < kth_idx[:, None]))).sum(dim=1)
    assert bool((better == k - 1).all())
