"""CPU oracle for brute-force cosine top-k (test infrastructure, see oracle/__init__.py).

The reference contains no similarity search (SURVEY.md section 0 fact 2:
src/imagescry/storage/*.py is SQLite CRUD); the expression below is the one
BASELINE.json's `north_star` states.  PARITY UNPINNED by the reference -- this file
*is* the definition the HIP path is held to.

Definition (DESIGN.md "search semantics"):
    dot(q, b)  = sum_d q_d * b_d          evaluated in float64 (products of fp16 or
                                          fp32 inputs are exact in float64)
    score(q,b) = float32( dot(q, b) / max(||q||_2, 1e-12) )     (||q|| in float64)
    top-k      = the k rows with the largest `score`, ordered by
                 (score descending, row index ascending)
Bank rows are taken as stored (the bank is L2-normalised once when it is built, with
the `F.normalize` formula of reference src/imagescry/models/embedding.py:74).
`torch.topk` on CPU does not break ties by lowest index, hence the explicit lexsort.
"""

from __future__ import annotations

import numpy as np
import torch
from torch import Tensor


def _to_numpy(x: Tensor | np.ndarray) -> np.ndarray:
    if isinstance(x, Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def l2_normalize_rows(x: Tensor, eps: float = 1e-12) -> Tensor:
    """`F.normalize(x, p=2, dim=1)` for a `[N, D]` matrix (reference: embedding.py:74)."""
    return torch.nn.functional.normalize(x.float(), p=2, dim=1, eps=eps)


def exact_scores(bank: Tensor | np.ndarray, queries: Tensor | np.ndarray) -> np.ndarray:
    """`score(q, b)` for every pair -> float32 `[Q, N]` (small inputs only)."""
    b = _to_numpy(bank).astype(np.float64)
    q = _to_numpy(queries).astype(np.float64)
    denom = np.maximum(np.sqrt((q * q).sum(axis=1)), 1e-12)
    return ((q @ b.T) / denom[:, None]).astype(np.float32)


def _select(scores: np.ndarray, idx: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    order = np.lexsort((idx, -scores.astype(np.float64)))[:k]
    return scores[order], idx[order]


def cosine_topk(
    bank: Tensor | np.ndarray,
    queries: Tensor | np.ndarray,
    k: int,
    *,
    index_base: int = 0,
    block_rows: int = 32768,
) -> tuple[np.ndarray, np.ndarray]:
    """Exact cosine top-k.  Returns `(scores float32 [Q,k], indices int64 [Q,k])`."""
    b_all = _to_numpy(bank)
    q = _to_numpy(queries).astype(np.float64)
    n, nq = b_all.shape[0], q.shape[0]
    if not 0 < k <= n:
        raise ValueError(f"k={k} must be in [1, N={n}]")
    denom = np.maximum(np.sqrt((q * q).sum(axis=1)), 1e-12)
    best_s = [np.empty(0, np.float32) for _ in range(nq)]
    best_i = [np.empty(0, np.int64) for _ in range(nq)]
    for r0 in range(0, n, block_rows):
        blk = b_all[r0 : r0 + block_rows].astype(np.float64)
        s = ((q @ blk.T) / denom[:, None]).astype(np.float32)
        nb = s.shape[1]
        kk = min(k, nb)
        kth = np.partition(s, nb - kk, axis=1)[:, nb - kk]
        for qi in range(nq):
            sel = np.nonzero(s[qi] >= kth[qi])[0]  # every tie with the k-th value is kept
            cs = np.concatenate([best_s[qi], s[qi, sel]])
            ci = np.concatenate([best_i[qi], sel.astype(np.int64) + r0])
            best_s[qi], best_i[qi] = _select(cs, ci, k)
    return np.stack(best_s), np.stack(best_i) + index_base


def cosine_topk_reference_style(bank: Tensor, queries: Tensor, k: int) -> tuple[Tensor, Tensor]:
    """The float32 torch expression a reference user would write: `F.normalize(q) @ bank.T` -> stable sort.

    float32 accumulation order (MKL) makes near-ties land differently from the exact
    definition; tests compare against it with a near-tie allowance, never bit-exactly.
    """
    s = l2_normalize_rows(queries) @ bank.float().T
    order = torch.sort(s, dim=1, descending=True, stable=True).indices[:, :k]
    return torch.gather(s, 1, order), order


def cosine_topk_torch_blocked(bank_f32: Tensor, queries: Tensor, k: int, block_rows: int = 65536) -> tuple[Tensor, Tensor]:
    """CPU baseline used by bench.py: blocked float32 GEMM + `torch.topk` per block, then a final top-k.

    Same expression as `cosine_topk_reference_style` but usable at N = 1M (a full stable
    sort of 1024 x 1M scores is not).  Ties are not index-ordered here; it is timed, not compared.
    """
    qn = l2_normalize_rows(queries)
    parts_s, parts_i = [], []
    for r0 in range(0, bank_f32.shape[0], block_rows):
        s = qn @ bank_f32[r0 : r0 + block_rows].T
        ts, ti = torch.topk(s, min(k, s.shape[1]), dim=1)
        parts_s.append(ts)
        parts_i.append(ti + r0)
    s = torch.cat(parts_s, dim=1)
    i = torch.cat(parts_i, dim=1)
    ts, sel = torch.topk(s, k, dim=1)
    return ts, torch.gather(i, 1, sel)


def topk_merge(scores: np.ndarray, indices: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """Merge per-shard partial top-k `[G, Q, kin]` into `[Q, k]` by (score desc, index asc)."""
    scores = np.asarray(scores)
    indices = np.asarray(indices)
    g, nq, kin = scores.shape
    out_s = np.empty((nq, k), np.float32)
    out_i = np.empty((nq, k), np.int64)
    for qi in range(nq):
        out_s[qi], out_i[qi] = _select(scores[:, qi, :].reshape(-1), indices[:, qi, :].reshape(-1), k)
    return out_s, out_i
