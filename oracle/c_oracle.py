"""ctypes wrapper of oracle/c/search_oracle.c (TEST INFRASTRUCTURE, see oracle/__init__.py)."""

from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

C_DIR = Path(__file__).resolve().parent / "c"
LIB = C_DIR / "liboracle_search.so"


def build() -> Path:
    subprocess.run(["make", "-C", str(C_DIR), "-s"], check=True)
    return LIB


def cosine_topk(bank: np.ndarray, queries: np.ndarray, k: int, index_base: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """Exact cosine top-k in C.  `bank` / `queries` of any float dtype are upcast to float32 (exact for fp16)."""
    if not LIB.exists():
        build()
    lib = ctypes.CDLL(str(LIB))
    b = np.ascontiguousarray(bank, dtype=np.float32)
    q = np.ascontiguousarray(queries, dtype=np.float32)
    n, d = b.shape
    nq = q.shape[0]
    s = np.empty((nq, k), np.float32)
    i = np.empty((nq, k), np.int64)
    fn = lib.isc_oracle_cosine_topk
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                   ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    rc = fn(b.ctypes.data, n, d, q.ctypes.data, nq, k, index_base, s.ctypes.data, i.ctypes.data)
    if rc != 0:
        raise ValueError("invalid arguments")
    return s, i
