"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/imagescry_hip.h
declares; the ctypes table in imagescry_amd/_lib.py covers exactly that set.  No compute is launched here."""

from __future__ import annotations

import ctypes
import re
from pathlib import Path

import pytest

from imagescry_amd import _lib, build

HEADER = Path(__file__).resolve().parents[1] / "include" / "imagescry_hip.h"


def declared_symbols() -> list[str]:
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(isc_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def library() -> ctypes.CDLL:
    build.build(verbose=False)
    return ctypes.CDLL(str(_lib.LIB_PATH))


def test_header_declares_the_expected_entry_points() -> None:
    names = declared_symbols()
    assert "isc_cosine_topk" in names and "isc_conv2d_nhwc" in names and "isc_channel_stats" in names
    assert len(names) >= 19


def test_library_exports_every_declared_symbol(library: ctypes.CDLL) -> None:
    missing = [name for name in declared_symbols() if not hasattr(library, name)]
    assert missing == []


def test_ctypes_table_matches_header() -> None:
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_search_entry_points_take_a_query_dtype() -> None:
    """ABI 4: `isc_cosine_topk(bank, dtype, N, D, queries, q_dtype, Q, ldq, ...)` -- the queries
    carry their own element type; header and ctypes table agree on where it sits."""
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    for name in ("isc_cosine_topk", "isc_cosine_topk_exhaustive"):
        proto = re.search(rf"\bint {name}\s*\(([^;]*?)\);", text, flags=re.S).group(1)
        params = [" ".join(a.split()) for a in proto.split(",")]
        assert params[4] == "const void* queries" and params[5] == "int q_dtype" and params[6] == "int Q", params
        assert len(_lib.SIGNATURES[name][1]) == len(params)
        assert _lib.SIGNATURES[name][1][5] is ctypes.c_int and _lib.SIGNATURES[name][1][7] is ctypes.c_int64  # ldq


def test_host_only_entry_points(library: ctypes.CDLL) -> None:
    """Functions that touch no device memory can run here: version, error strings, workspace sizing,
    argument validation."""
    lib = _lib.load()
    assert lib.isc_abi_version() == _lib.ISC_ABI_VERSION == 4
    assert lib.isc_build_flags() == 0  # the production build: no ablation variants
    assert _lib.strerror(0) == "ok"
    assert "workspace" in _lib.strerror(_lib.ISC_ERR_WORKSPACE)
    need = ctypes.c_size_t()
    assert lib.isc_cosine_topk_workspace_bytes(_lib.ISC_F16, 10_000_000, 768, 1024, 10, need) == 0
    # ~134 MiB of lane-private survivor segments + 64 MiB of per-query lists + 20 MiB of exact-pass partial lists
    assert 150e6 < need.value < 300e6
    assert lib.isc_cosine_topk_workspace_bytes(_lib.ISC_F16, 256, 768, 1, 10, need) == 0
    small = need.value
    assert small < 8e6
    # k <= N; k <= ISC_TOPK_MAX_K; any D (the packed layout zero-pads the last K step)
    assert lib.isc_cosine_topk_workspace_bytes(_lib.ISC_F16, 1000, 100, 4, 10, need) == 0
    assert lib.isc_bank_packed_bytes(_lib.ISC_F16, 10_000_000, 768, need) == 0
    assert need.value == 39063 * 12 * 256 * 128  # 39063 tiles x 12 K steps x 32 KiB
    assert lib.isc_cosine_topk_workspace_bytes(_lib.ISC_F32, 5, 32, 4, 10, need) == _lib.ISC_ERR_INVALID_ARG
    assert lib.isc_cosine_topk_workspace_bytes(_lib.ISC_F32, 500, 32, 4, 121, need) == _lib.ISC_ERR_UNSUPPORTED
    assert lib.isc_cosine_topk_workspace_bytes(_lib.ISC_U8, 500, 32, 4, 10, need) == _lib.ISC_ERR_INVALID_ARG
    # the fast path and the exact pass share their limits: a shape one accepts the other accepts too
    for fn in (lib.isc_cosine_topk_workspace_bytes, lib.isc_cosine_topk_exhaustive_workspace_bytes):
        assert fn(_lib.ISC_F16, 500, _lib.ISC_SEARCH_MAX_D, 70000, 10, need) == 0
        assert fn(_lib.ISC_F16, 500, _lib.ISC_SEARCH_MAX_D + 1, 4, 10, need) == _lib.ISC_ERR_UNSUPPORTED
    # a call with more than 1024 queries runs as passes over the workspace of 1024
    big, one = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.isc_cosine_topk_workspace_bytes(_lib.ISC_F16, 1_000_000, 64, 16384, 10, big) == 0
    assert lib.isc_cosine_topk_workspace_bytes(_lib.ISC_F16, 1_000_000, 64, 1024, 10, one) == 0
    assert big.value == one.value
    mul, inv = ctypes.c_int64(), ctypes.c_int64()
    for n in (1, 2, 3, 255, 256, 257, 1_250_000, 10_000_000, 2**31 - 2):
        assert lib.isc_bank_permutation(n, mul, inv) == 0
        assert (mul.value * inv.value) % n == (1 % n) and 0 < mul.value <= max(n - 1, 1)
    assert lib.isc_bank_permutation(2**31, mul, inv) == _lib.ISC_ERR_INVALID_ARG
    assert lib.isc_channel_stats_workspace_bytes(_lib.ISC_U8, 512, 3, 224, 224, need) == 0
    assert need.value == 3 * 512 * 4 * 16  # 4 chunks of 16384 pixels per plane, 16 bytes per partial
    # NULL pointers are rejected before anything is launched
    assert lib.isc_topk_merge(None, None, 1, 1, 1, 1, 0, 0, None, None, None) == _lib.ISC_ERR_INVALID_ARG
    assert lib.isc_l2norm_channels(None, 1, 1, 1, 1e-12, None, None) == _lib.ISC_ERR_INVALID_ARG
    with pytest.raises(ValueError):
        _lib.check(_lib.ISC_ERR_UNSUPPORTED, "x")
    with pytest.raises(_lib.HipLibraryError):
        _lib.check(_lib.ISC_ERR_LAUNCH, "x")


def test_production_library_holds_no_ablation_kernels(library: ctypes.CDLL) -> None:
    """The shipped library contains exactly one `k_dots_filter` per (dtype, tile shape, staging variant, level kind)
    that the search launches -- f16 / f32 x {64-query tile: filter and sample, each with the default and the non-temporal bank stream; 128-query tile:
    filter and sample (non-temporal); 256-query tile: split, unsplit, sample} --
    and nothing else: the wrong-result ablation variants only exist in -DISC_ABLATION builds, and no environment
    variable changes which kernel runs."""
    import shutil
    import subprocess

    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("nm not available")
    out = subprocess.run([nm, str(_lib.LIB_PATH)], check=True, capture_output=True, text=True).stdout
    # mangled template arguments: I <dtype: DF16_ | f> Li<tile>E Li<staging variant>E Lb<sample>E
    variants = sorted(set(re.findall(r"13k_dots_filterI(DF16_|f)Li(\d+)ELi(\d+)ELb([01])E", out)))
    # ... plus the REDO instantiations (20 / 32 / 33 = the redo forms of 0 / 12 / 13: the same loops against the fixed
    # thresholds of the queries k_final could not prove, kernels of their own so that profiles list them apart)
    expect = sorted((t, str(tnq), str(dbg), sample) for t in ("DF16_", "f")
                    for tnq, dbg, sample in ((64, 12, "0"), (64, 13, "0"), (64, 12, "1"), (64, 13, "1"), (256, 0, "0"),
                                             (256, 12, "0"), (256, 12, "1"), (64, 32, "0"), (64, 33, "0"), (256, 20, "0"),
                                             (256, 32, "0"),
                                             # the 128-query tile (64 < Q <= 128, always one query tile): non-temporal
                                             # stream in the sample and filter levels, its redo form
                                             (128, 13, "0"), (128, 13, "1"), (128, 33, "0")))
    assert variants == expect
    gemm = sorted(set(re.findall(r"k_gemm_f16_(dma|big)ILi(\d+)E", out)))
    assert gemm == [("big", "0"), ("dma", "0")]
    blob = _lib.LIB_PATH.read_bytes()
    assert b"ISC_DEBUG_MODE" not in blob and b"ISC_FORCE_TILE" not in blob and b"ISC_GEMM_DEBUG" not in blob
