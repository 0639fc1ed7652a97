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


def test_host_only_entry_points(library: ctypes.CDLL) -> None:
    """Functions that touch no device memory can run here: version, error strings, workspace sizing,
    argument validation."""
    lib = _lib.load()
    assert lib.isc_abi_version() == 1
    assert _lib.strerror(0) == "ok"
    assert "workspace" in _lib.strerror(_lib.ISC_ERR_WORKSPACE)
    need = ctypes.c_size_t()
    assert lib.isc_cosine_topk_workspace_bytes(_lib.ISC_F16, 10_000_000, 768, 1024, 10, need) == 0
    assert 100e6 < need.value < 250e6  # ~134 MiB of lane-private survivor segments + 32 MiB of per-query lists
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
    assert lib.isc_channel_stats_workspace_bytes(_lib.ISC_U8, 512, 3, 224, 224, need) == 0
    assert need.value == 3 * 512 * 4 * 16  # 4 chunks of 16384 pixels per plane, 16 bytes per partial
    # NULL pointers are rejected before anything is launched
    assert lib.isc_topk_merge(None, None, 1, 1, 1, 1, 0, 0, None, None, None) == _lib.ISC_ERR_INVALID_ARG
    assert lib.isc_l2norm_channels(None, 1, 1, 1, 1e-12, None, None) == _lib.ISC_ERR_INVALID_ARG
    with pytest.raises(ValueError):
        _lib.check(_lib.ISC_ERR_UNSUPPORTED, "x")
    with pytest.raises(_lib.HipLibraryError):
        _lib.check(_lib.ISC_ERR_LAUNCH, "x")
