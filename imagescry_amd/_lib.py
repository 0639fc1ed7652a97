"""ctypes binding of `libimagescry_hip.so` (the C ABI declared in include/imagescry_hip.h).

There is no CPU fallback: if the shared library has not been built, or no HIP device is
present when a kernel is requested, the call raises.  `imagescry_amd.build` compiles the
library in-tree with hipcc for gfx950.
"""

from __future__ import annotations

import ctypes
import math
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from pathlib import Path

import torch

LIB_NAME = "libimagescry_hip.so"
# ISC_LIB selects another build of the same ABI (the -DISC_ABLATION library used by scripts/ for ablation runs); it is
# never set in production, and `load()` refuses a library that reports ISC_BUILD_ABLATION unless ISC_ALLOW_ABLATION=1 is
# set as well (its ISC_DEBUG_MODE-style variants return wrong results by design)
LIB_PATH = Path(os.environ.get("ISC_LIB") or Path(__file__).resolve().parent / LIB_NAME)

ISC_U8, ISC_F16, ISC_F32 = 0, 1, 2
ISC_ACT_NONE, ISC_ACT_RELU, ISC_ACT_GELU, ISC_ACT_SILU, ISC_ACT_SIGMOID = 0, 1, 2, 3, 4
ISC_ACT_RESIDUAL_AFTER = 0x100
ISC_TOPK_MAX_K = 120
ISC_SEARCH_MAX_D = 8192
ISC_SEARCH_PASS_QUERIES = 1024  # queries per pass of isc_cosine_topk (the workspace is sized for one pass)
ISC_ABI_VERSION = 4
ISC_BUILD_ABLATION = 1
ISC_GEMM_A_PACKED, ISC_GEMM_W_PACKED, ISC_GEMM_OUT_PACKED, ISC_GEMM_TILE_256, ISC_GEMM_TILE_128 = 1, 2, 4, 8, 16
ISC_KERNEL_DOTS_FILTER, ISC_KERNEL_CONV, ISC_KERNEL_GEMM_F16 = 0, 1, 2

ISC_OK = 0
ISC_ERR_INVALID_ARG = -1
ISC_ERR_UNSUPPORTED = -2
ISC_ERR_WORKSPACE = -3
ISC_ERR_LAUNCH = -4
ISC_ERR_NO_DEVICE = -5
ISC_ERR_ALIGNMENT = -6


class HipLibraryError(RuntimeError):
    """The HIP library is missing, failed to load, or a kernel launch failed."""


# name -> (restype, argtypes); every symbol include/imagescry_hip.h declares
SIGNATURES: dict[str, tuple[object, list[object]]] = {
    "isc_abi_version": (c_int, []),
    "isc_build_flags": (c_int, []),
    "isc_strerror": (c_char_p, [c_int]),
    "isc_device_info": (c_int, [POINTER(c_int), POINTER(c_int), c_char_p, c_int]),
    "isc_timing_enable": (c_int, [c_int]),
    "isc_timing_read": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(c_int)]),
    "isc_channel_stats_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "isc_channel_stats": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "isc_normalize_clip": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_void_p,
         c_void_p],
    ),
    "isc_normalize_clip_nhwc4": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_void_p,
         c_void_p],
    ),
    "isc_resize_bilinear": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isc_l2norm_channels": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "isc_bank_packed_bytes": (c_int, [c_int, c_int64, c_int, POINTER(c_size_t)]),
    "isc_bank_permutation": (c_int, [c_int64, POINTER(c_int64), POINTER(c_int64)]),
    "isc_bank_pack": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_int64, c_int64, c_int64, c_int, c_float, c_void_p, c_int, c_void_p,
         c_void_p],
    ),
    "isc_bank_unpack": (c_int, [c_void_p, c_int, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "isc_nchw_to_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isc_conv2d_nhwc": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
         c_void_p, c_void_p],
    ),
    "isc_conv2d_nhwc_dual": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
         c_int, c_void_p, c_void_p],
    ),
    "isc_conv2d_nhwc_gated": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
         c_void_p, c_int, c_void_p, c_void_p],
    ),
    "isc_dwconv2d_nhwc": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p],
    ),
    "isc_dwconv2d_nhwc_pool": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
         c_void_p, c_void_p],
    ),
    "isc_se_gate": (
        c_int,
        [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p],
    ),
    "isc_linear_centered": (
        c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]
    ),
    "isc_feature_sums_workspace_bytes": (c_int, [c_int64, c_int, POINTER(c_size_t)]),
    "isc_feature_sums": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "isc_center_transpose": (c_int, [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "isc_gram_rows": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p]),
    "isc_im2col_nchw": (
        c_int,
        [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    ),
    "isc_maxpool_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isc_global_avgpool_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isc_pool_linear_l2norm": (
        c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]
    ),
    "isc_gemm_f16": (
        c_int,
        [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p],
    ),
    "isc_layernorm": (
        c_int,
        [c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int64, c_int, c_void_p],
    ),
    "isc_attention_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "isc_patchify_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "isc_vit_assemble": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "isc_cosine_topk_workspace_bytes": (c_int, [c_int, c_int64, c_int, c_int, c_int, POINTER(c_size_t)]),
    "isc_cosine_topk": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p, c_void_p,
         c_void_p, c_void_p, c_size_t, c_void_p],
    ),
    "isc_cosine_topk_exhaustive_workspace_bytes": (c_int, [c_int, c_int64, c_int, c_int, c_int, POINTER(c_size_t)]),
    "isc_cosine_topk_exhaustive": (
        c_int,
        [c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_int, c_int64, c_int, c_int64, c_void_p, c_void_p,
         c_void_p, c_size_t, c_void_p],
    ),
    "isc_topk_merge": (
        c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]
    ),
}

_lib: ctypes.CDLL | None = None


def load() -> ctypes.CDLL:
    """Load the in-tree shared library once and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -m imagescry_amd.build` (hipcc, gfx950). "
            "imagescry_amd has no CPU fallback."
        )
    try:
        lib = ctypes.CDLL(str(LIB_PATH))
    except OSError as exc:  # pragma: no cover - depends on the host's ROCm install
        raise HipLibraryError(f"could not load {LIB_PATH}: {exc}") from exc
    missing = [name for name in SIGNATURES if not hasattr(lib, name)]
    if missing:
        raise HipLibraryError(f"{LIB_PATH} does not export {missing}; rebuild it with `python -m imagescry_amd.build`")
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.isc_abi_version() != ISC_ABI_VERSION:
        raise HipLibraryError(
            f"ABI version mismatch: library reports {lib.isc_abi_version()}, binding expects {ISC_ABI_VERSION}; "
            "rebuild it with `python -m imagescry_amd.build`"
        )
    if lib.isc_build_flags() & ISC_BUILD_ABLATION and os.environ.get("ISC_ALLOW_ABLATION") != "1":
        raise HipLibraryError(
            f"{LIB_PATH} is an ablation build (-DISC_ABLATION: timing variants that return wrong results); it is only "
            "loaded with ISC_ALLOW_ABLATION=1 (scripts/), never by the product"
        )
    _lib = lib
    return lib


def strerror(status: int) -> str:
    return load().isc_strerror(status).decode()


def check(status: int, what: str) -> None:
    """Map a C status code to the exception the reference's Python surface would raise."""
    if status == ISC_OK:
        return
    msg = f"{what}: {strerror(status)} (status {status})"
    if status in (ISC_ERR_INVALID_ARG, ISC_ERR_UNSUPPORTED, ISC_ERR_ALIGNMENT):
        raise ValueError(msg)
    raise HipLibraryError(msg)


def require_device(t: torch.Tensor, name: str) -> None:
    """Kernels only exist for HIP devices; refuse anything else loudly."""
    if t.device.type != "cuda":
        raise HipLibraryError(
            f"{name} is on {t.device}; imagescry_amd runs on MI355X (HIP) devices only and has no CPU fallback"
        )


def ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def stream_handle(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.uint8:
        return ISC_U8
    if dtype == torch.float16:
        return ISC_F16
    if dtype == torch.float32:
        return ISC_F32
    raise TypeError(f"unsupported dtype {dtype}")


INF = math.inf


def timing_enable(enable: bool) -> None:
    check(load().isc_timing_enable(int(enable)), "isc_timing_enable")


def timing_read(kernel_id: int) -> tuple[float, int]:
    """(summed device milliseconds, launches) of one instrumented kernel since the last read."""
    total = ctypes.c_double()
    n = c_int()
    check(load().isc_timing_read(kernel_id, total, n), "isc_timing_read")
    return total.value, n.value
