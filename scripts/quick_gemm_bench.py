"""Bring-up timing of isc_gemm_f16 on the ViT-B/16 shapes (M = 512 * 197 tokens)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagescry_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
m = 512 * 197
for name, k, n, act, res in (("qkv", 768, 2304, 0, False), ("proj", 768, 768, 0, True), ("fc1", 768, 3072, 2, False),
                             ("fc2", 3072, 768, 0, True)):
    pk = os.environ.get("GEMM_LAYOUT", "packed") == "packed"
    from imagescry_amd.vit import pack_rows, packed_elems
    a = torch.randn(m, k, device=dev).half()
    w = (torch.randn(n, k, device=dev) * 0.02).half()
    if pk:
        a, w = pack_rows(a), pack_rows(w)
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if res else None
    out = torch.empty(packed_elems(m, n) if pk else m * n, dtype=torch.float32 if res else torch.float16, device=dev)
    flags = (3 | (0 if res else 4)) if pk else 0
    if os.environ.get("GEMM_TILE") == "256" and act == 0:
        flags |= 8
    if os.environ.get("GEMM_TILE") == "128":
        flags |= 16
    s = _lib.stream_handle(dev)
    def run():
        _lib.check(lib.isc_gemm_f16(a.data_ptr(), m, k, w.data_ptr(), n, b.data_ptr(), _lib.ptr(r), act, out.data_ptr(),
                                    _lib.ISC_F32 if res else _lib.ISC_F16, flags, s), "gemm")
    for _ in range(3): run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): run()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name:5s} M={m} K={k} N={n}: {dt*1e3:.3f} ms  {2*m*k*n/dt/1e12:.0f} TFLOP/s  ({2*m*k*n/dt/2.5e15:.3f} of peak)", flush=True)
    if os.environ.get("GEMM_VENDOR_REF"):
        # A known-good reference on the same device and the same random data (cdna_hip_programming.md rule 10: no ceiling
        # claims from one's own attempts): the vendor library's plain fp16 GEMM of the shape, no bias / activation /
        # residual -- MEASUREMENT ONLY, never a product path.
        a2 = torch.randn(m, k, device=dev).half()
        w2 = (torch.randn(n, k, device=dev) * 0.02).half()
        for _ in range(3): torch.matmul(a2, w2.T)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): torch.matmul(a2, w2.T)
        torch.cuda.synchronize(); dv = (time.perf_counter() - t0) / 10
        print(f"      vendor fp16 GEMM (torch.matmul, fp16 out, no epilogue): {dv*1e3:.3f} ms  ({2*m*k*n/dv/2.5e15:.3f} of peak)", flush=True)
