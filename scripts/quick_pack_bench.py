"""Timing of isc_bank_pack through EmbeddingBank construction (1 M-row blocks), fp16 and fp32 input."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagescry_amd import EmbeddingBank

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
for dt in (torch.float16, torch.float32):
    rows = torch.randn(n, 768, device=dev, dtype=dt)
    for norm in (False, True):
        EmbeddingBank(rows[: 1 << 20], dtype=torch.float16, normalize=norm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b = EmbeddingBank(rows, dtype=torch.float16, normalize=norm)
        torch.cuda.synchronize(); dtm = time.perf_counter() - t0
        gb = n * 768 * (rows.element_size() + 2) / 1e9
        print(f"pack {n} x 768 {dt} -> f16 normalize={norm}: {dtm*1e3:.1f} ms  {gb/dtm:.0f} GB/s  bound={float(b._norm_bound):.4f}", flush=True)
        del b
    del rows
