"""Bring-up timing of EmbeddingBank.search (not the contract bench; see bench.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagescry_amd import EmbeddingBank

dev = torch.device("cuda:0")
cases = [(1_000_000, 1024), (1_000_000, 256), (1_000_000, 16), (10_000_000, 1024), (10_000_000, 64)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for n, q in cases:
    g = torch.Generator(device=dev).manual_seed(1)
    bank = torch.empty((n, 768), dtype=torch.float16, device=dev)
    for r0 in range(0, n, 1 << 20):
        blk = torch.randn(min(1 << 20, n - r0), 768, generator=g, device=dev)
        bank[r0:r0 + blk.shape[0]] = torch.nn.functional.normalize(blk, dim=1).half()
    queries = torch.randn(q, 768, generator=g, device=dev).half()
    eb = EmbeddingBank(bank, dtype=torch.float16, normalize=False)
    del bank
    for _ in range(2):
        eb.search(queries, 10)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 5
    e0.record()
    for _ in range(iters):
        eb.search(queries, 10)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * n * q * 768
    print(f"dbg={os.environ.get('ISC_DEBUG_MODE', '0')} N={n} Q={q}: {ms:.3f} ms  {q / ms * 1e3:.0f} q/s  {flops / ms / 1e9:.1f} TFLOP/s  {n * 768 * 2 / ms / 1e6:.1f} GB/s status={eb.last_status.tolist()}", flush=True)
    del eb
