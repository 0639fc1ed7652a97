"""Do the small kernels at the tail of search i overlap the head of search i + 1 when the two run on different streams?
Two banks with the same rows (each has its own workspace), searches alternating between two streams, against one stream."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagescry_amd import EmbeddingBank
import bench

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
rows = bench.make_shard(0, n, 768, dev)
banks = [EmbeddingBank(rows, dtype=torch.float16, normalize=False) for _ in range(2)]
del rows
for q in (1, 16, 64, 1024):
    qq = torch.randn(q, 768, generator=torch.Generator().manual_seed(5)).half().to(dev)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for mode in ("one stream", "two streams"):
        def run(iters):
            for i in range(iters):
                j = i & 1 if mode == "two streams" else 0
                with torch.cuda.stream(streams[j]):
                    banks[j if mode == "two streams" else 0].search(qq, 10)
        run(6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 200
        run(iters)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / iters * 1e6
        print(f"N={n} Q={q} {mode}: {us:.1f} us per search  ({n * 1536 / us / 1e6:.3f} TB/s = {n * 1536 / us / 8e6:.3f} of 8)", flush=True)
