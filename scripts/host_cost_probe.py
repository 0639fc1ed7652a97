"""Host time of one EmbeddingBank.search call (enqueue only) against its device time, at a 1.25 M-row shard."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagescry_amd import EmbeddingBank
import bench

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
bank = EmbeddingBank(bench.make_shard(0, n, 768, dev), dtype=torch.float16, normalize=False)
big = EmbeddingBank(bench.make_shard(1, 6_000_000, 768, dev), dtype=torch.float16, normalize=False)
for q in (1, 16, 64):
    qq = torch.randn(q, 768, generator=torch.Generator().manual_seed(5)).half().to(dev)
    for _ in range(5):
        bank.search(qq, 10)
    torch.cuda.synchronize()
    # host cost: enqueue behind a long-running search of another bank, so the device never starves the queue
    big.search(qq, 10); big.search(qq, 10)
    t0 = time.perf_counter()
    iters = 50
    for _ in range(iters):
        bank.search(qq, 10)
    host_us = (time.perf_counter() - t0) / iters * 1e6
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 300
    for _ in range(iters):
        bank.search(qq, 10)
    torch.cuda.synchronize()
    dev_us = (time.perf_counter() - t0) / iters * 1e6
    print(f"N={n} Q={q}: host enqueue {host_us:.1f} us per search, end to end {dev_us:.1f} us per search", flush=True)
