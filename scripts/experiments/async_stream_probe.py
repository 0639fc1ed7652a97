import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from imagescry_amd import EmbeddingBank
import bench
dev = torch.device("cuda:0")
n = 1_250_000
bank = EmbeddingBank(bench.make_shard(0, n, 768, dev), dtype=torch.float16, normalize=False)
def serial(iters, qq):
    for _ in range(iters): bank.search(qq, 10)
def streamed(iters, qq):
    pending = None
    for _ in range(iters):
        h = bank.search_async(qq, 10)
        if pending is not None: pending.result()
        pending = h
    pending.result()
def streamed_keep(iters, qq):
    hs = []
    for _ in range(iters):
        hs.append(bank.search_async(qq, 10))
        if len(hs) > 1: hs[-2].result()
    hs[-1].result()
    return hs
for q in (1, 64):
    qq = torch.randn(q, 768, generator=torch.Generator().manual_seed(5)).half().to(dev)
    for name, fn in (("serial", serial), ("streamed", streamed), ("streamed, results kept", streamed_keep)):
        fn(8, qq); torch.cuda.synchronize()
        for iters in (20, 300):
            t0 = time.perf_counter(); r = fn(iters, qq); t1 = time.perf_counter(); torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / iters * 1e6
            print(f"Q={q} {name:24s} iters={iters:3d}: {us:6.1f} us per search (host enqueue {(t1 - t0) / iters * 1e6:.1f})", flush=True)
            del r
