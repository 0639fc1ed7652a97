import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from imagescry_amd import EmbeddingBank
import bench
dev = torch.device("cuda:0")
small = EmbeddingBank(bench.make_shard(0, 1_250_000, 768, dev), dtype=torch.float16, normalize=False)
for r in bench.query_sweep(small, 1_250_000, 768, 10, dev, qs=(1, 64)):
    print(r["queries"], r["ms_per_search"], r["ms_per_search_streamed"], flush=True)
big = EmbeddingBank(bench.make_shard(0, 10_000_000, 768, dev), dtype=torch.float16, normalize=False)
qq = torch.randn(1024, 768).half().to(dev)
for _ in range(3): big.search(qq, 10)
torch.cuda.synchronize()
for r in bench.query_sweep(small, 1_250_000, 768, 10, dev, qs=(1, 64)):
    print("after big:", r["queries"], r["ms_per_search"], r["ms_per_search_streamed"], flush=True)
