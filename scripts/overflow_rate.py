"""How often does a search need its second pass on an iid bank?  Many query seeds, status words summed.
    python scripts/overflow_rate.py [rows] [queries] [seeds] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagescry_amd import EmbeddingBank
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 512
seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 20
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
bank = EmbeddingBank(bench.make_shard(0, n, 768, dev), dtype=torch.float16, normalize=False)
tot = [0, 0, 0]
t_all = 0.0
for sd in range(seeds):
    qq = torch.randn(q, 768, generator=torch.Generator().manual_seed(1000 + sd)).half().to(dev)
    bank.search(qq, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    bank.search(qq, k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0; t_all += dt
    if os.environ.get('VERBOSE'): print(f'  seed {sd}: {dt*1e3:.2f} ms status {bank.last_status.tolist()}', flush=True)
    st = bank.last_status.tolist()
    tot[0] += st[0]; tot[1] += st[1]; tot[2] += st[3]
print(f"N={n} Q={q} k={k}: {seeds} query sets: overflowed buffers {tot[0]}, queries searched again {tot[1]} "
      f"({tot[1] / (seeds * q) * 100:.3f} %), exhaustive {tot[2]}; mean {t_all / seeds * 1e3:.2f} ms per search")
