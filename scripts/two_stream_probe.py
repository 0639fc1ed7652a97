"""Do the small kernels at the tail of search i overlap the head of search i + 1 when the two run on different streams?
Searches alternate between two streams (each needs its own workspace: two bank objects) against one stream.  Variants:
the two objects hold separate copies of the rows or share one packed image; the streams free-run or are ordered against
the caller's stream by events the way an async handle API would (ready: caller -> stream, done: stream -> caller)."""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagescry_amd import EmbeddingBank
import bench

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
rows = bench.make_shard(0, n, 768, dev)
two = [EmbeddingBank(rows, dtype=torch.float16, normalize=False) for _ in range(2)]
del rows
shared = [two[0], copy.copy(two[0])]
shared[1]._workspaces = {}  # same packed rows, its own workspaces
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
cur = torch.cuda.current_stream(dev)


def run(banks, mode, iters, qq):
    done = [None, None]
    for i in range(iters):
        j = i & 1
        if mode == "one stream":
            banks[0].search(qq, 10)
            continue
        if mode == "events":
            ready = torch.cuda.Event()
            ready.record(cur)
            streams[j].wait_event(ready)
        with torch.cuda.stream(streams[j]):
            banks[j].search(qq, 10)
            if mode == "events":
                ev = torch.cuda.Event()
                ev.record(streams[j])
        if mode == "events":
            if done[j ^ 1] is not None:
                cur.wait_event(done[j ^ 1])  # resolve the search before this one, as a pipelined caller does
            done[j] = ev


for q in (1, 16, 64):
    qq = torch.randn(q, 768, generator=torch.Generator().manual_seed(5)).half().to(dev)
    for name, banks in (("two copies", two), ("shared rows", shared)):
        for mode in ("one stream", "free-running", "events"):
            run(banks, mode, 6, qq)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            iters = 300
            run(banks, mode, iters, qq)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / iters * 1e6
            print(f"N={n} Q={q} {name:11s} {mode:12s}: {us:6.1f} us per search = {n * 1536 / us / 8e6:.3f} of 8 TB/s", flush=True)
