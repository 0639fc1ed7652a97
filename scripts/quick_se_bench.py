"""Bring-up timing of isc_se_gate (B = 512): `python scripts/quick_se_bench.py C,S ...`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagescry_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
for c in sys.argv[1:] or ["1536,64", "960,40", "512,32"]:
    C, S = (int(v) for v in c.split(","))
    B = 512
    pooled = torch.randn(B, C, device=dev)
    w1 = torch.randn(S, C, device=dev) * 0.02
    b1 = torch.randn(S, device=dev)
    w2 = torch.randn(C, S, device=dev) * 0.1
    b2 = torch.randn(C, device=dev)
    gate = torch.empty(B, C, device=dev)
    s = _lib.stream_handle(dev)
    def run():
        _lib.check(lib.isc_se_gate(pooled.data_ptr(), B, C, w1.data_ptr(), C, b1.data_ptr(), S, w2.data_ptr(), S,
                                   b2.data_ptr(), gate.data_ptr(), s), "se")
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print(f"abl={os.environ.get('ISC_SE_ABL', '0')} C={C} S={S}: {e0.elapsed_time(e1) * 20:.1f} us", flush=True)
