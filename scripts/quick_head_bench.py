"""Timing of the fused ResNet tail (isc_pool_linear_l2norm) at the bench shape: 512 x 7 x 7 x 2048 -> 768."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagescry_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
b, h, w, c, e = 512, 7, 7, 2048, 768
x = torch.randn(b, h, w, c, device=dev)
wt = torch.randn(e, c, device=dev) / c**0.5
bias = torch.randn(e, device=dev)
out = torch.empty(b, e, device=dev)
s = _lib.stream_handle(dev)
run = lambda: _lib.check(lib.isc_pool_linear_l2norm(x.data_ptr(), b, h, w, c, wt.data_ptr(), bias.data_ptr(), e, 1, 1e-12, out.data_ptr(), s), "head")
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print(f"isc_pool_linear_l2norm B={b}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per launch")
