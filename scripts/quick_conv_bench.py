"""Bring-up timing of isc_conv2d_nhwc on single layers: `python scripts/quick_conv_bench.py B,H,W,Cin,Cout,k,stride,res,act ...`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagescry_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
cases = sys.argv[1:] or ["512,14,14,256,1024,1,1,1,1", "512,14,14,256,1024,1,1,0,1", "512,14,14,256,1024,1,1,0,0",
                         "512,28,28,128,512,1,1,1,1", "512,28,28,128,512,1,1,0,1", "512,7,7,512,2048,1,1,1,1",
                         "512,7,7,512,2048,1,1,0,1"]
for c in cases:
    b, h, w, cin, cout, k, stride, res, act = (int(v) for v in c.split(","))
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x = torch.randn(b, h, w, cin, device=dev)
    kk = k * k * cin
    wt = torch.randn(cout, (kk + 31) // 32 * 32, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    r = torch.randn(b, ho, wo, cout, device=dev) if res else None
    out = torch.empty(b, ho, wo, cout, device=dev)
    s = _lib.stream_handle(dev)
    def run():
        _lib.check(lib.isc_conv2d_nhwc(x.data_ptr(), b, h, w, cin, wt.data_ptr(), cout, k, k, stride, pad, bias.data_ptr(),
                                       _lib.ptr(r), act, out.data_ptr(), s), "conv")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * b * ho * wo * kk * cout
    by = (x.numel() + out.numel() * (2 if res else 1)) * 4
    print(f"{c:36s} {us:8.1f} us  {fl/us/1e6:6.1f} TFLOP/s ({fl/us/1e6/157.3:.2f})  {by/us/1e3:6.0f} GB/s", flush=True)
