"""Bring-up timing of isc_normalize_clip_nhwc4 ([B, H, W, 4] float32 out: the four-pixels-per-thread form on 16-byte-aligned input
against the one-pixel form, input pointer off by one byte) and of isc_normalize_clip ([B, C, H, W] float32 out) at the encoders'
batch, 512 x 3 x 224 x 224 uint8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagescry_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.load()
b, c, h, w = 512, 3, 224, 224
raw = torch.randint(0, 256, (b * c * h * w + 16,), dtype=torch.uint8, device=dev)
mean = torch.tensor([120.0, 118.0, 121.0], device=dev)
std = torch.tensor([70.0, 71.0, 72.0], device=dev)
y = torch.empty((b, h, w, 4), device=dev)
s = _lib.stream_handle(dev)
for off, name in ((0, "aligned (PX = 4)"), (1, "off by one byte (PX = 1)")):
    x = raw[off:off + b * c * h * w]
    def run():
        _lib.check(lib.isc_normalize_clip_nhwc4(x.data_ptr(), _lib.ISC_U8, b, c, h, w, mean.data_ptr(), std.data_ptr(), 1, 1e-6,
                                                -3.0, 3.0, y.data_ptr(), s), "normalize")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"{name}: {us:.1f} us  {(b*c*h*w + y.numel()*4)/us/1e6:.2f} TB/s", flush=True)

y2 = torch.empty((b, c, h, w), device=dev)
x = raw[: b * c * h * w]
def run2():
    _lib.check(lib.isc_normalize_clip(x.data_ptr(), _lib.ISC_U8, b, c, h, w, mean.data_ptr(), std.data_ptr(), 1, 1e-6, -3.0, 3.0,
                                      y2.data_ptr(), s), "normalize")
for _ in range(3): run2()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run2()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 50
print(f"NCHW float32 out: {us:.1f} us  {(b*c*h*w + y2.numel()*4)/us/1e6:.2f} TB/s", flush=True)
