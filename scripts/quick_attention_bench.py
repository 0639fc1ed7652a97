"""Bring-up timing of isc_attention_f16 at the ViT-B/16 batch-512 shape (not the contract bench)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagescry_amd import _lib
from imagescry_amd.vit import packed_elems

dev = torch.device("cuda:0")
lib = _lib.load()
b, t, heads, hd = 512, 197, 12, 64
d = heads * hd
g = torch.Generator(device=dev).manual_seed(0)
for packed in (1, 0):
    n = packed_elems(b * t, 3 * d) if packed else b * t * 3 * d
    qkv = (torch.randn(n, generator=g, device=dev) * 0.5).half()
    out = torch.empty(packed_elems(b * t, d) if packed else b * t * d, dtype=torch.float16, device=dev)
    s = _lib.stream_handle(dev)
    for _ in range(3):
        _lib.check(lib.isc_attention_f16(qkv.data_ptr(), b, t, heads, hd, out.data_ptr(), packed, s), "att")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        lib.isc_attention_f16(qkv.data_ptr(), b, t, heads, hd, out.data_ptr(), packed, s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / it * 1e3
    fl = 4.0 * t * t * hd * b * heads
    print(f"attention packed={packed}: {us:.1f} us  {fl / us / 1e6:.1f} TFLOP/s  abl={os.environ.get('ISC_ATT_ABL', '0')}", flush=True)
