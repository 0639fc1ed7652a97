"""Bring-up timing of the embedders' predict_step (not the contract bench; see bench.py)."""
import sys, time
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from imagescry_amd import EfficientNetEmbedder, ImageBatch, ResNet50Embedder, ViTB16Embedder, _lib, vit

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "effnet_s"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 512
if len(sys.argv) > 3:  # resnet50 only: stages whose projection shortcut runs inside conv3, e.g. "1,2" (A/B aid)
    from imagescry_amd import resnet50
    resnet50.FUSED_SHORTCUT_STAGES = tuple(int(v) for v in sys.argv[3].split(",") if v)
model = (ResNet50Embedder() if which == "resnet50" else ViTB16Embedder() if which == "vit_b16"
         else EfficientNetEmbedder(backbone_size=which.split("_")[1])).to(dev)
kid = _lib.ISC_KERNEL_GEMM_F16 if which == "vit_b16" else _lib.ISC_KERNEL_CONV
images = torch.randint(0, 256, (b, 3, 224, 224), dtype=torch.uint8).to(dev)
batch = ImageBatch(indices=torch.arange(b, device=dev), images=images)
for _ in range(2):
    model.predict_step(batch)
torch.cuda.synchronize()
_lib.timing_enable(True); _lib.timing_read(kid)
t0 = time.perf_counter()
for _ in range(3):
    out = model.predict_step(batch)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
ms, n = _lib.timing_read(kid)
print(f"{which} B={b}: {dt*1e3:.2f} ms/step  {b/dt:.0f} img/s  MFMA kernels {ms/3:.2f} ms over {n//3} launches  out {tuple(out.embeddings.shape)}")
if which == "vit_b16":
    fl = vit.gemm_flops() * b
    print(f"  {fl/1e12:.2f} TFLOP/step  -> {fl/dt/1e12:.0f} TFLOP/s end to end, GEMM kernels alone {fl/(ms/3*1e-3)/1e12:.0f} TFLOP/s (upper bound: attention flops included)")
