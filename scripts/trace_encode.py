"""One warm predict_step of ResNet50Embedder under rocprofv3 --kernel-trace (see scripts/trace_encode.sh)."""
import sys
import torch
sys.path.insert(0, ".")
from imagescry_amd import ImageBatch, ResNet50Embedder

dev = torch.device("cuda:0")
model = ResNet50Embedder(seed=0).to(dev)
images = torch.randint(0, 256, (512, 3, 224, 224), dtype=torch.uint8).to(dev)
batch = ImageBatch(indices=torch.arange(512, device=dev), images=images)
for _ in range(3):
    model.predict_step(batch)
torch.cuda.synchronize()
