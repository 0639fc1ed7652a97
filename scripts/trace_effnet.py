"""Warm predict_steps of EfficientNetEmbedder("s") under rocprofv3 --kernel-trace (see scripts/trace_effnet_layers.sh)."""
import sys
import torch
sys.path.insert(0, ".")
from imagescry_amd import EfficientNetEmbedder, ImageBatch

dev = torch.device("cuda:0")
model = EfficientNetEmbedder(backbone_size="s", seed=0).to(dev)
images = torch.randint(0, 256, (512, 3, 224, 224), dtype=torch.uint8).to(dev)
batch = ImageBatch(indices=torch.arange(512, device=dev), images=images)
for _ in range(3):
    model.predict_step(batch)
torch.cuda.synchronize()
