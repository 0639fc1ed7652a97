"""Bring-up timing of the EfficientNetV2-S encode step (bench.py's `encode_efficientnet_v2_s` line alone)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = argparse.Namespace(batch=512)
print(json.dumps(bench.bench_encode_efficientnet(args, torch.device("cuda:0"), steps=3, warmup=1)), flush=True)
