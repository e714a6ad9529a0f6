"""tools/vit_bench.py -- the CLIP ViT-L/14 tower alone (23 blocks, 336^2) at a given batch, eager launches, for rocprofv3 passes:
    python tools/vit_bench.py --batch 8 --iters 5"""
import argparse
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt4roi_amd import synthetic as syn           # noqa: E402
from gpt4roi_amd.vit import ClipVisionTower        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = "cuda:0"
v = syn.CLIP_L14
tower = ClipVisionTower(syn.vit_state(v["hidden"], v["inter"], v["layers"], 336, seed=1, device=dev, dtype=torch.bfloat16), heads=v["heads"], device=dev)
img = torch.randn(a.batch, 3, 336, 336, device=dev)
for _ in range(a.iters):
    tower.forward(img)
torch.cuda.synchronize()
print("done")
