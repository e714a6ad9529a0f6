"""tools/roi_bench.py -- the production multi-level NHWC RoIAlign launch alone, at the bench shape (P = 24, C = 1024, 32 RoIs per
image, deferred GroupNorm affine on), for timing and for rocprofv3 counter passes:  python tools/roi_bench.py [--batch B] [--iters N]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt4roi_amd import kernels as K          # noqa: E402
from gpt4roi_amd import synthetic as syn      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rois", type=int, default=32)
a = ap.parse_args()
dev, P, C = "cuda:0", 24, 1024
g = torch.Generator().manual_seed(0)
sizes = [8 * P, 4 * P, 2 * P, P]
feats = [torch.randn(a.batch, s, s, C, generator=g).to(dev).to(torch.bfloat16) for s in sizes]
affs = [torch.randn(a.batch, 2, C, generator=g).to(dev) for _ in sizes]
rois = torch.cat([torch.cat([torch.full((a.rois, 1), float(b)), syn.boxes(a.rois, g) * 14 * P], 1) for b in range(a.batch)]).to(dev)
scales = [8 / 14, 4 / 14, 2 / 14, 1 / 14]
out = K.roi_align_mlvl(feats, rois, 14, scales, affines=affs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    K.roi_align_mlvl(feats, rois, 14, scales, affines=affs, out=out)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / a.iters
alg = sum(f.numel() for f in feats) * 2 + out.numel() * 2 + rois.numel() * 4
print(f"roi_align_mlvl batch {a.batch}: {us:.1f} us per launch, algorithmic bytes {alg / 1e6:.1f} MB -> {alg / us / 1e6:.2f} TB/s")
