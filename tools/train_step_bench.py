#!/usr/bin/env python3
"""Times the stage-1 training step (SURVEY.md 8d config 3 shape at B images per GPU) on the full-size model:
CLIP ViT-L/14 @336 -> region module (trainable) -> LLaMA-7B (frozen) -> CE -> backward -> clip -> AdamW.
Prints one JSON line with ms/step and the per-kernel-family breakdown of one instrumented step."""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gpt4roi_amd import kernels as K  # noqa: E402
from gpt4roi_amd import synthetic as syn  # noqa: E402
from gpt4roi_amd.train import FullTrainer, RegionTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--rois", type=int, default=32)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--llama-layers", type=int, default=32)
ap.add_argument("--stage", type=int, default=1, help="1: region module trainable; 2: everything but the ViT")
ap.add_argument("--by-shape", action="store_true", help="GEMM / conv rows of the breakdown per problem shape, with TFLOP/s")
a = ap.parse_args()
dev = torch.device("cuda:0")
margs = SimpleNamespace(image_size=336, llama_layers=a.llama_layers)
model, ids = bench.build_model(margs, dev, 0)
model.llama._alloc_cache(a.batch)
P = 24
g = torch.Generator().manual_seed(0)
img = torch.randn(a.batch, 3, 336, 336, generator=g).to(dev)
boxes = [syn.boxes(a.rois, g).to(dev) for _ in range(a.batch)]
prompt = torch.stack([syn.prompt_ids(ids, P, a.rois, g) for _ in range(a.batch)]).to(dev)
labels = prompt.clone()
labels[:, :42 + P * P] = -100
labels[labels >= 32000] = -100
t0 = time.time()
tr = (FullTrainer if a.stage == 2 else RegionTrainer)(model, lr=2e-5)
torch.cuda.synchronize()
print(f"trainer ready in {time.time()-t0:.1f}s; mem {torch.cuda.memory_allocated()/2**30:.1f} GiB", flush=True)
losses = [tr.step(prompt, img, boxes, labels).item()]          # warm-up (allocations, plans)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.steps):
    losses.append(tr.step(prompt, img, boxes, labels).item())
torch.cuda.synchronize()
ms = (time.time() - t0) / a.steps * 1e3
K.PROFILER.start(detail=a.by_shape)
tr.step(prompt, img, boxes, labels)
agg = K.PROFILER.stop()
top = sorted(((v["ms"], k, v["calls"]) for k, v in agg.items()), reverse=True)[:14]
if a.by_shape:
    print("per-shape rows of one instrumented step (ms total, calls, avg us, TFLOP/s):")
    for v_ms, k, c in sorted(((v["ms"], k, v["calls"]) for k, v in agg.items()), reverse=True)[:60]:
        fl = agg[k]["flops"]
        print(f"  {v_ms:9.3f} {c:5d} {v_ms / c * 1e3:9.1f} {fl / v_ms / 1e9 if fl else 0:8.1f}  {k}")
print(json.dumps(dict(metric=("stage-2 training step (ViT-L/14@336 frozen; region module, projector and LLaMA-7B trainable)" if a.stage == 2 else
                              "stage-1 training step (ViT-L/14@336 frozen, region module trainable, LLaMA-7B frozen)"),
                      batch=a.batch, rois=a.rois, tokens=int(prompt.size(1)), ms_per_step=round(ms, 2),
                      region_tokens_per_s=round(a.batch * a.rois / ms * 1e3, 1), losses=[round(x, 4) for x in losses],
                      peak_mem_GiB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                      kernels=[dict(name=k, ms=round(m, 3), calls=c) for m, k, c in top])))
