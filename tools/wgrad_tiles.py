#!/usr/bin/env python3
"""Tile / split-K choice for the weight-gradient GEMMs: [1024 x 1024 x K] with K = pixels of one level."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K
def timeit(fn, iters=8, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
R = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
for Kd in (38848, 9856, 2624, 704):
    a, w = R(1024, Kd + 448)[:, 24:24 + Kd], R(1024, Kd + 448)[:, 224:224 + Kd]
    for tile, splits in ((4, 8), (0, 4), (0, 8), (10, 8), (9, 8), (9, 16), (22, 8), (22, 16), (1, 8), (1, 16)):
        if splits * 64 > Kd: continue
        t = timeit(lambda: K.gemm(a, w, out_dtype=torch.float32, splits=splits, tile_cfg=tile))
        print(f"K={Kd} tile{tile} splits{splits}: {t:8.1f} us {2.0*1024*1024*Kd/t/1e6:7.1f} TF/s", flush=True)
