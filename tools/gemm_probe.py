#!/usr/bin/env python3
"""tools/gemm_probe.py -- a handful of launches of the GEMM / conv kernels for rocprofv3 PMC passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K
dev = "cuda"
torch.manual_seed(0)
R = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
tiles = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "8"])]
for (M, N, Kd) in [(4096, 4096, 4096), (768, 12288, 4096), (768, 22016, 4096), (768, 4096, 4096)]:
    a, w = R(M, Kd), R(N, Kd)
    for t in tiles:
        for _ in range(3):
            K.gemm(a, w, tile_cfg=t)
x = R(1, 192, 192, 1024); w = R(1024, 9 * 1024)
for t in tiles:
    for _ in range(2):
        K.conv3x3(x, w, tile_cfg=t)
torch.cuda.synchronize()
