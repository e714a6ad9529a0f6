#!/usr/bin/env python3
"""tools/mall_probe.py -- does a weight matrix that is still in the memory-side cache (Infinity Cache, 256 MB) stream faster than one that
comes from HBM?  The decode GEMV (g4r_gemv_rmsnorm_bf16) over the SAME weights again and again (hot: 33-180 MB stay resident) against
a rotation over enough copies to exceed the cache (cold), each as one captured graph of 24 launches.  Prices a prefetch-into-cache
scheme for the decode step's launch ramps before anyone builds it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K

dev = "cuda"
x = (torch.randn(1, 4096, device=dev) * 0.5).to(torch.bfloat16)
gam = torch.ones(4096, device=dev)
for name, N in (("o_proj 4096x4096 (33.5 MB)", 4096), ("q|k|v 12288x4096 (100 MB)", 12288), ("gate|up 22016x4096 (180 MB)", 22016)):
    ncopies = max(2, int(700e6 // (N * 4096 * 2)) + 1)
    ws = [(torch.randn(N, 4096, device=dev) * 0.02).to(torch.bfloat16) for _ in range(ncopies)]
    out = torch.empty(1, N, dtype=torch.bfloat16, device=dev)
    res = {}
    for mode in ("hot", "cold"):
        seq = [ws[0]] * 24 if mode == "hot" else [ws[i % ncopies] for i in range(24)]
        for w in seq[:ncopies]:
            K.gemv(x, w, norm_weight=gam, eps=1e-6, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for w in seq:
                K.gemv(x, w, norm_weight=gam, eps=1e-6, out=out)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) * 1e3 / (20 * 24)
    mb = N * 4096 * 2 / 1e6
    print(f"{name}: hot {res['hot']:.2f} us ({mb / res['hot']:.2f} TB/s)  cold {res['cold']:.2f} us ({mb / res['cold']:.2f} TB/s)  copies {ncopies}")
