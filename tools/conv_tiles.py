#!/usr/bin/env python3
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
w = R(1024, 9 * 1024)
for Hs in (192, 96, 48, 24):
    x = R(1, Hs, Hs, 1024)
    for tile in (0, 1, 9, 22, 24, 4):
        t = timeit(lambda: K.conv3x3(x, w, tile_cfg=tile))
        print(f"conv {Hs}: tile{tile} {t:8.1f} us {2.0*Hs*Hs*1024*9216/t/1e6:7.1f} TF/s", flush=True)
x = R(4, 32, 14, 14, 1024); w4 = R(1024, 4 * 9 * 1024)
for tile in (0, 1, 9, 22, 24, 4):
    t = timeit(lambda: K.conv3x3(x, w4, groups=4, tile_cfg=tile))
    print(f"pconv: tile{tile} {t:8.1f} us {2.0*32*196*1024*4*9216/t/1e6:7.1f} TF/s", flush=True)
a = R(36864, 1088); w1 = R(1024, 1088)
for tile in (0, 1, 9, 22, 24):
    t = timeit(lambda: K.gemm(a, w1, tile_cfg=tile))
    print(f"1x1 192: tile{tile} {t:8.1f} us {2.0*36864*1024*1088/t/1e6:7.1f} TF/s", flush=True)
