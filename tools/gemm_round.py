#!/usr/bin/env python3
"""tools/gemm_round.py -- does a 256x128 ring GEMM sized to whole rounds of the 256 CUs + a split-K
remainder beat the 128x128 kernel on the LLaMA prefill shapes?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
R = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
M = 767
a = R(M, 4096)
for N in (10880, 12288, 21760, 22016, 4096):
    w = R(N, 4096)
    for tile in (0, 8, 1):
        t = timeit(lambda: K.gemm(a, w, tile_cfg=tile))
        print(f"N={N:6d} tile{tile}: {t:7.1f} us  {2.0*M*N*4096/t/1e6:7.1f} TF/s  tiles128={-(-M//128)*-(-N//128)} tiles256x128={-(-M//256)*-(-N//128)}", flush=True)
for N, sp in ((1408, 8), (1408, 4), (256, 8), (256, 16)):
    w = R(N, 4096)
    for tile in (0, 4, 8):
        t = timeit(lambda: K.gemm(a, w, tile_cfg=tile, splits=sp))
        print(f"remainder N={N} splits={sp} tile{tile}: {t:.1f} us", flush=True)
a2 = R(M, 11008)
w = R(4096, 11008)
for tile, sp in ((7, 1), (8, 2), (8, 3), (1, 2), (0, 1), (0, 2)):
    t = timeit(lambda: K.gemm(a2, w, tile_cfg=tile, splits=sp))
    print(f"down 4096x11008 tile{tile} splits{sp}: {t:.1f} us {2.0*M*4096*11008/t/1e6:.1f} TF/s", flush=True)
w = R(4096, 4096)
for tile, sp in ((4, 1), (8, 2), (8, 3), (1, 2), (0, 2)):
    t = timeit(lambda: K.gemm(a, w, tile_cfg=tile, splits=sp))
    print(f"O 4096x4096 tile{tile} splits{sp}: {t:.1f} us {2.0*M*4096*4096/t/1e6:.1f} TF/s", flush=True)
