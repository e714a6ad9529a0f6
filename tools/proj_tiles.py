#!/usr/bin/env python3
"""Tile / split-K choice for the N = 4096 LLaMA projections at M = 767 (o_proj K = 4096, down_proj K = 11008)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
R = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
flush = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
for (M, N, Kd) in [(767, 4096, 4096), (767, 4096, 11008)]:
    a, res = R(M, Kd), R(M, N)
    ws_ = [R(N, Kd) for _ in range(8)]                     # rotate weights: no L2 reuse between iterations
    it = [0]
    def run(tile, splits):
        it[0] += 1
        return K.gemm(a, ws_[it[0] % 8], residual=res, tile_cfg=tile, splits=splits)
    t = timeit(lambda: K.gemm(a, ws_[0], residual=res))
    print(f"{M}x{N}x{Kd} default: {t:7.1f} us", flush=True)
    for tile, splits in ((0, 1), (0, 2), (7, 1), (7, 2), (10, 2), (24, 2), (24, 3), (24, 4), (24, 5), (22, 4), (4, 2), (4, 3)):
        t = timeit(lambda: run(tile, splits))
        print(f"{M}x{N}x{Kd} tile{tile} splits{splits}: {t:7.1f} us {2.0*M*N*Kd/t/1e6:6.0f} TF/s", flush=True)
