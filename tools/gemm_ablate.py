#!/usr/bin/env python3
"""tools/gemm_ablate.py -- loads-only / MFMA-only ablation of the GEMM main loop (cdna guide 5.5)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K, _lib
dev = "cuda"
torch.manual_seed(0)
R = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
lib = _lib.lib()
for (M, N, Kd) in [(4096, 4096, 4096), (8192, 8192, 4096), (768, 12288, 4096), (768, 22016, 4096)]:
    a, w = R(M, Kd), R(N, Kd)
    for tile in (0, 9, 22, 24):
        row = []
        for mode in (0, 1, 2):
            lib.g4r_gemm_debug_mode(mode)
            t = timeit(lambda: K.gemm(a, w, tile_cfg=tile))
            name = {0: 'full', 1: 'mfma-only', 2: 'loads-only', 3: 'W-only', 5: 'A-only'}[mode]
            row.append(f"{name} {t*1e6:7.1f}us {2.0*M*N*Kd/t/1e12:7.1f}TF")
        lib.g4r_gemm_debug_mode(0)
        print(f"{M}x{N}x{Kd} tile{tile}: " + " | ".join(row), flush=True)
x = R(1, 192, 192, 1024); w = R(1024, 9 * 1024)
for tile in (9, 22, 24):
    t = timeit(lambda: K.conv3x3(x, w, tile_cfg=tile), iters=5)
    print(f"conv3x3 192 tile{tile}: {t*1e6:.1f}us {2.0*192*192*1024*9216/t/1e12:.1f}TF", flush=True)
