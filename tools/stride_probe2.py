"""LDS-DMA vs register-staged global->LDS rate (loads-only ablation), 128x128 tile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K, _lib
lib = _lib.lib
dev = "cuda"
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for (M, N, Kd) in [(4096, 4096, 4096), (768, 12288, 4096)]:
    a = torch.randn(M, Kd, device=dev).bfloat16(); w = torch.randn(N, Kd, device=dev).bfloat16()
    for tile in (0, 2, 10, 1, 9):
        row = []
        for mode, name in ((0, "full"), (1, "mfma"), (2, "loads")):
            lib().g4r_gemm_debug_mode(mode)
            t = timeit(lambda: K.gemm(a, w, tile_cfg=tile))
            lib().g4r_gemm_debug_mode(0)
            row.append(f"{name} {t*1e6:6.1f}us {2.0*M*N*Kd/t/1e12:6.0f}TF")
        print(f"{M}x{N}x{Kd} tile{tile} ({K.TILE_NAMES.get(tile)}): " + " | ".join(row), flush=True)
