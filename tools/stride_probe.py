"""Does the row stride of A / W (power of two vs padded) change the L2 -> LDS rate of the GEMM?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K
from gpt4roi_amd import _lib
lib = _lib.lib

dev = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


for (M, N, Kd) in [(4096, 4096, 4096), (768, 12288, 4096), (768, 22016, 4096), (768, 4096, 11008)]:
    for pad in (0, 64, 128, 192):
        a = torch.randn(M, Kd + pad, device=dev).bfloat16()[:, :Kd]
        w = torch.randn(N, Kd + pad, device=dev).bfloat16()[:, :Kd]
        row = []
        for tile in (0, 9, 22):
            for mode, name in ((0, "full"), (2, "loads")):
                if tile == 22 and mode:
                    continue
                lib().g4r_gemm_debug_mode(mode)
                t = timeit(lambda: K.gemm(a, w, tile_cfg=tile))
                lib().g4r_gemm_debug_mode(0)
                row.append(f"t{tile} {name} {t*1e6:6.1f}us {2.0*M*N*Kd/t/1e12:6.0f}TF")
        print(f"{M}x{N}x{Kd} pad{pad}: " + " | ".join(row), flush=True)
