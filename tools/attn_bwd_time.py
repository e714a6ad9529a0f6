"""tools/attn_bwd_time.py -- forward and backward attention launch times at the training shape (8 x 699 tokens, 32 heads x 128)
and at T = 2048, by HIP events."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt4roi_amd import kernels as K  # noqa: E402

dev = "cuda:0"
for (B, T, H, D) in [(8, 699, 32, 128), (1, 2048, 32, 128), (8, 577, 16, 64)]:
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v, do = (torch.randn(B, T, H * D, generator=g, device=dev).to(torch.bfloat16) for _ in range(4))
    lse = torch.empty((B, H, T), dtype=torch.float32, device=dev)
    o = K.flash_attn(q, k, v, H, 1 / math.sqrt(D), True, lse=lse)
    K.flash_attn_bwd(q, k, v, o, do, lse, H, 1 / math.sqrt(D), True)
    torch.cuda.synchronize()
    def t(fn, n=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / n
    tf = t(lambda: K.flash_attn(q, k, v, H, 1 / math.sqrt(D), True, lse=lse))
    tb = t(lambda: K.flash_attn_bwd(q, k, v, o, do, lse, H, 1 / math.sqrt(D), True))
    fl = 4.0 * B * H * T * T * D * 0.5
    print(f"B {B} T {T} H {H} D {D}: forward {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s)  backward {tb:7.1f} us ({3.5 * fl / tb / 1e6:6.1f} TF/s)  ratio {tb / tf:.2f}")
