"""Ablation timing of csrc/attention_v2.hip (results are WRONG under ablation; only the time is read):
bits 1 = no V pieces, 2 = no K pieces, 4 = no softmax, 8 = no PV, 16 = no QK."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K
from gpt4roi_amd import _lib
lib = _lib.lib()
def timeit(fn, iters=20, warm=3):
    """GPU time per launch: `iters` launches captured in one hipGraph and replayed (an eager Python loop is host-bound at
    ~10 us per ctypes launch, which hides anything shorter)."""
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=side):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
R = lambda *s: (torch.randn(*s, device="cuda") * 0.7).to(torch.bfloat16)
for (B, H, D, T, c, var) in [(1, 32, 128, 767, True, 42), (1, 32, 128, 2048, True, 42), (1, 16, 64, 577, False, 24), (8, 16, 64, 577, False, 42)]:
    q, k, v = R(B, T, H * D), R(B, T, H * D), R(B, T, H * D)
    line = f"B{B} H{H} D{D} T{T} c{int(c)} v{var}:"
    for bits in [0, 31, 31 + 32, 31 + 64, 31 + 128, 31 + 96, 255]:
        lib.g4r_attn_debug_variant(var + 1000 * bits)
        t = timeit(lambda: K.flash_attn(q, k, v, H, 1 / math.sqrt(D), c))
        line += f"  [{bits}] {t:.1f}"
    print(line)
lib.g4r_attn_debug_variant(0)
