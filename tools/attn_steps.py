"""Step time and fixed cost of the attention forward variants: Tq = 767 query rows (the LLaMA prefill grid: 6 x 32 workgroups),
non-causal, Tk = 128 .. 2048 keys (every workgroup walks Tk / 128 steps with two key groups) -> launch time against steps,
least-squares slope (us per step) and intercept (fixed us per launch), per variant.  hipGraph-replay timing."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K, _lib
lib = _lib.lib()
R = lambda *s: (torch.randn(*s, device="cuda") * 0.7).to(torch.bfloat16)


def timeit(fn, iters=20, warm=3):
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


for (H, D, Tq, variants) in [(32, 128, 767, [1, 142, 42, 242]), (16, 64, 577, [1, 124, 24, 224, 142, 242])]:
    q = R(1, Tq, H * D)
    tks = [128, 256, 512, 1024, 2048]
    kv = {tk: (R(1, tk, H * D), R(1, tk, H * D)) for tk in tks}
    for var in variants:
        lib.g4r_attn_debug_variant(var)
        ts = []
        for tk in tks:
            k, v = kv[tk]
            ts.append(timeit(lambda: K.flash_attn(q, k, v, H, 1 / math.sqrt(D), False)))
        n = len(tks)
        xs = [tk / 64 for tk in tks]                      # key tiles walked by a workgroup
        mx, my = sum(xs) / n, sum(ts) / n
        slope = sum((x - mx) * (y - my) for x, y in zip(xs, ts)) / sum((x - mx) ** 2 for x in xs)
        icpt = my - slope * mx
        fl = lambda tk: 4.0 * H * Tq * tk * D
        print(f"D{D} Tq{Tq} variant {var:3d}: " + "  ".join(f"Tk{tk} {t:6.1f}us" for tk, t in zip(tks, ts))
              + f"  | {slope:.3f} us per 64-key tile, fixed {icpt:.1f} us, Tk2048: {fl(2048) / ts[-1] / 1e6:.0f} TF/s")
lib.g4r_attn_debug_variant(0)
