import torch, math, sys
sys.path.insert(0, ".")
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
for (B, H, D, T, c) in [(1, 16, 64, 577, False), (1, 32, 128, 767, True), (1, 32, 128, 2048, True), (8, 32, 128, 2048, True)]:
    q, k, v = R(B, T, H * D), R(B, T, H * D), R(B, T, H * D)
    t = timeit(lambda: K.flash_attn(q, k, v, H, 1 / math.sqrt(D), c))
    print(f"attn B{B} H{H} D{D} T{T} causal{int(c)}: {t:.1f} us, {4.0*B*H*T*T*D/(2 if c else 1)/t/1e6:.1f} TF/s")
    if D == 128:
        lse = torch.empty((B, H, T), dtype=torch.float32, device="cuda")
        o = K.flash_attn(q, k, v, H, 1 / math.sqrt(D), c, lse=lse)
        do = R(B, T, H * D)
        t = timeit(lambda: K.flash_attn_bwd(q, k, v, o, do, lse, H, 1 / math.sqrt(D), c))
        print(f"attn bwd B{B} H{H} D{D} T{T} causal{int(c)}: {t:.1f} us, {10.0*B*H*T*T*D/(2 if c else 1)/t/1e6:.1f} TF/s")
