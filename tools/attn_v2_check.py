"""Attention forward, second form (csrc/attention_v2.hip): every instantiated (NWG, NG) variant against an fp32 torch
statement of the op (output and log2-domain LSE), then timed next to the first form (variant 1) on the shapes of the path:
   python tools/attn_v2_check.py [--time-only]
"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K
from gpt4roi_amd import _lib

lib = _lib.lib()
dev = "cuda"


def ref(q, k, v, H, scale, causal):
    B, Tq, HD = q.shape
    Tk, D = k.size(1), HD // H
    qh, kh, vh = (t.float().view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * scale
    if causal:
        i = torch.arange(Tq, device=q.device)[:, None] + (Tk - Tq)
        s = s.masked_fill(torch.arange(Tk, device=q.device)[None, :] > i, float("-inf"))
    lse2 = torch.logsumexp(s, -1) * 1.4426950408889634
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Tq, HD), lse2


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




R = lambda *s: (torch.randn(*s, device=dev) * 0.7).to(torch.bfloat16)
VARS = {128: [42, 142, 41], 64: [24, 124, 42, 142, 41]}
ok = True
if "--time-only" not in sys.argv:
    cases = [(1, 32, 128, 767, 767, True), (1, 16, 64, 577, 577, False), (2, 4, 128, 33, 33, True), (1, 8, 128, 100, 300, True),
             (1, 4, 64, 32, 1000, False), (2, 3, 128, 200, 200, False), (1, 2, 64, 257, 257, True), (3, 16, 64, 577, 577, False),
             (1, 2, 128, 129, 129, True), (1, 2, 128, 64, 2048, True)]
    for (B, H, D, Tq, Tk, causal) in cases:
        q, k, v = R(B, Tq, H * D), R(B, Tk, H * D), R(B, Tk, H * D)
        q[:, : , : D] *= 6.0                                   # head 0: peaked rows (the rescale path)
        want, wlse = ref(q, k, v, H, 1 / math.sqrt(D), causal)
        for var in [1] + VARS[D]:
            lib.g4r_attn_debug_variant(var)
            lse = torch.full((B, H, Tq), float("nan"), dtype=torch.float32, device=dev)
            got = K.flash_attn(q, k, v, H, 1 / math.sqrt(D), causal, lse=lse)
            torch.cuda.synchronize()
            e = (got.float() - want).abs().max().item()
            el = (lse - wlse).abs().max().item()
            good = e < 2e-2 and el < 2e-2 and not torch.isnan(got.float()).any()
            ok &= good
            print(f"{'ok ' if good else 'BAD'} B{B} H{H} D{D} Tq{Tq} Tk{Tk} causal{int(causal)} variant {var}: max |err| {e:.2e}, lse {el:.2e}")
    # strided fused-qkv views (ViT) and a KV-cache view
    B, T, H, D = 1, 577, 16, 64
    qkv = R(B, T, 3 * H * D)
    q, k, v = qkv[:, :, :H * D], qkv[:, :, H * D:2 * H * D], qkv[:, :, 2 * H * D:]
    want, _ = ref(q, k, v, H, 0.125, False)
    for var in VARS[64]:
        lib.g4r_attn_debug_variant(var)
        e = (K.flash_attn(q, k, v, H, 0.125, False).float() - want).abs().max().item()
        ok &= e < 2e-2
        print(f"{'ok ' if e < 2e-2 else 'BAD'} strided qkv views, variant {var}: {e:.2e}")
lib.g4r_attn_debug_variant(0)
for (B, H, D, T, c) in [(1, 16, 64, 577, False), (8, 16, 64, 577, False), (1, 32, 128, 767, True), (8, 32, 128, 699, True), (1, 32, 128, 2048, True)]:
    q, k, v = R(B, T, H * D), R(B, T, H * D), R(B, T, H * D)
    fl = 4.0 * B * H * T * T * D / (2 if c else 1)
    line = f"attn B{B} H{H} D{D} T{T} causal{int(c)}:"
    for var in [1] + VARS[D]:
        lib.g4r_attn_debug_variant(var)
        t = timeit(lambda: K.flash_attn(q, k, v, H, 1 / math.sqrt(D), c))
        line += f"  v{var} {t:.1f} us ({fl / t / 1e6:.0f} TF/s)"
    print(line)
lib.g4r_attn_debug_variant(0)
print("ALL OK" if ok else "FAILURES")
