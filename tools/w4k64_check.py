#!/usr/bin/env python3
"""tools/w4k64_check.py -- the round-5 one-wave-per-SIMD GEMM tile (tile_cfg 34) against the ring ping-pong tile (24) and
fp32 torch arithmetic: dense shapes (ragged edges, one / two / three K tiles, K slices, every epilogue), the implicit-GEMM
3x3 convolution (single map, groups, all pyramid levels in one launch), then timings (burst of 8, median of 5) of tiles 24 /
26 / 34 and the vendor library on the bench's shapes.   python tools/w4k64_check.py [--no-time]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402
from gpt4roi_amd._lib import lib  # noqa: E402

dev = "cuda:0"
torch.cuda.set_device(0)
g = torch.Generator(device=dev).manual_seed(5)
R = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)    # noqa: E731
bad = 0


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-9)).item()


def report(name, got, ref24, ref32, tol=1.2e-2, exact=True):
    global bad
    e24 = rel(got, ref24) if ref24 is not None else -1.0
    e32 = rel(got, ref32)
    ok = e32 < tol and (not exact or e24 == 0.0)
    bad += not ok
    print(json.dumps({"case": name, "vs_tile24": e24, "vs_fp32": round(e32, 5), "ok": ok}), flush=True)


# ---- dense ----
for (M, N, Kd) in [(300, 520, 128), (767, 1024, 4096), (1000, 300, 64), (256, 256, 192), (513, 4096, 1024), (2049, 777 * 8, 320),
                   (12272, 4096, 4096)]:
    a, w = R(M, Kd), R(N, Kd)
    ref32 = a.float() @ w.float().t()
    report(f"dense {M}x{N}x{Kd}", K.gemm(a, w, tile_cfg=34), K.gemm(a, w, tile_cfg=24), ref32)
a, w = R(900, 1024), R(768, 1024)
bias = torch.randn(768, device=dev, generator=g)
res = R(900, 768)
for act in (None, "relu", "quick_gelu", "silu"):
    r32 = a.float() @ w.float().t() + bias
    r32 = {None: r32, "relu": r32.relu(), "quick_gelu": r32 * torch.sigmoid(1.702 * r32), "silu": torch.nn.functional.silu(r32)}[act]
    report(f"epilogue bias+{act}+residual", K.gemm(a, w, bias=bias, residual=res, act=act, tile_cfg=34),
           K.gemm(a, w, bias=bias, residual=res, act=act, tile_cfg=24), r32 + res.float())
report("fp32 out", K.gemm(a, w, out_dtype=torch.float32, tile_cfg=34), K.gemm(a, w, out_dtype=torch.float32, tile_cfg=24),
       a.float() @ w.float().t(), tol=1e-5)
report("swiglu", K.gemm(a, w, act="swiglu", tile_cfg=34), K.gemm(a, w, act="swiglu", tile_cfg=24),
       K.gemm(a, w, act="swiglu", tile_cfg=0), tol=1e-2)
a, w = R(767, 11008), R(4096, 11008)
report("K slices x4 (767x4096x11008)", K.gemm(a, w, splits=4, tile_cfg=34), None, a.float() @ w.float().t(), exact=False)
a, w = R(300, 384), R(520, 384)
report("K slices x3 (300x520x384)", K.gemm(a, w, splits=3, tile_cfg=34), None, a.float() @ w.float().t(), exact=False)

# ---- fused q|k|v projection + RoPE + KV-cache append (epilogue mode 5) ----
for (B, T, heads) in [(1, 767, 32), (3, 300, 4)]:
    HD = heads * 128
    h, wqkv = R(B * T, 512), R(3 * HD, 512)
    ang = torch.rand(1024, 64, device=dev, generator=g) * 6.28
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    outs = {}
    for t in (24, 34):
        q = torch.zeros(B, T, HD, dtype=torch.bfloat16, device=dev)
        kc = torch.zeros(B, 1024, HD, dtype=torch.bfloat16, device=dev)
        vc = torch.zeros(B, 1024, HD, dtype=torch.bfloat16, device=dev)
        assert K.gemm_qkv_rope(h, wqkv, B, T, heads, 128, q, kc, vc, cos, sin, 5, tile_cfg=t) is not None
        outs[t] = torch.cat([q.flatten(), kc.flatten(), vc.flatten()])
    e = rel(outs[34], outs[24])
    nz = (outs[34] != 0).float().mean().item()
    ok = e == 0.0 and nz > 0.1
    bad += not ok
    print(json.dumps({"case": f"qkv+rope B{B} T{T} heads{heads}", "vs_tile24": e, "nonzero": round(nz, 3), "ok": ok}), flush=True)
# ---- ragged N (generic fallback of the new epilogue), bias on the 16-bit park ----
a, w = R(700, 256), R(1000, 256)
bias = torch.randn(1000, device=dev, generator=g)
report("ragged N=1000 + bias + relu", K.gemm(a, w, bias=bias, act="relu", tile_cfg=34), K.gemm(a, w, bias=bias, act="relu", tile_cfg=24),
       (a.float() @ w.float().t() + bias).relu())
a, w = R(700, 256), R(1024, 256)
bias = torch.randn(1024, device=dev, generator=g)
for act in (None, "relu", "quick_gelu", "silu"):
    r32 = a.float() @ w.float().t() + bias
    r32 = {None: r32, "relu": r32.relu(), "quick_gelu": r32 * torch.sigmoid(1.702 * r32), "silu": torch.nn.functional.silu(r32)}[act]
    report(f"16-bit park bias+{act}", K.gemm(a, w, bias=bias, act=act, tile_cfg=34), K.gemm(a, w, bias=bias, act=act, tile_cfg=24), r32)
report("fp32 out + bias", K.gemm(a, w, bias=bias, out_dtype=torch.float32, tile_cfg=34), K.gemm(a, w, bias=bias, out_dtype=torch.float32, tile_cfg=24),
       a.float() @ w.float().t() + bias, tol=1e-5)

# ---- 3x3 convolution (implicit GEMM) ----
for (B, H, W, Ci, Co, G) in [(1, 48, 48, 128, 256, 1), (2, 24, 20, 64, 320, 1), (1, 14, 14, 128, 256, 4)]:
    x = R(G, B, H, W, Ci) if G > 1 else R(B, H, W, Ci)
    ws = [torch.randn(Co, Ci, 3, 3, device=dev, generator=g) * 0.05 for _ in range(G)]
    wk = K.prep_conv3x3_weight(ws)
    xs = x if G > 1 else x[None]
    ref = sum(torch.nn.functional.conv2d(xs[i].float().permute(0, 3, 1, 2), ws[i].to(torch.bfloat16).float(), padding=1)
              for i in range(G)).permute(0, 2, 3, 1)
    report(f"conv {B}x{H}x{W}x{Ci}->{Co} g{G}", K.conv3x3(x, wk, groups=G, tile_cfg=34), K.conv3x3(x, wk, groups=G, tile_cfg=24), ref,
           exact=False)
sizes = [(16, 16), (8, 8), (4, 4), (2, 2)]
mm = K.MlvlMaps(2, sizes, 128, dev)
mm.flat.copy_(R(*mm.flat.shape))
wc = torch.randn(256, 128, 3, 3, device=dev, generator=g) * 0.05
wk = K.prep_conv3x3_weight(wc)
lib().g4r_gemm_debug_mode(60)          # the ring ping-pong kernel (A/B arm)
ref = K.conv3x3_mlvl(mm, wk).flat.clone()
lib().g4r_gemm_debug_mode(0)
got = K.conv3x3_mlvl(mm, wk).flat.clone()
r32 = torch.cat([torch.nn.functional.conv2d(mm.levels[l].float().permute(0, 3, 1, 2), wc.to(torch.bfloat16).float(), padding=1)
                 .permute(0, 2, 3, 1).reshape(-1, 256) for l in range(4)])
report("conv all levels in one launch", got, ref, r32, exact=False)
print("ALL OK" if bad == 0 else f"{bad} FAILED", flush=True)
if "--no-time" in sys.argv or bad:
    sys.exit(1 if bad else 0)


# ---- timings ----
def burst(fn, n=8, rounds=5):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    return sorted(ts)[len(ts) // 2]


for (M, N, Kd) in [(4096, 4096, 4096), (8192, 8192, 4096), (12272, 12288, 4096), (12272, 4096, 4096), (12272, 4096, 11008),
                   (12272, 22016, 4096), (9232, 3072, 1024), (9232, 4096, 1024), (767, 12288, 4096), (3068, 12288, 4096)]:
    a, w = R(M, Kd), R(N, Kd)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    row = {"shape": [M, N, Kd]}
    for t in (24, 26, 34):
        fn = lambda: K.gemm(a, w, out=out, tile_cfg=t)          # noqa: E731
        fn(); torch.cuda.synchronize()
        us = burst(fn)
        row[f"tile{t}"] = [round(us, 1), round(2.0 * M * N * Kd / us / 1e6, 1)]
    wt = w.t()
    fn = lambda: torch.mm(a, wt, out=out)                       # noqa: E731
    fn(); torch.cuda.synchronize()
    us = burst(fn)
    row["vendor"] = [round(us, 1), round(2.0 * M * N * Kd / us / 1e6, 1)]
    print(json.dumps(row), flush=True)
# the fuse-round conv (all levels, batch 1 and 4)
for B in (1, 4):
    mm = K.MlvlMaps(B, [(192, 192), (96, 96), (48, 48), (24, 24)], 1024, dev)
    mm.flat.copy_(R(*mm.flat.shape))
    wk = R(1024, 9 * 1024)
    out = K.MlvlMaps(B, mm.sizes, 1024, dev)
    row = {"conv_mlvl_batch": B}
    for mode in (60, 0):
        lib().g4r_gemm_debug_mode(mode)
        fn = lambda: K.conv3x3_mlvl(mm, wk, out=out)            # noqa: E731
        fn(); torch.cuda.synchronize()
        us = burst(fn, n=4)
        row["pp32" if mode == 60 else "w4k64"] = [round(us, 1), round(2.0 * mm.flat.size(0) * 1024 * 9216 / us / 1e6, 1)]
    lib().g4r_gemm_debug_mode(0)
    print(json.dumps(row), flush=True)
