#!/usr/bin/env python3
"""tools/microbench.py -- per-kernel timings on the GPU box (not part of bench.py's contract)."""
import json, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K

dev = "cuda"
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

res = []
def rec(name, secs, flops=None, bytes_=None):
    r = {"name": name, "us": round(secs * 1e6, 1)}
    if flops: r["TF/s"] = round(flops / secs / 1e12, 1)
    if bytes_: r["GB/s"] = round(bytes_ / secs / 1e9, 1)
    res.append(r); print(json.dumps(r), flush=True)

torch.manual_seed(0)
def R(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)

for (M, N, Kd) in [(800, 4096, 4096), (800, 12288, 4096), (800, 22016, 4096), (800, 4096, 11008), (800, 32064, 4096),
                   (4096, 4096, 4096), (577, 3072, 1024), (577, 4096, 1024), (577, 1024, 4096), (48960, 1024, 1088)]:
    a, w = R(M, Kd), R(N, Kd)
    for tile in (0, 1, 4, 5, 6, 7, 8):
        try:
            t = timeit(lambda: K.gemm(a, w, tile_cfg=tile))
            rec(f"gemm {M}x{N}x{Kd} tile{tile}", t, 2.0 * M * N * Kd)
        except Exception as ex:
            print("ERR", M, N, Kd, tile, ex)
    t = timeit(lambda: a @ w.t())
    rec(f"torch(hipblaslt) {M}x{N}x{Kd}", t, 2.0 * M * N * Kd)
    del a, w

# conv3x3 at the fuse-round shapes (P=24): 192,96,48,24
for Hs in (192, 96, 48, 24):
    x = R(1, Hs, Hs, 1024); w = R(1024, 9 * 1024)
    for tile in (0, 1, 5, 6, 7, 8):
        t = timeit(lambda: K.conv3x3(x, w, tile_cfg=tile), iters=5)
        rec(f"conv3x3 {Hs}x{Hs}x1024 tile{tile}", t, 2.0 * Hs * Hs * 1024 * 9216)
# pconv: 32 rois, 4 levels
x = R(4, 32, 14, 14, 1024); w = R(1024, 4 * 9 * 1024)
for tile in (0, 1, 4, 5, 6, 7, 8):
    t = timeit(lambda: K.conv3x3(x, w, groups=4, tile_cfg=tile), iters=5)
    rec(f"pconv 32 rois tile{tile}", t, 2.0 * 32 * 196 * 1024 * 4 * 9216)
# flatten_linear
a, w = R(32, 200704), R(1024, 200704)
for sp in (16, 32, 49, 64):
    t = timeit(lambda: K.gemm(a, w, splits=sp, tile_cfg=4, out_dtype=torch.float32))
    rec(f"flatten_linear splits{sp}", t, 2.0 * 32 * 1024 * 200704, 2.0 * 1024 * 200704)
del a, w
# attention
for (B, H, D, T, causal) in [(1, 16, 64, 577, False), (1, 32, 128, 800, True), (1, 32, 128, 2048, True)]:
    q, k, v = R(B, T, H * D), R(B, T, H * D), R(B, T, H * D)
    t = timeit(lambda: K.flash_attn(q, k, v, H, 1 / math.sqrt(D), causal))
    rec(f"attn B{B} H{H} D{D} T{T} causal{int(causal)}", t, 4.0 * B * H * T * T * D / (2 if causal else 1))
# roi_align multi-level, P=24, 32 rois
P = 24
feats = [R(1, s, s, 1024) for s in (8 * P, 4 * P, 2 * P, P)]
g = torch.Generator().manual_seed(0)
xy = torch.rand(32, 2, generator=g) * 0.6; wh = torch.rand(32, 2, generator=g) * 0.3 + 0.05
rois = torch.cat([torch.zeros(32, 1), xy * 14 * P, (xy + wh) * 14 * P], 1).to(dev)
t = timeit(lambda: K.roi_align_mlvl(feats, rois, 14, [8 / 14, 4 / 14, 2 / 14, 1 / 14]))
alg = sum(f.numel() for f in feats) * 2 + 4 * 32 * 196 * 1024 * 2
rec("roi_align_mlvl P24 32rois bf16", t, None, alg)
# elementwise
x = R(1, 577, 1024)
t = timeit(lambda: K.upsample_coord(x[:, 1:], 24, 24, 192, 192, 1088)); rec("upsample_coord 192", t, None, 192 * 192 * 1088 * 2)
own, top = R(1, 192, 192, 1024), R(1, 96, 96, 1024)
t = timeit(lambda: K.fuse_shuffle(own, top, own)); rec("fuse_shuffle 192", t, None, 192 * 192 * 1024 * 2 * 2)
h = R(767, 4096); gam = torch.ones(4096, device=dev)
t = timeit(lambda: K.rmsnorm(h, gam)); rec("rmsnorm 767x4096", t, None, 767 * 4096 * 4)
h1 = R(577, 1024); g1 = torch.ones(1024, device=dev)
t = timeit(lambda: K.layernorm(h1, g1, g1)); rec("layernorm 577x1024", t, None, 577 * 1024 * 4)
gm, bt = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
t = timeit(lambda: K.groupnorm_affine(own, gm, bt, 64)); rec("gn_stats 192", t, None, 192 * 192 * 1024 * 2)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/microbench.json", "w"), indent=1)
