#!/usr/bin/env python3
"""tools/order_ab.py -- workgroup -> tile order A/B of the production 256 x 256 kernel (debug modes of g4r_gemm_debug_mode):
dense: 30 plain order, 32 / 31 / 33 = groups of 4 / 8 / 16 row tiles (default: 8 when >= 12 row tiles);
fuse-round convolution (all levels, batch 4 and 16): 0 default (N fastest: an XCD's wave = 8 pixel tiles x 4 weight panels),
41 = 16 x 2, 42 = 32 x 1.  Burst timings; run under rocprofv3 --pmc FETCH_SIZE for the traffic."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402
from gpt4roi_amd._lib import lib  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
R = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)    # noqa: E731


def burst(fn, n=6, rounds=5):
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


for (M, N, Kd) in [(12272, 12288, 4096), (12272, 22016, 4096), (12272, 4096, 11008), (12272, 4096, 4096)]:
    a, w = R(M, Kd), R(N, Kd)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    row = {"dense": [M, N, Kd]}
    for mode, name in ((0, "default(8)"), (30, "plain"), (32, "group4"), (33, "group16")):
        lib().g4r_gemm_debug_mode(mode)
        fn = lambda: K.gemm(a, w, out=out, tile_cfg=34)          # noqa: E731
        fn(); torch.cuda.synchronize()
        us = burst(fn)
        row[name] = [round(us, 1), round(2.0 * M * N * Kd / us / 1e6)]
    lib().g4r_gemm_debug_mode(0)
    print(json.dumps(row), flush=True)
for B in (4, 16):
    mm = K.MlvlMaps(B, [(192, 192), (96, 96), (48, 48), (24, 24)], 1024, dev)
    mm.flat.copy_(R(*mm.flat.shape))
    wk = R(1024, 9 * 1024)
    out = K.MlvlMaps(B, mm.sizes, 1024, dev)
    row = {"conv_mlvl_batch": B}
    for mode, name in ((0, "8x4"), (41, "16x2"), (42, "32x1")):
        lib().g4r_gemm_debug_mode(mode)
        fn = lambda: K.conv3x3_mlvl(mm, wk, out=out)            # noqa: E731
        fn(); torch.cuda.synchronize()
        us = burst(fn, n=3)
        row[name] = [round(us, 1), round(2.0 * mm.flat.size(0) * 1024 * 9216 / us / 1e6)]
    lib().g4r_gemm_debug_mode(0)
    print(json.dumps(row), flush=True)
