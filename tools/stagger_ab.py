#!/usr/bin/env python3
"""tools/stagger_ab.py -- start de-phasing of the one-wave-per-SIMD GEMM (debug modes 70 + n: the first 256 workgroups start
phase x n x 0.25 us late, phase = CU slot mod 8; 79 = off): burst timings on the merged LLaMA shapes and the fuse-round conv."""
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


MODES = (79, 72, 74, 76, 78, 79, 73, 75, 77)
for (M, N, Kd) in [(12272, 12288, 4096), (12272, 22016, 4096), (12272, 4096, 11008), (12272, 4096, 4096), (4096, 4096, 4096)]:
    a, w = R(M, Kd), R(N, Kd)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    row = {"dense": [M, N, Kd]}
    for mode in MODES:
        lib().g4r_gemm_debug_mode(mode)
        fn = lambda: K.gemm(a, w, out=out, tile_cfg=34)          # noqa: E731
        fn(); torch.cuda.synchronize()
        us = burst(fn)
        row.setdefault("off" if mode == 79 else f"{(mode - 70) * 0.25:.2f}us", []).append(round(2.0 * M * N * Kd / us / 1e6))
    lib().g4r_gemm_debug_mode(0)
    print(json.dumps(row), flush=True)
mm = K.MlvlMaps(4, [(192, 192), (96, 96), (48, 48), (24, 24)], 1024, dev)
mm.flat.copy_(R(*mm.flat.shape))
wk = R(1024, 9 * 1024)
out = K.MlvlMaps(4, mm.sizes, 1024, dev)
row = {"conv_mlvl_batch": 4}
for mode in MODES:
    lib().g4r_gemm_debug_mode(mode)
    fn = lambda: K.conv3x3_mlvl(mm, wk, out=out)            # noqa: E731
    fn(); torch.cuda.synchronize()
    us = burst(fn, n=3)
    row.setdefault("off" if mode == 79 else f"{(mode - 70) * 0.25:.2f}us", []).append(round(2.0 * mm.flat.size(0) * 1024 * 9216 / us / 1e6))
lib().g4r_gemm_debug_mode(0)
print(json.dumps(row), flush=True)
