#!/usr/bin/env python3
"""tools/vit_gemm_sweep.py -- the four GEMMs of a CLIP ViT-L/14 block (qkv + bias, o + bias + residual, fc1 + bias + QuickGELU,
fc2 + bias + residual) at batch 1 / 8 / 16 (M = 577 B) over the candidate tiles, WITH their production epilogues, burst of 16
launches, median of 5 -- what kernels.pick_tile's K = 1024 rules are chosen from.  Prints one JSON line per (shape, tile)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(3)
R = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.5).to(torch.bfloat16)    # noqa: E731


def burst(fn, n=16, rounds=5):
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


for B in (1, 8, 16):
    M = 577 * B
    for (N, Kd, what, act, res) in [(3072, 1024, "qkv", None, False), (1024, 1024, "o", None, True), (4096, 1024, "fc1", "quick_gelu", False),
                                    (1024, 4096, "fc2", None, True)]:
        a, w = R(M, Kd), R(N, Kd)
        bias = torch.randn(N, device=dev, generator=g)
        resid = R(M, N) if res else None
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        row = {"batch": B, "gemm": what, "shape": [M, N, Kd], "default_tile": K.pick_tile(M, N, Kd)}
        cands = [("default", None, 1), ("0", 0, 1), ("7", 7, 1), ("13", 13, 1), ("14", 14, 1), ("28", 28, 1), ("34", 34, 1), ("24", 24, 1)]
        if Kd >= 4096:
            cands += [("34x2", 34, 2), ("28x2", 28, 2), ("14x2", 14, 2), ("0x2", 0, 2)]
        for name, t, sp in cands:
            try:
                fn = lambda: K.gemm(a, w, bias=bias, residual=resid, act=act, out=out, tile_cfg=t, splits=sp)     # noqa: E731
                fn()
                torch.cuda.synchronize()
                us = burst(fn)
                row[name] = [round(us, 1), round(2.0 * M * N * Kd / us / 1e6)]
            except Exception as ex:                                      # a tile that does not serve the shape
                row[name] = str(ex)[:40]
        print(json.dumps(row), flush=True)
