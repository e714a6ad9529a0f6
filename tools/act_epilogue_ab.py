#!/usr/bin/env python3
"""tools/act_epilogue_ab.py (profiles/r05_act_epilogue.txt) -- the gated-activation epilogues of the production tile (QuickGELU of the ViT's fc1, SwiGLU of
LLaMA's gate|up) against the same GEMM without an activation: what the activation costs on top of the K loop.
Burst timing (6 launches per event pair, median of 5)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402

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


for name, (M, N, Kd), act in (("vit_fc1", (9232, 4096, 1024), "quick_gelu"), ("vit_fc1_batch8", (4616, 4096, 1024), "quick_gelu"),
                              ("llama_gate_up", (12272, 22016, 4096), "swiglu")):
    a, w = R(M, Kd), R(N, Kd) * 0.05
    bias = torch.randn(N, device=dev) * 0.1 if act != "swiglu" else None
    row = {"gemm": name, "shape": [M, N, Kd]}
    for label, ac in (("plain", None), (act, act)):
        fn = lambda: K.gemm(a, w, bias=bias, act=ac, tile_cfg=K.BIG_TILE)             # noqa: E731
        fn(); torch.cuda.synchronize()
        us = burst(fn)
        row[label] = [round(us, 1), round(2.0 * M * N * Kd / us / 1e6)]
    print(json.dumps(row), flush=True)
