#!/usr/bin/env python3
"""tools/m767_sweep.py -- the LLaMA-7B prefill GEMMs of ONE batch-1 request (M = 767) over candidate (tile, K slices), with the
production epilogues where they matter (o_proj / down_proj + residual; K slices include the reduce launch): what the M < 1024 rules
of kernels.pick_tile / long_k_plan are chosen from.  Burst of 16, median of 5."""
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


M = 767
for (N, Kd, what, res, cands) in [
        (12288, 4096, "q|k|v", False, [(28, 1), (34, 1), (24, 1)]),
        (4096, 4096, "o_proj", True, [(7, 1), (34, 1), (34, 2), (34, 4), (34, 5), (28, 2), (28, 4), (0, 1)]),
        (21760, 4096, "gate|up main (85 column tiles)", False, [(34, 1), (24, 1), (28, 1)]),
        (4096, 11008, "down_proj", True, [(28, 4), (34, 4), (34, 5), (34, 3), (28, 5), (7, 1)]),
        (21760, 4096, "lm_head main (fp32 out)", False, [(34, 1), (24, 1)]),
        (10246, 4096, "lm_head remainder (fp32 out)", False, [(None, 1), (34, 1), (34, 2), (28, 1)])]:
    a, w = R(M, Kd), R(N, Kd)
    resid = R(M, N) if res else None
    f32 = "fp32" in what
    out = torch.empty(M, N, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
    row = {"gemm": what, "shape": [M, N, Kd], "default": K.pick_tile(M, N, Kd)}
    for t, sp in cands:
        try:
            fn = lambda: K.gemm(a, w, residual=resid, out=out, tile_cfg=t, splits=sp)     # noqa: E731
            fn()
            torch.cuda.synchronize()
            us = burst(fn)
            row[f"{t}x{sp}"] = [round(us, 1), round(2.0 * M * N * Kd / us / 1e6)]
        except Exception as ex:
            row[f"{t}x{sp}"] = str(ex)[:50]
    print(json.dumps(row), flush=True)
