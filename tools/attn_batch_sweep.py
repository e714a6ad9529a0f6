#!/usr/bin/env python3
"""tools/attn_batch_sweep.py -- round 6: the instantiated (NWG, NG) variants of the prefill attention forward at the shapes of the
MERGED step (16 requests: LLaMA 16 x 767 x 32 heads x 128 causal; CLIP 16 x 577 x 16 heads x 64) and of one request, by hipGraph
replay.  The production choice (variant 142: 4 waves x 2 key groups) was tuned on one request."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K
from gpt4roi_amd import _lib
from attn_v2_check import timeit

lib = _lib.lib()
dev = "cuda"
R = lambda *s: (torch.randn(*s, device=dev) * 0.7).to(torch.bfloat16)
VARS = {128: [0, 1, 42, 142, 41], 64: [0, 1, 24, 124, 42, 142, 41]}
for (B, H, D, T, causal, what) in [(16, 32, 128, 767, True, "LLaMA merged 16"), (1, 32, 128, 767, True, "LLaMA one request"),
                                   (8, 32, 128, 699, True, "LLaMA training batch"), (16, 16, 64, 577, False, "CLIP batch 16"),
                                   (1, 16, 64, 577, False, "CLIP batch 1")]:
    q, k, v = R(B, T, H * D), R(B, T, H * D), R(B, T, H * D)
    o = torch.empty_like(q)
    flops = 4.0 * B * H * T * T * D * (0.5 if causal else 1.0)
    row = []
    for var in VARS[D]:
        lib.g4r_attn_debug_variant(var)
        us = timeit(lambda: K.flash_attn(q, k, v, H, 1 / math.sqrt(D), causal, out=o))
        row.append(f"{var}: {us:.1f} us ({flops / us / 1e6:.0f} TF/s)")
    lib.g4r_attn_debug_variant(0)
    print(f"{what:22s} B{B} H{H} D{D} T{T}: " + "  ".join(row), flush=True)
