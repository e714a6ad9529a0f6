"""tools/attn_probe.py -- A/B of the two head_dim-128 forward-attention instantiations (4 waves x 128 query rows vs
2 waves x 64) on the LLaMA prefill shapes; HIP-event timing, interleaved rounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt4roi_amd import _lib  # noqa: E402
from gpt4roi_amd import kernels as K  # noqa: E402

dev = "cuda"
lib = _lib.lib()
for (B, T) in ((1, 767), (2, 767), (8, 699), (1, 2048)):
    H, D = 32, 128
    q, k, v = (torch.randn(B, T, H * D, device=dev).to(torch.bfloat16) for _ in range(3))
    outs, times = {}, {1: [], 2: []}
    for rnd in range(12):
        for var in (1, 2):
            lib.g4r_attn_debug_variant(var)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o = K.flash_attn(q, k, v, H, D ** -0.5, True)
            e1.record()
            torch.cuda.synchronize()
            times[var].append(e0.elapsed_time(e1) * 1e3)
            outs[var] = o
    lib.g4r_attn_debug_variant(0)
    fl = 4.0 * B * H * T * T * D * 0.5
    med = {v_: sorted(t[2:])[len(t[2:]) // 2] for v_, t in times.items()}
    print(f"B={B} T={T}: 4-wave {med[1]:.1f} us ({fl / med[1] / 1e6:.0f} TF/s)  2-wave {med[2]:.1f} us ({fl / med[2] / 1e6:.0f} TF/s)  "
          f"max|diff| {float((outs[1].float() - outs[2].float()).abs().max()):.3e}")
