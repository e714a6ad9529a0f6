"""Loop cycles (s_memtime stamps of tile 25 = tile 24 + stamps) against the event-timed launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K
dev = "cuda"
for (M, N, Kd) in [(4096, 4096, 4096), (767, 21760, 4096), (767, 12288, 4096)]:
    a = (torch.rand(M, Kd, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(N, Kd, device=dev) * 2 - 1) / 37).bfloat16()
    ws = torch.zeros(64, dtype=torch.float32, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for tile in (25, 24):
        for _ in range(3):
            K.gemm(a, w, tile_cfg=tile, workspace=ws, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            K.gemm(a, w, tile_cfg=tile, workspace=ws, out=out)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        st = ws.view(torch.int64).cpu().tolist()
        msg = f"{M}x{N}x{Kd} tile {tile}: {us:.1f} us/launch back-to-back"
        if tile == 25:
            for g in (0, 1):
                s = st[g * 8:(g + 1) * 8]
                msg += f" | group{g}: loop {s[7]-s[6]} ticks ({(s[7]-s[6])/(Kd//32):.0f}/tile), tile16 {s[5]-s[0]}"
        print(msg)
