"""Phase timeline of the ping-pong GEMM (tile 23 = tile 22 + s_memtime stamps).  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K

dev = "cuda"
for (M, N, Kd) in [(4096, 4096, 4096), (768, 12288, 4096)]:
    a = torch.randn(M, Kd, device=dev).bfloat16()
    w = torch.randn(N, Kd, device=dev).bfloat16()
    ws = torch.zeros(64, dtype=torch.float32, device=dev)
    for _ in range(3):
        K.gemm(a, w, tile_cfg=23, workspace=ws)
    torch.cuda.synchronize()
    st = ws.view(torch.int64).cpu().tolist()
    for g in (0, 1):
        s = st[g * 8:(g + 1) * 8]
        print(f"{M}x{N}x{Kd} group{g}: read0+issue {s[5]-s[0]}  bar {s[1]-s[5]}  mma0+bar {s[2]-s[1]}  read1+bar {s[3]-s[2]} "
              f" mma1+bar {s[4]-s[3]}  | K tile total {s[4]-s[0]}  loop avg {(s[7]-s[6])/(Kd//64):.0f} ticks (100 MHz?)")
