"""Phase timeline of the ring ping-pong GEMM (tile 25 = tile 24 + s_memtime stamps).  Run on the GPU box."""
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
        K.gemm(a, w, tile_cfg=int(os.environ.get("PP_TILE", "25")), workspace=ws)
    torch.cuda.synchronize()
    st = ws.view(torch.int64).cpu().tolist()
    for g in (0, 1):
        s = st[g * 8:(g + 1) * 8]
        print(f"{M}x{N}x{Kd} group{g}: ds_read issue {s[1]-s[0]}  DMA issue {s[2]-s[1]}  vmcnt wait {s[3]-s[2]}  "
              f"lgkm+barrier {s[4]-s[3]}  mma+barrier {s[5]-s[4]}  | K32 tile {s[5]-s[0]}  loop avg {(s[7]-s[6])/(Kd//32):.0f}")
