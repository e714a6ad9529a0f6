"""Per-workgroup timeline of the ring ping-pong GEMM (tile 25 = tile 24 + s_memtime stamps): entry / prologue / K loop /
epilogue of EVERY workgroup, to see where a launch's wall time goes beyond one workgroup's K loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gpt4roi_amd import kernels as K
dev = "cuda"
for (M, N, Kd, act) in [(4096, 4096, 4096, None), (767, 21760, 4096, None), (767, 12288, 4096, None), (8192, 8192, 4096, None)]:
    a = (torch.rand(M, Kd, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(N, Kd, device=dev) * 2 - 1) / 37).bfloat16()
    nwg = -(-M // 256) * -(-N // 256)
    ws = torch.zeros(2 * (16 + 8 * nwg), dtype=torch.float32, device=dev)
    for _ in range(3):
        K.gemm(a, w, tile_cfg=25, workspace=ws, act=act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.gemm(a, w, tile_cfg=25, workspace=ws, act=act); e1.record(); torch.cuda.synchronize()
    st = ws.view(torch.int64)[16:16 + 8 * nwg].view(nwg, 8).cpu().numpy()
    t0 = st[:, 0].min()
    entry, pro, loop, epi, end = st[:, 0] - t0, st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 3] - t0
    q = lambda x: f"min {x.min():7d} med {int(np.median(x)):7d} max {x.max():7d}"
    print(f"== {M}x{N}x{Kd} act={act}: {nwg} workgroups, event-timed launch {e0.elapsed_time(e1)*1e3:.1f} us; ticks at 2.4 GHz: "
          f"last exit {end.max()/2400:.1f} us")
    print(f"   entry after first  {q(entry)}\n   prologue           {q(pro)}\n   K loop             {q(loop)}\n"
          f"   epilogue           {q(epi)}\n   exit after first   {q(end)}")
    wall = (st[:, 6] - st[:, 5]).astype(np.float64)            # wall_clock64: constant 100 MHz
    clk = (st[:, 3] - st[:, 0]) / np.maximum(wall, 1) * 0.1   # s_memtime ticks per ns = GHz
    print(f"   workgroup lifetime {np.median(wall)*10/1e3:.1f} us (median, 100 MHz wall clock); s_memtime ticks / wall = "
          f"{np.median(clk):.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f})")
    for x in range(8):
        sel = st[:, 4] == x
        if sel.any() and x < 2:
            print(f"   XCC {x}: {int(sel.sum()):3d} wgs, loop mean {loop[sel].mean():9.0f}, entry mean {entry[sel].mean():8.0f}, exit max {end[sel].max():8d}")
