#!/usr/bin/env python3
"""tools/wg_timeline.py [--tiles 25,35] [--shapes MxNxK,...] -- per-workgroup timeline of a probed GEMM tile (25 = ring
ping-pong + stamps, 35 = one-wave-per-SIMD K 64 + stamps): entry / prologue / K loop / epilogue of EVERY workgroup in
s_memtime ticks, cycles per 64 of K inside the loop, the clock (ticks per 100 MHz wall tick), and the launch's wall time."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tiles", default="25,35")
ap.add_argument("--shapes", default="4096x4096x4096,12272x12288x4096,12272x4096x11008")
a = ap.parse_args()
dev = "cuda"
for shp in a.shapes.split(","):
    M, N, Kd = (int(v) for v in shp.split("x"))
    g = torch.Generator(device=dev).manual_seed(1)
    x = (torch.randn(M, Kd, device=dev, generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, Kd, device=dev, generator=g) * 0.5).bfloat16()
    nwg = -(-M // 256) * -(-N // 256)
    for tile in (int(t) for t in a.tiles.split(",")):
        ws = torch.zeros(2 * (16 + 8 * nwg), dtype=torch.float32, device=dev)
        for _ in range(3):
            K.gemm(x, w, tile_cfg=tile, workspace=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        K.gemm(x, w, tile_cfg=tile, workspace=ws)
        e1.record()
        torch.cuda.synchronize()
        st = ws.view(torch.int64)[16:16 + 8 * nwg].view(nwg, 8).cpu().numpy()
        t0 = st[:, 0].min()
        pro, loop, epi, life = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 3] - st[:, 0]
        wall = (st[:, 6] - st[:, 5]).astype(np.float64)                 # 100 MHz
        ghz = life / np.maximum(wall, 1) * 0.1
        q = lambda v: f"min {int(v.min()):8d} med {int(np.median(v)):8d} max {int(v.max()):8d}"      # noqa: E731
        print(f"== tile {tile} {M}x{N}x{Kd}: {nwg} workgroups ({nwg / 256:.2f} waves of 256), launch {e0.elapsed_time(e1) * 1e3:.1f} us")
        if st[:, 7].any():
            print(f"   epilogue: park (accumulators -> LDS) {q(st[:, 7] - st[:, 2])}, read-back + stores {q(st[:, 3] - st[:, 7])}")
        print(f"   prologue  {q(pro)}\n   K loop    {q(loop)}   = {np.median(loop) / (Kd / 64):.0f} ticks per 64 of K (MFMA floor 2048)\n"
              f"   epilogue  {q(epi)}\n   lifetime  {q(life)}   loop share {np.median(loop) / np.median(life):.3f}\n"
              f"   s_memtime ticks per ns (clock): med {np.median(ghz):.3f} min {ghz.min():.3f} max {ghz.max():.3f}; "
              f"last exit {(st[:, 3].max() - t0) / np.median(ghz) / 1e3:.1f} us after first entry", flush=True)
