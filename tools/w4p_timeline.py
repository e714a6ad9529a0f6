#!/usr/bin/env python3
"""tools/w4p_timeline.py -- per-tile s_memtime stamps of the persistent GEMM (gemm_bf16_w4k64p_kernel), probe build only:

    G4R_EXTRA_HIPCC_FLAGS=-DG4R_W4P_PROBE python -m gpt4roi_amd.build --force && python tools/w4p_timeline.py

Per workgroup and tile: [tile top, K loop begin, K loop end, epilogue end] -> cycles spent waiting for the first K tile,
in the K loop (per 64 of K against the MFMA floor of 2048), in the epilogue (+ the next tile's setup / pieces when they come
first), per epilogue mode."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402

DEV = "cuda"


def med(x):
    x = sorted(x)
    return x[len(x) // 2]


def run(name, fn, ws, nk):
    ws.zero_()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ws.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    st = ws.view(torch.int64)[:256 * 64].view(256, 16, 4).cpu()
    wait, loop, epi, tiles = [], [], [], []
    for b in range(256):
        for t in range(16):
            s = st[b, t]
            if int(s[3]) == 0:
                break
            wait.append(int(s[1] - s[0])); loop.append(int(s[2] - s[1])); epi.append(int(s[3] - s[2]))
        tiles.append(t)
    life = [int(st[b, tiles[b] - 1, 3] - st[b, 0, 0]) for b in range(256) if tiles[b] > 0]
    row = {"case": name, "launch_us": round(e0.elapsed_time(e1) * 1e3, 1), "tiles_per_wg": med(tiles), "wait_first_k_tile": med(wait),
           "k_loop": med(loop), "k_loop_per_64": round(med(loop) / nk, 1), "k_loop_max": max(loop), "epilogue_plus_next_setup": med(epi),
           "epilogue_max": max(epi), "wg_lifetime": med(life), "loop_share": round(sum(loop) / max(1, sum(life)), 3),
           "clock_GHz_if_launch_is_lifetime": round(med(life) / (e0.elapsed_time(e1) * 1e3) / 1e3, 3)}
    print(json.dumps(row), flush=True)


def main():
    dt = torch.bfloat16
    g = torch.Generator(device=DEV).manual_seed(1)
    mk = lambda *s, sc=0.5: (torch.randn(*s, device=DEV, generator=g) * sc).to(dt)      # noqa: E731
    ws = torch.zeros(1 << 20, dtype=torch.float32, device=DEV)
    M = 12272
    x = mk(M, 4096)
    w = mk(12288, 4096, sc=0.02)
    out = torch.empty(M, 12288, dtype=dt, device=DEV)
    run("plain 12272x12288x4096 (P16, pieces first)", lambda: K.gemm(x, w, out=out, workspace=ws, tile_cfg=34), ws, 64)
    wo = mk(4096, 4096, sc=0.02)
    res = mk(M, 4096)
    out_o = torch.empty(M, 4096, dtype=dt, device=DEV)
    run("o_proj + residual 12272x4096x4096 (WIDE, epilogue first)", lambda: K.gemm(x, wo, residual=res, out=out_o, workspace=ws, tile_cfg=34), ws, 64)
    wgu = mk(21760, 4096, sc=0.02)
    out_gu = torch.empty(M, 10880, dtype=dt, device=DEV)
    run("gate|up + SwiGLU 12272x21760x4096 (pieces first)", lambda: K.gemm(x, wgu, act="swiglu", out=out_gu, workspace=ws, tile_cfg=34), ws, 64)
    f = mk(M, 11008)
    wd = mk(4096, 11008, sc=0.02)
    run("down_proj + residual 12272x4096x11008", lambda: K.gemm(f, wd, residual=res, out=out_o, workspace=ws, tile_cfg=34), ws, 172)


if __name__ == "__main__":
    main()
