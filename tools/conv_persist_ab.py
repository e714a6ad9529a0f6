#!/usr/bin/env python3
"""tools/conv_persist_ab.py -- round 6: the one-launch-per-round convolution (all pyramid levels, implicit GEMM) in the persistent
LEAN form (debug mode 63, TOOLS BUILD only: G4R_EXTRA_HIPCC_FLAGS=-DG4R_TOOLS_BUILD python -m gpt4roi_amd.build --force) against the
per-tile form (the production dispatch) and the ring kernel (mode 60): bit-identity (same K order as the per-tile form), fp32
reference on a small pyramid, and timing at the bench geometry (P = 24, C = 1024, batch 16 / 4 / 1).  Result (round 6): identical
results, 4-6 % slower -- not dispatched (profiles/r06_conv_persist_ab.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from gpt4roi_amd import kernels as K
from gpt4roi_amd._lib import lib
from vendor_ab import burst_time

DEV = "cuda"


def rnd(*s, scale=0.5, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*s, generator=g, device=DEV) * scale).to(dtype)


def conv(mm, wk, mode, act=None):
    lib().g4r_gemm_debug_mode(mode)
    try:
        return K.conv3x3_mlvl(mm, wk, act=act).flat.clone()
    finally:
        lib().g4r_gemm_debug_mode(0)


ok = True
# small pyramid, many images: 510 tiles, levels on tile boundaries; vs fp32 conv2d
mm = K.MlvlMaps(48, [(32, 32), (16, 16), (8, 8), (4, 4)], 64, DEV)
mm.flat.copy_(rnd(*mm.flat.shape, seed=1))
wc = rnd(512, 64, 3, 3, scale=0.05, seed=2, dtype=torch.float32)
wk = K.prep_conv3x3_weight(wc)
for act in (None, "relu"):
    a, b, c = conv(mm, wk, 63, act), conv(mm, wk, 0, act), conv(mm, wk, 60, act)
    ref = torch.cat([F.conv2d(m.float().permute(0, 3, 1, 2), wc.to(torch.bfloat16).float(), padding=1).permute(0, 2, 3, 1).reshape(-1, 512) for m in mm.levels])
    if act:
        ref = ref.relu()
    e = ((a.float() - ref).abs().max() / ref.abs().max()).item()
    print(f"small pyramid act={act}: persistent == per-tile: {torch.equal(a, b)}; vs ring kernel max diff {(a.float() - c.float()).abs().max().item():.3e}; vs fp32 {e:.3e}")
    ok &= torch.equal(a, b) and e < 6e-3
st = torch.cuda.current_stream()
for B in (16, 4, 1):
    mm = K.MlvlMaps(B, [(192, 192), (96, 96), (48, 48), (24, 24)], 1024, DEV)
    mm.flat.copy_(rnd(*mm.flat.shape, seed=3))
    wk = K.prep_conv3x3_weight(rnd(1024, 1024, 3, 3, scale=0.02, seed=4, dtype=torch.float32))
    a, b = conv(mm, wk, 63), conv(mm, wk, 0)
    same = torch.equal(a, b)
    ok &= same
    if not same:
        d = (a.float() - b.float()).abs()
        bad_rows = (d.max(1).values > 0).nonzero().flatten()
        print(f"   mismatching rows: {bad_rows.numel()} of {a.size(0)}; max diff {d.max().item():.3e}; first {bad_rows[:8].tolist()} last {bad_rows[-4:].tolist()}; "
              f"rows mod 256 of the first: {[int(r) % 256 for r in bad_rows[:8]]}; bad cols of first row: {(d[bad_rows[0]] > 0).nonzero().flatten()[:6].tolist()} n={(d[bad_rows[0]] > 0).sum().item()}")
        starts = [0]
        for lv in mm.levels:
            starts.append(starts[-1] + lv.numel() // 1024)
        for li in range(4):
            nb = ((bad_rows >= starts[li]) & (bad_rows < starts[li + 1])).sum().item()
            print(f"   level {li}: rows [{starts[li]}, {starts[li + 1]}): {nb} bad")
    t = {}
    for name, mode in (("persistent", 63), ("per_tile", 0)):
        lib().g4r_gemm_debug_mode(mode)
        for _ in range(3):
            K.conv3x3_mlvl(mm, wk)
        t[name] = sorted(burst_time(lambda: K.conv3x3_mlvl(mm, wk), 4, st) for _ in range(5))[2]
    lib().g4r_gemm_debug_mode(0)
    fl = 2.0 * mm.flat.size(0) * 1024 * 9216
    print(f"batch {B:2d}: rows {mm.flat.size(0)}, persistent == per-tile: {same}; persistent {t['persistent']:.1f} us ({fl / t['persistent'] / 1e6:.0f} TF/s), "
          f"per tile {t['per_tile']:.1f} us ({fl / t['per_tile'] / 1e6:.0f} TF/s), ratio {t['per_tile'] / t['persistent']:.3f}", flush=True)
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
