#!/usr/bin/env python3
"""tools/conv_time.py -- the one-launch-per-round 3x3 convolution over all pyramid levels (production dispatch) at the bench geometry
(P = 24, C = 1024, batch 16), sustained for --seconds: us per launch and TF/s.  With G4R_LIB=<a library built under G4R_BUILD_TAG> the
A/B arm of a compile-time switch (e.g. -DG4R_W4_NT_STORES=1)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
P, C = 24, 1024
mm = K.MlvlMaps(a.batch, [(8 * P, 8 * P), (4 * P, 4 * P), (2 * P, 2 * P), (P, P)], C, "cuda")
g = torch.Generator(device="cuda").manual_seed(1)
mm.flat.copy_((torch.randn(*mm.flat.shape, generator=g, device="cuda") * 0.5).to(torch.bfloat16))
wk = K.prep_conv3x3_weight((torch.randn(C, C, 3, 3, generator=g, device="cuda") * 0.02))
rows = mm.flat.shape[0]
for _ in range(3):
    out = K.conv3x3_mlvl(mm, wk, act="relu")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n, t0 = 0, time.perf_counter()
e0.record()
while time.perf_counter() - t0 < a.seconds:
    for _ in range(4):
        K.conv3x3_mlvl(mm, wk, act="relu")
    n += 4
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
print(f"conv3x3 all levels batch {a.batch}: rows {rows}, {us:.1f} us per launch, {2.0 * rows * C * 9 * C / us / 1e6:.1f} TF/s  (lib {os.environ.get('G4R_LIB', 'shipped')})")
