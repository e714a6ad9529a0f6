#!/usr/bin/env python3
"""tools/ab_one.py --impl hand:34|hand:24|vendor --shape M,N,K [--iters 20] -- one GEMM implementation, one shape, a fixed
number of launches: the target of rocprofv3 counter passes (A/B of the hand kernel and the vendor kernel, calibration only)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gpt4roi_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--impl", default="hand:34")
ap.add_argument("--shape", default="12272,12288,4096")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
M, N, Kd = (int(v) for v in a.shape.split(","))
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.randn(M, Kd, device=dev, generator=g) * 0.5).bfloat16()
w = (torch.randn(N, Kd, device=dev, generator=g) * 0.5).bfloat16()
y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
wt = w.t()
for impl in a.impl.split("+"):
    if impl.startswith("hand"):
        tile = int(impl.split(":")[1])
        fn = lambda: K.gemm(x, w, out=y, tile_cfg=tile)      # noqa: E731
    else:
        fn = lambda: torch.mm(x, wt, out=y)                  # noqa: E731
    for _ in range(a.iters):
        fn()
    torch.cuda.synchronize()
