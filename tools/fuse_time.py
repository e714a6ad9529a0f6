"""One fuse round's glue kernels at the 336^2 pyramid (4 levels 192/96/48/24, C = 1024, batch 1), hipGraph-timed:
merged shuffle, merged GroupNorm statistics, multi-level RoIAlign."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K, synthetic as syn
dev = "cuda"
B, C, sizes = 1, 1024, [(192, 192), (96, 96), (48, 48), (24, 24)]
g = torch.Generator().manual_seed(0)
z = K.MlvlMaps(B, sizes, C, dev)
z.flat.copy_(torch.randn(z.flat.shape, generator=g).to(torch.bfloat16))
gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
beta = (0.1 * torch.randn(C, generator=g)).to(dev)
lvl_list = [(l, min(l + 1, 3), max(l - 1, 0)) for l in range(4)]
out = K.MlvlMaps(B, sizes, C, dev)
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=torch.cuda.Stream()):
        for _ in range(iters): fn()
    gr.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
affs = K.groupnorm_affine_mlvl(z, gamma, beta, 64, 1e-5)
mb = z.flat.numel() * 2 / 1e6
t = timeit(lambda: K.groupnorm_affine_mlvl(z, gamma, beta, 64, 1e-5))
print(f"groupnorm_affine_mlvl (stats + finalize): {t:.1f} us  ({mb / t * 1e3:.0f} GB/s of map reads)")
t = timeit(lambda: [K.groupnorm_affine(m, gamma, beta, 64, 1e-5) for m in z.levels])
print(f"  per-level launches (4 x 2): {t:.1f} us")
t = timeit(lambda: K.fuse_shuffle_mlvl(z.levels, affs, lvl_list, out))
print(f"fuse_shuffle_mlvl: {t:.1f} us  ({2 * mb / t * 1e3:.0f} GB/s read+write)")
t = timeit(lambda: [K.fuse_shuffle(z.levels[a], z.levels[b], z.levels[c], affs[a], affs[b], affs[c], out=out.levels[a]) for a, b, c in lvl_list])
print(f"  per-level launches (4): {t:.1f} us")
rois = torch.cat([torch.zeros(32, 1), syn.boxes(32, g) * 336.0], 1).to(dev)
t = timeit(lambda: K.roi_align_mlvl(z.levels, rois, 14, [1 / 1.75, 1 / 3.5, 1 / 7.0, 1 / 14.0], affines=affs))
print(f"roi_align_mlvl (bf16, deferred GN): {t:.1f} us = {151.65 / t * 1e3:.0f} GB/s algorithmic")
