"""Per-node cost of a dependent kernel chain inside a captured hipGraph (what every launch boundary of the image path pays):
N launches of a one-workgroup kernel / of a chip-filling short kernel, replayed; us per node."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import kernels as K

dev = "cuda"
g1 = torch.ones(64, dtype=torch.float32, device=dev)
x1 = torch.randn(1, 64, device=dev).to(torch.bfloat16)
y1 = torch.empty_like(x1)
gN = torch.ones(4096, dtype=torch.float32, device=dev)
xN = torch.randn(767, 4096, device=dev).to(torch.bfloat16)
yN = torch.empty_like(xN)


def per_node(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=side):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


for name, fn in [("rmsnorm 1 x 64 (one workgroup)", lambda: K.rmsnorm(x1, g1, out=y1)),
                 ("rmsnorm 767 x 4096 (767 workgroups, 12.6 MB)", lambda: K.rmsnorm(xN, gN, out=yN))]:
    t1, t2 = per_node(fn, 50), per_node(fn, 400)
    print(f"{name}: 50 nodes {t1:.1f} us, 400 nodes {t2:.1f} us -> {(t2 - t1) / 350:.2f} us per node")
