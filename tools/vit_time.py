"""ViT tower alone, batch 1 (cold weights: 600 MB of them, replayed from a hipGraph), with / without the weight prefetch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt4roi_amd import synthetic as syn
from gpt4roi_amd.vit import ClipVisionTower
dev = "cuda"
v = syn.CLIP_L14
tower = ClipVisionTower(syn.vit_state(v["hidden"], v["inter"], v["layers"], 336, seed=0, device=dev, dtype=torch.bfloat16), heads=16, device=dev)
img = torch.randn(1, 3, 336, 336, device=dev)
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)     # 1 GiB: evicts L2 + Infinity Cache between replays
for pf in (False, True, False, True):
    tower.prefetch_weights = pf
    for _ in range(2): tower.forward(img)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=torch.cuda.Stream()):
        keep = tower.forward(img)
    ts = []
    for _ in range(8):
        flush.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"prefetch={pf}: ViT-L/14@336 batch 1 tower {ts[len(ts)//2]:.3f} ms (min {ts[0]:.3f})")
