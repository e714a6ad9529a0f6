"""tools/merge_sweep.py -- region-tokens/s of the configs[1] launch sequence for (requests merged per sequence) x (sequences in
flight), ONE model, ONE box, one process (box-to-box variance is +-5 %, so the default of bench.py is chosen from this).
    python tools/merge_sweep.py [--combos 1x1,1x2,2x2,4x1,4x2,4x3,6x2,8x1,8x2] [--steps 6]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--combos", default="1x1,1x2,2x2,4x1,4x2,4x3,6x2,8x1,8x2")
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
args = bench.parse.__wrapped__() if hasattr(bench.parse, "__wrapped__") else None
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = "cuda:0"
torch.cuda.set_device(0)
torch.set_grad_enabled(False)
model, ids = bench.build_model(args, dev, seed=100, dtype=torch.bfloat16 if a.dtype == "bf16" else torch.float16)
out = {}
for c in a.combos.split(","):
    B, S = (int(x) for x in c.split("x"))
    img, boxes, prompt = bench.make_inputs(args, ids, dev, seed=1, batch=B)
    t = bench.timed_replay(model, img, boxes, prompt, steps=max(a.steps, 2 * S), warmup=2, streams=S)
    out[c] = round(args.rois * B / t, 1)
    print(f"{B} merged x {S} in flight: {1e3 * t:8.2f} ms per sequence, {out[c]:8.1f} region-tokens/s, {1e3 * t / B:6.2f} ms per request", flush=True)
    del img, boxes, prompt
    torch.cuda.empty_cache()
print(json.dumps(out))
