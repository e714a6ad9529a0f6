"""tools/ragged_merge.py -- requests of DIFFERENT prompt lengths merged into one launch sequence (right-padded to the longest,
a prepared llama.RaggedLayout as attention_mask), hipGraph replay, against the same number of equal-length requests.
    python tools/ragged_merge.py [--batch 16] [--steps 6]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
from gpt4roi_amd.llama import RaggedLayout     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=6)
a = ap.parse_args()
sys.argv = [sys.argv[0]]
args = bench.parse()
dev = "cuda:0"
torch.cuda.set_device(0)
torch.set_grad_enabled(False)
model, ids = bench.build_model(args, dev, seed=100, dtype=torch.bfloat16)
img, boxes, prompt = bench.make_inputs(args, ids, dev, seed=1, batch=a.batch)
B, T = prompt.shape
g = torch.Generator().manual_seed(3)
# a shorter request = fewer region references (4 tokens each) and no trailing question: the merged run keeps the padded ids
# (the masked tail is attended by nobody), the request alone is the truncated prompt with the boxes it still refers to
drop = [0] + [int(x) for x in torch.randint(0, 25, (B - 1,), generator=g)]
lens = [T if k == 0 else T - 20 - 4 * k for k in drop]
mask = torch.zeros(B, T, dtype=torch.bool)
for b, n in enumerate(lens):
    mask[b, :n] = True
layout = RaggedLayout.of(mask.to(dev), model.llama.max_positions)
t_eq = bench.timed_replay(model, img, boxes, prompt, steps=a.steps)
t_rg = bench.timed_replay(model, img, boxes, prompt, steps=a.steps, attention_mask=layout)
# correctness of the merged ragged sequence against each request alone (last kept position, all logits of request 1)
model.llama.reset(B)
lg = model(input_ids=prompt, images=img, bboxes=boxes, attention_mask=layout)
b = 1
alone = model(input_ids=prompt[b:b + 1, :lens[b]], images=img[b:b + 1], bboxes=[boxes[b][:args.rois - drop[b]]])
err = float((lg[b, :lens[b]] - alone[0]).abs().max() / alone[0].abs().max())
out = dict(batch=B, prompt_tokens=T, kept_tokens=lens, equal_ms=round(1e3 * t_eq, 2), ragged_ms=round(1e3 * t_rg, 2),
           equal_region_tokens_per_s=round(args.rois * B / t_eq, 1), ragged_region_tokens_per_s=round(args.rois * B / t_rg, 1),
           request1_logits_vs_alone_rel_err=round(err, 5))
print(json.dumps(out))
