#!/usr/bin/env python3
"""SURVEY.md 8d config 5 shape on ONE GPU: 224x224 crop, 64 RoIs, prompt + N greedily generated tokens (mixed
prefill + decode).  Prints one JSON line: prefill ms, decode ms/token, end-to-end requests/s and tokens/s."""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch
torch.set_grad_enabled(False)   # inference tool: no autograd seam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gpt4roi_amd import synthetic as syn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rois", type=int, default=64)
ap.add_argument("--new-tokens", type=int, default=512)
ap.add_argument("--requests", type=int, default=3)
ap.add_argument("--batch", type=int, default=1, help="requests served together (SURVEY.md 8d config 5: 64 RoIs x B): batched "
                "vision + prefill, then one decode graph step per token for the whole batch")
a = ap.parse_args()
dev = torch.device("cuda:0")
model, ids = bench.build_model(SimpleNamespace(image_size=224, llama_layers=32), dev, 0)
P = 16
g = torch.Generator().manual_seed(0)
Bq = a.batch
img = torch.randn(Bq, 3, 224, 224, generator=g).to(dev)
boxes = model.prepare_boxes([syn.boxes(a.rois, g).to(dev) for _ in range(Bq)], 224)
prompt = torch.stack([syn.prompt_ids(ids, P, a.rois, g) for _ in range(Bq)]).to(dev)


def request(n_new):
    """One batch of Bq requests: vision tower + region module + splice (batched), prefill, n_new greedy tokens each."""
    emb = model.embed_inputs(prompt, img, boxes)
    if Bq > 1:
        return model.llama.decode_graph_batch(emb, n_new)[0]
    return model.llama.greedy_graph(emb, n_new)


request(8)                                              # warm-up + decode-graph capture
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.requests):
    request(2)
torch.cuda.synchronize()
t_prefill = (time.perf_counter() - t0) / a.requests     # vision + splice + prefill + 2 tokens
t0 = time.perf_counter()
for _ in range(a.requests):
    out = request(a.new_tokens)
torch.cuda.synchronize()
t_full = (time.perf_counter() - t0) / a.requests
print(json.dumps(dict(workload=f"224^2 crop, {a.rois} RoIs, prompt {prompt.size(1)} tokens + {a.new_tokens} greedy tokens, batch {Bq}",
                      prefill_ms=round(1e3 * t_prefill, 2), batch_ms=round(1e3 * t_full, 2),
                      decode_ms_per_step=round(1e3 * (t_full - t_prefill) / (a.new_tokens - 2), 3),
                      requests_per_s=round(Bq / t_full, 3), generated_tokens_per_s=round(Bq * a.new_tokens / t_full, 1),
                      region_tokens_per_s=round(Bq * a.rois / t_full, 1), tokens_out=len(out))))
