#!/usr/bin/env python3
"""tools/two_streams.py -- does running two independent batch-1 images on two HIP streams fill the
wave-quantisation tails?  (experiment; bench.py's headline stays single-stream unless this pays)"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)   # inference tool: no autograd seam
import bench
args = bench.parse()
dev = "cuda:0"
torch.cuda.set_device(0)
model, ids = bench.build_model(args, dev, 100)
image, boxes, prompt = bench.make_inputs(args, ids, dev, 0)
# second context: shares every weight, owns its KV cache
models = [model]
for _ in range(3):
    m2 = copy.copy(model)
    m2.llama = copy.copy(model.llama)
    m2.llama._alloc_cache(1)
    models.append(m2)
streams = [torch.cuda.Stream() for _ in models]
def run(n_streams, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        s = i % n_streams
        with torch.cuda.stream(streams[s]):
            models[s](input_ids=prompt, images=image, bboxes=boxes)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps
for n in (2, 1, 2, 3, 2):
    run(n, 4)
    dt = run(n, 12)
    print(f"{n} stream(s): {dt*1e3:.2f} ms/image  {args.rois/dt:.1f} region-tokens/s", flush=True)
