"""Stage-2 step (everything but the ViT trainable) at the per-GPU batch of BASELINE configs[3]
(16 images of 336^2, 32 regions each, T = 767) on ONE MI355X: does a 7 B replica + fp32 masters + Adam state + the batch's
activations fit 288 GB with per-layer gradient checkpointing, and what does a step cost?"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gpt4roi_amd import synthetic as syn
from gpt4roi_amd.train import FullTrainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--no-checkpoint", action="store_true")
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = "cuda:0"
args = argparse.Namespace(image_size=336, rois=32, llama_layers=32)
model, ids = bench.build_model(args, dev, seed=7)
model.gradient_checkpointing = not a.no_checkpoint
g = torch.Generator().manual_seed(3)
B, P = a.batch, 24
images = torch.randn(B, 3, 336, 336, generator=g).to(dev)
boxes = [syn.boxes(32, g).to(dev) for _ in range(B)]
prompt = torch.stack([syn.prompt_ids(ids, P, 32, g) for _ in range(B)]).to(dev)
labels = prompt.clone(); labels[:, :42 + P * P] = -100; labels[labels >= 32000] = -100
model.llama.reset(B)
tr = FullTrainer(model, lr=2e-5)
torch.cuda.reset_peak_memory_stats(dev)
losses, times = [], []
for s in range(a.steps + 1):
    torch.cuda.synchronize(); t0 = time.time()
    losses.append(float(tr.step(prompt, images, boxes, labels)))
    torch.cuda.synchronize(); times.append(time.time() - t0)
print(json.dumps({"batch_per_gpu": B, "tokens": int(prompt.numel()), "checkpoint": not a.no_checkpoint,
                  "ms_per_step": [round(1e3 * t, 1) for t in times], "loss": [round(l, 4) for l in losses],
                  "peak_mem_GiB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                  "reserved_GiB": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 1)}))
