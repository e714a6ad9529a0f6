#!/usr/bin/env python3
"""bench.py -- region-tokens/s of the region-feature path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1: either launched as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py
    --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or as the plain command above -- bench.py then starts
    the N ranks itself through the same launcher (one process per GPU, RCCL over xGMI) and rank 0 prints the line.

Workload (BASELINE.json configs[1], the one the metric is quoted on): batch-1 requests of ONE 336x336 image with 32 RoIs
each: CLIP ViT-L/14 (23 of 24 blocks) -> 4-level pyramid + 5 fuse rounds -> multi-level
RoIAlign -> pconvs / flatten_linear / pos-embed / updims -> mm_projector -> splice + <bbox>
injection -> LLaMA-7B prefill forward with logits for every position.  A "step" is ONE launch sequence over
`--batch` such requests merged (continuous batching, default 16: the weights are streamed once for all of them and the
LLaMA GEMMs get M = 8 x 767 rows, i.e. whole waves of tiles on the 256 CUs; profiles/r04_merge_sweep.txt: 1 x 1 1652, 1 x 2 1856,
4 x 2 2011, 8 x 2 2096 region-tokens/s on one box); `value` counts every request's 32 region
tokens.  `--batch 1 --streams 1` is the strictly serial batch-1 latency, reported beside the headline as `single_request`.
Inputs (images, boxes, token ids) and all weights are resident in HBM before the timed region.  Weights are seeded random tensors of the real shapes (no checkpoints in this
environment).  Multi-GPU: the path shards by image with no data-path collective (inference
replicas), so scaling is weak and `value` = all ranks' region tokens / max-rank time.

Extra objects on the JSON line:
  roofline     -- the kernel family with the largest share of a step, timed per launch with HIP
                  events on the launch stream in an instrumented (untimed) extra step
  cpu_baseline -- the reference's CPU path for the stages north_star names (mmcv CPU RoIAlign as
                  compiled in oracle/_ref, 1 thread by construction, + CLIP ViT-L/14 fp32 on the
                  host cores), rank 0 / N = 1 only
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA peak, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0       # HBM3E spec peak, same table


def forward_flops_per_request(P, rois, T, llama_layers=32, C=1024, hidden=4096, inter=11008, vocab=32006):
    """Forward FLOPs (2 x MAC) of ONE request of the path, SURVEY.md 8d: ViT-L/14 23 blocks (8 S C^2 + 16 S C^2 + 4 S^2 C, S = P^2 + 1) +
    patch embed; 1x1 input convs + 5 fuse rounds of 3x3 convs over the 85 P^2 pyramid pixels; per RoI 4 pconvs (14 x 14 bins) +
    flatten_linear + updims; projector; LLaMA prefill (q/k/v/o + SwiGLU MLP + lm_head per token, causal attention)."""
    S = P * P + 1
    npix = 85 * P * P
    return (23 * (24.0 * S * C * C + 4.0 * S * S * C) + 2.0 * (S - 1) * 588 * C
            + 2.0 * 1026 * C * npix + 5 * 2.0 * 9 * C * C * npix
            + rois * (4 * 2.0 * 9 * C * C * 196 + 2.0 * 196 * C * 1024 + 2.0 * 1024 * hidden)
            + 2.0 * P * P * C * hidden
            + T * (llama_layers * (8.0 * hidden * hidden + 6.0 * hidden * inter) + 2.0 * hidden * vocab)
            + llama_layers * 2.0 * T * T * hidden)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--image-size", type=int, default=336)
    ap.add_argument("--rois", type=int, default=32)
    ap.add_argument("--llama-layers", type=int, default=32, help="debug only; anything but 32 marks the line invalid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent launch sequences in flight per GPU, one HIP stream each (1 = strictly serial; with 16 "
                         "merged requests a second sequence no longer helps: profiles/r04_merge_sweep.txt)")
    ap.add_argument("--dtype", default="fp16", choices=["bf16", "fp16"],
                    help="16-bit storage type of the whole path.  fp16 (the default) is what the reference SERVES configs[1] in "
                         "(app.py:74-98, boxes :271, image :296) and the storage type whose greedy ids equal HF fp32's on every "
                         "tested draw (64 of 64 new tokens, profiles/r05_greedy_parity.json); bf16 is its training dtype -- "
                         "same MFMA rate, same bytes, reported under extras")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (fp16, 224^2, single request)")
    ap.add_argument("--batch", type=int, default=16,
                    help="batch-1 requests merged into ONE launch sequence per step (continuous batching: the weights are "
                         "streamed once for all of them); value counts every request's region tokens")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a hipGraph")
    ap.add_argument("--train-steps", type=int, default=4,
                    help="extra (not part of `value`): timed stage-1 training steps (SURVEY.md 8d config 3: batch 8 per GPU, "
                         "region module + projector trainable, gradient exchange over RCCL when N > 1); 0 = skip")
    ap.add_argument("--train-batch", type=int, default=8)
    ap.add_argument("--train-deadline", type=float, default=300.0,
                    help="N > 1 only: seconds the training leg (with its gradient exchange) may take before the line is printed without it")
    ap.add_argument("--decode-batch", type=int, default=8, help="extra: batched greedy decode of this many sequences (<= 1 skips it)")
    ap.add_argument("--mixed-tokens", type=int, default=512,
                    help="extra: SURVEY.md 8d config 5 on one GPU (224^2 crop, 64 RoIs, --decode-batch requests together): greedy "
                         "tokens generated per request after the prefill; 0 = skip")
    ap.add_argument("--stage2-steps", type=int, default=2,
                    help="extra: timed stage-2 training steps (SURVEY.md 8d config 4 per GPU: LLaMA-7B unfrozen, fp32 masters + "
                         "fused AdamW, --train-batch images); N = 1 only; 0 = skip")
    ap.add_argument("--parity-tokens", type=int, default=64,
                    help="N = 1 only, measured in THIS run (never part of `value`): greedy ids of the merged launch sequence -- the "
                         "--batch requests prefilled together (M = batch x T) and decoded together for this many tokens -- against HF "
                         "LlamaForCausalLM fp32 built from the same weights, from the path's own prompt embeddings; 0 = skip")
    ap.add_argument("--decode-tokens", type=int, default=32,
                    help="extra (not part of `value`): greedy KV-cache decode steps timed after the prefill")
    return ap.parse_args()


def build_model(args, device, seed, dtype=torch.bfloat16):
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.llama import LlamaDecoder
    from gpt4roi_amd.spi_llava import SPILlavaLlamaModel
    from gpt4roi_amd.vit import ClipVisionTower
    ids = syn.token_ids(32000)
    bf = dtype
    v = syn.CLIP_L14
    vsd = syn.vit_state(v["hidden"], v["inter"], v["layers"], args.image_size, seed=seed, device=device, dtype=bf)
    tower = ClipVisionTower(vsd, heads=v["heads"], device=device, dtype=bf)
    del vsd
    l = syn.LLAMA_7B
    lsd = syn.llama_state(l["hidden"], l["inter"], args.llama_layers, ids.vocab, seed=seed + 1, device=device, dtype=bf)
    dec = LlamaDecoder(lsd, heads=l["heads"], max_positions=2048, device=device, dtype=bf)
    del lsd
    model = SPILlavaLlamaModel(tower, dec, ids, embed_dims=v["hidden"])
    model.spi_module.to(device)
    model.mm_projector.to(device)
    syn.spi_state_gpu(model.spi_module, seed=seed + 2)
    # the projector too comes from a seeded generator (round 6: it was left at nn.Linear's default, i.e. a different draw in every
    # process -- the live greedy-id leg then met different fp32 near-ties from run to run)
    gp = torch.Generator(device=device).manual_seed(seed + 3)
    with torch.no_grad():
        pj = model.mm_projector
        pj.weight.copy_(torch.randn(pj.weight.shape, generator=gp, device=device) / pj.weight.size(1) ** 0.5)
        pj.bias.copy_(torch.randn(pj.bias.shape, generator=gp, device=device) * 0.05)
    model.prepare()
    # the kernel-ready copies are what the path reads; drop the fp32 masters of the big layers
    torch.cuda.empty_cache()
    return model, ids


def make_inputs(args, ids, device, seed, batch=None, image_size=None):
    from gpt4roi_amd import synthetic as syn
    g = torch.Generator().manual_seed(seed)
    size = image_size or args.image_size
    P = size // 14
    B = max(1, args.batch if batch is None else batch)
    image = torch.randn(B, 3, size, size, generator=g).to(device)
    boxes = [syn.boxes(args.rois, g).to(device) for _ in range(B)]
    prompt = torch.stack([syn.prompt_ids(ids, P, args.rois, g) for _ in range(B)]).to(device)
    return image, boxes, prompt


def timed_replay(model, image, boxes, prompt, steps, warmup=2, streams=1, attention_mask=None):
    """Seconds per launch sequence of `model` over (image, boxes, prompt), hipGraph replay on `streams` HIP streams
    (one request context each), device-synchronised both sides.  Used by the extra legs (never by `value`).
    attention_mask: a prepared llama.RaggedLayout (requests of different prompt lengths merged; tools/ragged_merge.py)."""
    size = image.size(-1)
    ctxs = [model] + [model.clone_context() for _ in range(streams - 1)]
    sts = [torch.cuda.Stream(device=image.device) for _ in ctxs]
    reqs = [c.prepare_boxes(boxes, size) for c in ctxs]
    graphs = []
    for c, st, rq in zip(ctxs, sts, reqs):
        with torch.cuda.stream(st):
            c(input_ids=prompt, images=image, bboxes=rq, attention_mask=attention_mask)
            c(input_ids=prompt, images=image, bboxes=rq, attention_mask=attention_mask)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            keep = c(input_ids=prompt, images=image, bboxes=rq, attention_mask=attention_mask)
        graphs.append((g, keep))
    torch.cuda.synchronize()

    def run(n):
        for i in range(n):
            with torch.cuda.stream(sts[i % len(sts)]):
                graphs[i % len(sts)][0].replay()
    run(warmup * len(sts))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    del graphs
    return dt


def cpu_baseline(args):
    """Reference CPU RoIAlign (oracle/_ref: the reference's own sources, single-threaded by
    construction) on the 4 pyramid levels + CLIP ViT-L/14 fp32 on the host cores, one image."""
    import numpy as np
    from oracle import roi_align as O
    from gpt4roi_amd import synthetic as syn
    P = args.image_size // 14
    g = torch.Generator().manual_seed(0)
    kind = "reference" if O.load_ref() is not None else "port"
    rois = torch.cat([torch.zeros(args.rois, 1), syn.boxes(args.rois, g) * 14 * P], 1).numpy().astype(np.float32)
    t_roi = 0.0
    for lvl, stride in enumerate([14 / 8, 14 / 4, 14 / 2, 14]):
        side = P * 2 ** (3 - lvl)
        x = torch.randn(1, 1024, side, side, generator=g).numpy()
        t0 = time.perf_counter()
        (O.ref_forward if kind == "reference" else O.forward)(x, rois, 14, np.float32(1.0 / stride), 2)
        t_roi += time.perf_counter() - t0
    # CLIP ViT-L/14 on the host cores: HF transformers' own CLIPVisionModel (the arithmetic the reference calls at
    # spi_llava.py:66-67; third-party, pinned by the reference at git cae78c46 -- the container's release is timed), all
    # 24 layers with output_hidden_states=True as the reference runs it, fp32, random weights, on 32 of the host's threads:
    # this M = 577 workload does not scale past a few dozen threads -- with all 256 threads of the GPU box one forward took
    # 59.5 s against 0.40 s at 32 (measured in round 2, profiles/r02_bench.log), so the all-core timing is opt-in
    # (G4R_CPU_BASELINE_ALL_CORES=1) and the line states both the threads used and the host's core count.
    import transformers
    from transformers import CLIPVisionConfig, CLIPVisionModel
    v = syn.CLIP_L14
    cfg = CLIPVisionConfig(hidden_size=v["hidden"], intermediate_size=v["inter"], num_hidden_layers=v["layers"],
                           num_attention_heads=v["heads"], patch_size=14, image_size=args.image_size)
    torch.manual_seed(0)
    hf = CLIPVisionModel(cfg).eval()
    img = torch.randn(1, 3, args.image_size, args.image_size, generator=g)
    t_by_threads = {}
    with torch.no_grad():
        counts = {min(32, os.cpu_count())}
        if os.environ.get("G4R_CPU_BASELINE_ALL_CORES") == "1":
            counts.add(os.cpu_count())
        for nthr in sorted(counts):
            torch.set_num_threads(nthr)
            hf(pixel_values=img, output_hidden_states=True)               # thread pool warm-up
            t0 = time.perf_counter()
            for _ in range(3):
                hf(pixel_values=img, output_hidden_states=True)
            t_by_threads[nthr] = (time.perf_counter() - t0) / 3
    nbest = min(t_by_threads, key=t_by_threads.get)
    t_vit = t_by_threads[nbest]
    return {"value": round(args.rois / (t_roi + t_vit), 3), "unit": "region-tokens/s", "cores": nbest,
            "kind": kind, "host_cpu_count": os.cpu_count(),
            "parts": {"roi_align": {"kind": kind, "what": "the reference's own mmcv cpu/roi_align.cpp compiled unmodified (oracle/_ref)"
                                    if kind == "reference" else "oracle/roi_align_oracle.c restatement",
                                    "threads": 1, "seconds": round(t_roi, 4)},
                      "vit": {"kind": "third-party", "what": f"HF transformers {transformers.__version__} CLIPVisionModel, 24 layers, "
                                                               "output_hidden_states=True, fp32",
                              "threads": nbest, "seconds": round(t_vit, 4),
                              "seconds_by_threads": {str(k): round(x, 4) for k, x in t_by_threads.items()}}},
            "sample": (f"1 image {args.image_size}^2, {args.rois} RoIs: mmcv CPU roi_align fp32 x4 levels (1 thread, single-threaded "
                       f"by construction, {t_roi:.3f} s) + HF CLIPVisionModel ViT-L/14 fp32 ({nbest} of {os.cpu_count()} host "
                       f"threads, {t_vit:.3f} s); fuse convs / LLaMA-7B are not part of the CPU sample"),
            "roi_align_s": round(t_roi, 4), "vit_s": round(t_vit, 4)}


def vit_roofline(model, args, device):
    """CLIP ViT-L/14 alone (FLOPs_ViT(23 layers) / t_ViT / 2.5 PF) at batch 1, at the batch the headline merges
    and at the batch the training configs feed it (configs 3/4: B = 8/16).  Timed the way the path runs it: ONE hipGraph
    replay of the tower per forward, HIP events on the launch stream (the eager timing of round 3 charged the
    tower ~1 ms of host launch gaps at batch 1)."""
    tower = model.vision_tower[0]
    S = (args.image_size // 14) ** 2 + 1
    C = tower.hidden
    per_layer = 24.0 * S * C * C + 4.0 * S * S * C                  # 8SC^2 (q,k,v,o) + 16SC^2 (MLP) + 4S^2C (attention)
    flops1 = len(tower.layers) * per_layer + 2.0 * (S - 1) * 588 * C
    out = {"timing": "hipGraph replay of the tower, HIP events on the launch stream"}
    st = torch.cuda.Stream(device=device)
    for B in sorted({1, max(1, args.batch), 8}):
        img = torch.randn(B, 3, args.image_size, args.image_size, device=device)
        with torch.cuda.stream(st):
            for _ in range(2):
                tower.forward(img)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            keep = tower.forward(img)
        n = 8
        with torch.cuda.stream(st):
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                g.replay()
            e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        del g, keep
        tf = B * flops1 / (ms * 1e-3) / 1e12
        out[f"batch{B}"] = {"ms": round(ms, 3), "ms_per_image": round(ms / B, 3), "achieved": round(tf, 1), "peak": PEAK_BF16_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4), "flops": int(B * flops1)}
    return out


def live_parity_leg(args, model, image, boxes, prompt, device):
    """Greedy-id parity of the configuration `value` is timed on, MEASURED IN THIS RUN (the line used to
    quote a file written by a test on another box, for the single-request dispatch).  The --batch requests of the timed
    step go through stages a4-a15 (ViT, region module, projector, splice) and ONE merged decoder prefill (M = batch x T rows:
    the dense 256 x 256 tile in whole waves) + `decode_graph_batch`; the reference side is HF `LlamaForCausalLM` (the
    arithmetic spi_llava.py:198-205 calls) in fp32 with eager attention, built from the SAME weights (export_hf_state_dict),
    prefilled and decoded one request at a time from the SAME prompt embeddings.  What this leg does not cover -- the vision
    and region stages against their oracle -- is tests/test_fullwidth_gpu.py::test_sixteen_merged_requests_... .
    Returns exact-match lengths per request; a difference is reported with HF's fp32 logit gap between the two choices."""
    from transformers import LlamaConfig, LlamaForCausalLM
    n_new = args.parity_tokens
    dec = model.llama
    t0 = time.perf_counter()
    emb = model.embed_inputs(prompt, image, model.prepare_boxes(boxes, args.image_size))
    got = dec.decode_graph_batch(emb, n_new) if emb.size(0) > 1 else [dec.greedy_graph(emb, n_new)]
    torch.cuda.synchronize()
    t_hip = time.perf_counter() - t0
    sd = dec.export_hf_state_dict()
    V, C = sd["lm_head.weight"].shape
    cfg = LlamaConfig(vocab_size=V, hidden_size=C, intermediate_size=sd["model.layers.0.mlp.down_proj.weight"].size(1),
                      num_hidden_layers=len(dec.layers), num_attention_heads=dec.heads, num_key_value_heads=dec.heads,
                      rms_norm_eps=1e-6, max_position_embeddings=2048, attention_bias=False, tie_word_embeddings=False,
                      rope_theta=10000.0, attn_implementation="eager")
    with torch.device(device):
        hf = LlamaForCausalLM(cfg).float().eval()
    r = hf.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    del sd
    embed = hf.get_input_embeddings()
    lens, first = [], None
    t0 = time.perf_counter()
    for b in range(emb.size(0)):
        o = hf(inputs_embeds=emb[b:b + 1].float(), use_cache=True)
        past, last = o.past_key_values, o.logits[0, -1]
        k = n_new
        for s_ in range(n_new):
            nxt = int(last.argmax())
            if nxt != got[b][s_]:
                k = s_
                if first is None:
                    first = {"request": b, "step": s_, "hf_fp32_gap_between_the_two_choices": float(last[nxt] - last[got[b][s_]]),
                             "hf_fp32_logit_range": float(last.max() - last.min())}
                break
            o = hf(inputs_embeds=embed(torch.tensor([[nxt]], device=device)), past_key_values=past, use_cache=True)
            past, last = o.past_key_values, o.logits[0, -1]
        lens.append(k)
        del o, past
    torch.cuda.synchronize()
    t_hf = time.perf_counter() - t0
    del hf
    torch.cuda.empty_cache()
    return {"scope": f"merged{emb.size(0)}: the {emb.size(0)} requests of the timed step, ONE merged decoder prefill (M = {emb.size(0) * emb.size(1)}) "
                     f"+ decode_graph_batch, from the path's own prompt embeddings", "measured_in_this_run": True,
            "against": "HF transformers LlamaForCausalLM fp32, eager attention, the same weights (export_hf_state_dict), one request at a "
                       "time, on the GPU", "dtype": args.dtype, "new_tokens": n_new, "requests": int(emb.size(0)),
            "exact": all(k == n_new for k in lens), "exact_len": lens, "first_divergence": first,
            "seconds": {"hip_embed_prefill_decode": round(t_hip, 2), "hf_fp32": round(t_hf, 2)}}


def train_leg(args, model, ids, device, rank, world, dist, agg_device):
    """Stage-1 training step (train_stage1.sh: region module [+ projector] trainable, ViT / LLaMA frozen) on RefCOCO-shaped
    synthetic batches (B images per GPU, 1..15 regions each, refcoco.py:55), data-parallel over the
    ranks: gradients go to the bucketed reduce-scatter + all-gather exchange as the backward produces them.  Timed like
    the headline (barrier + synchronize both sides, max over ranks)."""
    from gpt4roi_amd import replicas
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.train import RegionTrainer
    B, P = args.train_batch, args.image_size // 14
    g = torch.Generator().manual_seed(5000 + rank)
    n_i = torch.randint(1, 16, (B,), generator=g).tolist()
    images = torch.randn(B, 3, args.image_size, args.image_size, generator=g).to(device)
    boxes = [syn.boxes(n, g).to(device) for n in n_i]
    prompt = torch.stack([syn.prompt_ids(ids, P, n, g, question_len=20 + 4 * (15 - n)) for n in n_i]).to(device)
    labels = prompt.clone()
    labels[:, :42 + P * P] = -100
    labels[labels >= 32000] = -100
    torch.cuda.reset_peak_memory_stats(device)
    tr = RegionTrainer(model, lr=2e-5, train_projector=True)
    tr.step(prompt, images, boxes, labels)
    tr.step(prompt, images, boxes, labels)
    losses = []

    def step():
        losses.append(tr.step(prompt, images, boxes, labels))
    dt_local = replicas.timed_steps(step, args.train_steps, torch.cuda.synchronize, dist)
    regions = sum(n_i)
    tot, dt = replicas.aggregate(regions * args.train_steps, dt_local, dist, device=agg_device)
    tot = int(round(tot))
    # per-family fractions of the MFMA peak over ONE instrumented extra step (HIP events per launch on the launch stream),
    # rank 0 only, single-rank runs only (with world > 1 the extra step would need every rank in its collectives)
    train_roofline = None
    if rank == 0 and world == 1 and not args.no_roofline:
        from gpt4roi_amd import kernels as K
        K.PROFILER.start()
        tr.step(prompt, images, boxes, labels)
        agg = K.PROFILER.stop()
        tsum = sum(a["ms"] for a in agg.values())
        fams = {"gemm": lambda t: t.startswith("gemm_bf16_nt") or t.startswith("gemv") or t == "small_linear",
                "conv": lambda t: t.startswith("conv3x3_igemm"), "weight_grad_tn": lambda t: t.startswith("conv3x3_wgrad_tn")
                or t.startswith("gemm_tn"), "attention_fwd": lambda t: t.startswith("flash_attn<"),
                "attention_bwd": lambda t: t.startswith("flash_attn_bwd")}
        train_roofline = {"instrumented_step_kernel_ms": round(tsum, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "families": {}}
        for fname, pred in fams.items():
            ms_ = sum(a["ms"] for t, a in agg.items() if pred(t))
            fl_ = sum(a["flops"] for t, a in agg.items() if pred(t))
            if ms_ > 0:
                tf_ = fl_ / (ms_ * 1e-3) / 1e12
                train_roofline["families"][fname] = {"ms": round(ms_, 2), "share": round(ms_ / tsum, 3), "achieved": round(tf_, 1),
                                                     "frac": round(tf_ / PEAK_BF16_TFLOPS, 4),
                                                     "launches": sum(a["calls"] for t, a in agg.items() if pred(t))}
        other = sorted(((t, a["ms"]) for t, a in agg.items() if not any(pred(t) for pred in fams.values())), key=lambda kv: -kv[1])
        train_roofline["other_top"] = {t: round(ms_, 2) for t, ms_ in other[:8]}
    out = {"what": "stage-1 step (SURVEY.md 8d config 3): forward + hand-written backward + exchange + clip + fused AdamW",
           "batch_per_gpu": B, "tokens_per_sequence": int(prompt.size(1)), "regions_per_step_all_ranks": tot // max(1, args.train_steps),
           "steps": args.train_steps, "ms_per_step": round(1e3 * dt / args.train_steps, 2),
           "images_per_s": round(B * world * args.train_steps / dt, 2),
           "region_tokens_trained_per_s": round(tot / dt, 1), "peak_mem_GiB": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1),
           "loss_first_last": [round(float(losses[0]), 4), round(float(losses[-1]), 4)], "roofline": train_roofline,
           "exchange": None if tr.reducer is None else {"algo": tr.reducer.algo, "buckets": tr.reducer.describe(),
                                                        "overlapped_with_backward": True}}
    return out


def stage2_leg(args, model, ids, device):
    """Stage-2 step (train_stage2.sh / SURVEY.md 8d config 4, the per-GPU part): everything but the vision tower trains --
    forward + hand-written backward of every stage incl. all LLaMA weight gradients + clip + fused AdamW on fp32 masters.
    One rank (the exchange of the 6.7 B gradients is covered by the gloo tests and needs the 8-GPU node).  Two batches on ONE
    trainer: (a) --train-batch RefCOCO-shaped images (1..15 regions), activations kept; (b) configs[3]'s own per-GPU share --
    16 images x 32 regions (global 128 over 8 GPUs, train_stage2.sh:40-52), T = 767 -- with per-layer decoder checkpointing
    (`--gradient_checkpointing True`, train_stage2.sh:47), which is what makes it fit beside the replicated masters + Adam."""
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.train import FullTrainer
    P = args.image_size // 14
    torch.cuda.empty_cache()
    tr = FullTrainer(model, lr=2e-5)

    def run(B, n_i, seed, checkpoint, what):
        g = torch.Generator().manual_seed(seed)
        images = torch.randn(B, 3, args.image_size, args.image_size, generator=g).to(device)
        boxes = [syn.boxes(n, g).to(device) for n in n_i]
        prompt = torch.stack([syn.prompt_ids(ids, P, n, g, question_len=20 + 4 * (15 - min(n, 15))) for n in n_i]).to(device)
        labels = prompt.clone()
        labels[:, :42 + P * P] = -100
        labels[labels >= 32000] = -100
        model.gradient_checkpointing = bool(checkpoint)
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(device)
        losses = [tr.step(prompt, images, boxes, labels).item() for _ in range(2)]      # warm-up: allocations, plans, the allocator's pools
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.stage2_steps):
            losses.append(tr.step(prompt, images, boxes, labels).item())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.stage2_steps
        return {"what": what, "batch_per_gpu": B, "tokens_per_sequence": int(prompt.size(1)), "decoder_checkpointing": bool(checkpoint),
                "steps": args.stage2_steps, "ms_per_step": round(1e3 * dt, 2), "images_per_s": round(B / dt, 2),
                "tokens_per_s": round(B * prompt.size(1) / dt, 1), "region_tokens_trained_per_s": round(sum(n_i) / dt, 1),
                "peak_mem_GiB": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1),
                "loss_first_last": [round(losses[0], 4), round(losses[-1], 4)]}
    try:
        B = args.train_batch
        n_i = torch.randint(1, 16, (B,), generator=torch.Generator().manual_seed(6000)).tolist()
        out = run(B, n_i, 6000, False, "stage-2 step (SURVEY.md 8d config 4, one GPU's share): forward + backward with every LLaMA-7B "
                                         "weight gradient + clip + fused AdamW on fp32 masters")
        try:
            out["config4_batch16"] = run(16, [32] * 16, 6001, True,
                                         "the same step at configs[3]'s OWN per-GPU batch: 16 images x 32 regions (global 128 on 8 GPUs), "
                                         "per-layer decoder checkpointing (train_stage2.sh:40-52)")
        except Exception as ex:                              # e.g. out of memory beside the replicated masters + Adam state
            out["config4_batch16"] = {"error": repr(ex)[:300]}
            torch.cuda.empty_cache()
    finally:
        model.gradient_checkpointing = False
    return out


class _Deadline(Exception):
    pass


def with_deadline(fn, seconds, device_index):
    """Run fn() on a worker thread and wait at most `seconds`: the multi-rank training leg's collectives (RCCL over xGMI) have
    never run on more than one GPU here, and a rank stuck in one must not cost the run its headline line.  On a timeout the
    caller reports the leg as an error and the process leaves through os._exit after the line is printed (a stuck collective
    cannot be cancelled)."""
    import threading
    box = {}

    def run():
        try:
            if torch.cuda.is_available():
                torch.cuda.set_device(device_index)
            torch.set_grad_enabled(False)           # (thread-local: the main thread switched it off for the whole run)
            box["out"] = fn()
        except Exception as ex:                     # noqa: BLE001
            box["err"] = ex
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        raise _Deadline(f"still running after {seconds} s")
    if "err" in box:
        raise box["err"]
    return box["out"]


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start the N ranks through torch.distributed.run (one
    process per GPU, rendezvous on 127.0.0.1) and return their exit code.  The reference's launch line is the same launcher:
    train_stage1.sh:11 `torchrun --nproc_per_node=4`."""
    import socket
    import subprocess
    if "G4R_FORCE_DEVICE" not in os.environ:                 # (test hook: several ranks on one GPU over gloo)
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the path has no CPU fallback")
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}")
    # test hooks (not used by the driver): run >1 rank on a 1-GPU box over gloo to exercise this code path
    backend = os.environ.get("G4R_DIST_BACKEND", "nccl")                  # "nccl" is RCCL on ROCm
    if "G4R_FORCE_DEVICE" in os.environ:
        local = int(os.environ["G4R_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
    from gpt4roi_amd import kernels as K

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    model, ids = build_model(args, device, seed=100 + rank, dtype=dtype)
    image, boxes, prompt = make_inputs(args, ids, device, seed=rank)

    from gpt4roi_amd import replicas
    last = {}
    # `--streams S` request contexts share the weights; each has its own KV cache and HIP stream.
    # A step is still ONE batch-1 image; steps are issued round-robin over the contexts.
    ctxs = [model] + [model.clone_context() for _ in range(max(1, args.streams) - 1)]
    streams = [torch.cuda.Stream(device=device) for _ in ctxs]
    counter = {"i": 0}
    # host-side request preparation happens once, outside the launch sequence (RoI table, offsets)
    reqs = [c.prepare_boxes(boxes, args.image_size) for c in ctxs]

    torch.set_grad_enabled(False)        # inference (app.py runs under torch.inference_mode): no autograd seam, no host sync

    def eager(i):
        return ctxs[i](input_ids=prompt, images=image, bboxes=reqs[i])      # logits [1, T, V] fp32

    # One image = ~900 kernel launches; captured once per context into a hipGraph so that a step costs
    # the host one replay call (eager Python launching is what bounds 2-3 concurrent requests).
    graphs = [None] * len(ctxs)
    if not args.no_graph:
        try:
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    eager(i)
                    eager(i)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    last["graph_out_%d" % i] = eager(i)
                graphs[i] = g
            torch.cuda.synchronize()
        except Exception as ex:                                   # keep the bench alive; say so in the JSON
            graphs = [None] * len(ctxs)
            last["graph_error"] = repr(ex)
            torch.cuda.synchronize()

    def step():
        i = counter["i"] % len(ctxs)
        counter["i"] += 1
        with torch.cuda.stream(streams[i]):
            if graphs[i] is not None:
                graphs[i].replay()
            else:
                eager(i)

    def serial_step():
        model(input_ids=prompt, images=image, bboxes=reqs[0])

    for _ in range(args.warmup * len(ctxs)):
        step()
    torch.cuda.synchronize()
    for c in ctxs:
        c.check_status()
    # barrier + synchronize on both sides of exactly `steps` steps; MAX over ranks
    seg0 = torch.cuda.memory_stats(device).get("num_device_alloc", 0)
    dt_local = replicas.timed_steps(step, args.steps, torch.cuda.synchronize, dist)
    device_allocs_in_timed_region = torch.cuda.memory_stats(device).get("num_device_alloc", 0) - seg0
    _, dt = replicas.aggregate(args.rois * args.batch * args.steps, dt_local, dist, device=device if backend == "nccl" else "cpu")
    # the same K steps strictly serial on one stream (per-image latency), reported beside the headline
    serial_step()
    dt_serial = replicas.timed_steps(serial_step, args.steps, torch.cuda.synchronize, dist) if len(ctxs) > 1 else dt_local
    logits = model(input_ids=prompt, images=image, bboxes=boxes)
    assert torch.isfinite(logits[0, -1]).all(), "non-finite logits"
    del logits

    roofline, kernels = None, None
    if rank == 0 and not args.no_roofline:
        K.PROFILER.start()
        serial_step()
        agg = K.PROFILER.stop()
        tot = sum(a["ms"] for a in agg.values())
        kernels = {k: {"calls": a["calls"], "ms": round(a["ms"], 3), "share": round(a["ms"] / tot, 3),
                       "avg_us": round(1e3 * a["ms"] / a["calls"], 2),
                       "TFLOP/s": round(a["flops"] / (a["ms"] * 1e-3) / 1e12, 1) if a["flops"] else None,
                       "GB/s": round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1)}
                   for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        # dominant kernel BY DEVICE SYMBOL (the labels split one symbol -- the 192 x 256 ring kernel with and
        # without K slices -- into two rows and so reported the conv): "+splitk" launches run the same kernel symbol
        by_sym = {}
        for k, a_ in agg.items():
            b_ = by_sym.setdefault(k.replace("+splitk", ""), dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            for f_ in ("calls", "ms", "flops", "bytes"):
                b_[f_] += a_[f_]
        dom, a = max(by_sym.items(), key=lambda kv: kv[1]["ms"])
        ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None,
                    "launches_per_step": a["calls"], "avg_launch_us": round(1e3 * a["ms"] / a["calls"], 2),
                    "share_of_step": round(a["ms"] / tot, 3)}
        # HBM traffic, effective clock and MFMA-pipe utilisation per launch from the committed counter passes (collected
        # separately, as rocprofv3 requires: the newest profiles/rNN_pmc_report.json, written by tools/pmc_report.py from
        # `rocprofv3 --pmc ... --kernel-trace -- python bench.py --steps 2 --streams 1 --no-graph`); null when no
        # profile is committed for this kernel
        pmc, pmc_file = {}, None
        try:
            import glob
            import re as _re
            cands = glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_report.json"))
            # newest round first (r03 > r02), then the most recently written file of that round
            cands.sort(key=lambda f: (int(_re.search(r"r(\d+)_", os.path.basename(f)).group(1)), os.path.getmtime(f)))
            if cands:
                pmc_file = os.path.relpath(cands[-1], ROOT)
                pmc = json.load(open(cands[-1]))
        except Exception:
            pass
        # kernel names as the counter pass records them, matched by prefix (template tails change between rounds)
        PMC_PREFIX = {"gemm_bf16_nt<256x256w4k64>": ("void gemm_bf16_w4k64_kernel<0, false",       # (all epilogue modes of the per-tile form
                                                     "void gemm_bf16_w4k64p_kernel<0,"),          #  and of the persistent form, round 6)
                      "conv3x3_igemm<256x256w4k64>": "void gemm_bf16_w4k64_kernel<2, false",
                      "gemm_bf16_nt<256x256pp32>": "void gemm_bf16_pp32_kernel<0, false, 256, 256",
                      "gemm_bf16_nt<192x256pp32>": "void gemm_bf16_pp32_kernel<0, false, 192, 256",
                      "conv3x3_igemm<256x256pp32>": "void gemm_bf16_pp32_kernel<2, false, 256, 256",   # the one-launch-per-round form
                      "gemm_bf16_nt<128x128w8s4>": "void gemm_bf16_nt_kernel<128, 128, 2, 4, 0, true, 4, 64, 0>",
                      "roi_align_mlvl_nhwc": "void roi_align_mlvl_nhwc_kernel<unsigned short, true>"}

        def pmc_rec(tag):
            """launch-weighted mean over every counter-pass row of the tag's kernel symbol family (the epilogue modes of the
            one-wave-per-SIMD kernel are template instances of one symbol)"""
            pre = PMC_PREFIX.get(tag)
            pre = (pre,) if isinstance(pre, str) else (pre or ())
            rows = [v for k, v in pmc.items() if k != "_meta" and any(k.startswith(x) for x in pre)]
            if not rows:
                return {}
            n = sum(r.get("launches", 1) for r in rows)
            out = {"launches": n, "symbols": len(rows)}
            for key in ("avg_us", "clock_GHz", "mfma_util", "hbm_read_bytes", "hbm_write_bytes"):
                if all(r.get(key) is not None for r in rows):
                    val = sum(r[key] * r.get("launches", 1) for r in rows) / n
                    out[key] = int(val) if key.endswith("bytes") else round(val, 4)
            return out
        rec = pmc_rec(dom)
        # do the committed counters belong to THIS tree's kernels?  (sha of gpt4roi_amd/csrc at collection time, tools/pmc_report.py)
        try:
            import hashlib
            h_ = hashlib.sha256()
            cdir = os.path.join(ROOT, "gpt4roi_amd", "csrc")
            for f_ in sorted(os.listdir(cdir)):
                if f_.endswith((".hip", ".h")):
                    h_.update(open(os.path.join(cdir, f_), "rb").read())
            want_sha = (pmc.get("_meta") or {}).get("kernel_sources_sha256_16")
            roofline["pmc_counters_match_kernel_sources"] = (want_sha == h_.hexdigest()[:16]) if want_sha else None
        except Exception:
            roofline["pmc_counters_match_kernel_sources"] = None
        if not rec and pmc:
            roofline["pmc_error"] = f"no counter row for {dom!r} in {pmc_file}"
        counters_ok = roofline["pmc_counters_match_kernel_sources"] is not False      # (False: counters of OTHER kernel sources)
        if not counters_ok:
            roofline["traffic_note"] = (f"null: the committed counters ({pmc_file}) were collected on different kernel sources; "
                                        "re-run tools/pmc_report.py")
            rec = {}
        if "hbm_read_bytes" in rec:
            roofline["traffic"] = rec["hbm_read_bytes"] + rec.get("hbm_write_bytes", 0)
            roofline["traffic_source"] = (f"{pmc_file} (PMC FETCH_SIZE x 2 x 1024 + WRITE_SIZE x 1024 per launch, "
                                          "the gfx950 correction of MI355X_MICROARCH.md)")
            roofline["algorithmic_bytes_per_launch"] = int(a["bytes"] / a["calls"])
        if rec.get("mfma_util") is not None:
            roofline["pmc"] = {"mfma_util": rec["mfma_util"], "clock_GHz": rec["clock_GHz"],
                               "source": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), clock = GRBM_GUI_ACTIVE / 8 / duration"}
        conv_tag = max((t_ for t_ in agg if t_.startswith("conv3x3_igemm<256x256")), key=lambda t_: agg[t_]["ms"], default=None)
        cv = agg.get(conv_tag)
        if cv and cv.get("flops"):
            crec = pmc_rec(conv_tag) if counters_ok else {}
            tf = cv["flops"] / (cv["ms"] * 1e-3) / 1e12
            roofline["conv"] = {"kernel": conv_tag, "bound": "mfma", "achieved": round(tf, 1),
                                "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_BF16_TFLOPS, 4),
                                "avg_launch_us": round(1e3 * cv["ms"] / cv["calls"], 2),
                                "algorithmic_bytes_per_launch": int(cv["bytes"] / cv["calls"]),
                                "traffic": (crec["hbm_read_bytes"] + crec.get("hbm_write_bytes", 0)) if "hbm_read_bytes" in crec else None,
                                "pmc": ({"mfma_util": crec["mfma_util"], "clock_GHz": crec["clock_GHz"]} if crec.get("mfma_util") is not None else None)}
        # per-family fractions of the MFMA peak over the whole instrumented step (all launches of the family)
        fams = {"gemm": lambda t: t.startswith("gemm_bf16_nt") or t.startswith("gemv") or t == "small_linear",
                "conv": lambda t: t.startswith("conv3x3_igemm"), "attention": lambda t: t.startswith("flash_attn")}
        roofline["families"] = {}
        for fname, pred in fams.items():
            ms_ = sum(a_["ms"] for t, a_ in agg.items() if pred(t))
            fl_ = sum(a_["flops"] for t, a_ in agg.items() if pred(t))
            if ms_ > 0:
                tf_ = fl_ / (ms_ * 1e-3) / 1e12
                roofline["families"][fname] = {"ms_per_step": round(ms_, 3), "ms_per_request": round(ms_ / max(1, args.batch), 3),
                                               "share_of_step": round(ms_ / tot, 3), "achieved": round(tf_, 1), "unit": "TFLOP/s",
                                               "frac": round(tf_ / PEAK_BF16_TFLOPS, 4),
                                               "launches": sum(a_["calls"] for t, a_ in agg.items() if pred(t))}
        ra = agg.get("roi_align_mlvl_nhwc")
        if ra:
            gbs = ra["bytes"] / (ra["ms"] * 1e-3) / 1e9
            roofline["roi_align"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                     "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes": int(ra["bytes"]),
                                     "avg_launch_us": round(1e3 * ra["ms"] / ra["calls"], 2),
                                     "traffic": (lambda r: r["hbm_read_bytes"] + r.get("hbm_write_bytes", 0) if "hbm_read_bytes" in r
                                                 else None)(pmc_rec("roi_align_mlvl_nhwc") if counters_ok else {})}

    decode = None
    if rank == 0 and args.decode_tokens > 0:
        # configs[1] continues with greedy decode from the KV cache; reported beside the headline,
        # never inside it (weight-streaming bound: 13.5 GB of bf16 weights per token)
        emb = model.embed_inputs(prompt[:1], image[:1], model.prepare_boxes(boxes[:1], args.image_size))
        model.llama.greedy_graph(emb, 8)                                  # warm-up + graph capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.llama.greedy_graph(emb, args.decode_tokens + 2)             # prefill + N+2 tokens
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        t0 = time.perf_counter()
        model.llama.greedy_graph(emb, 2)                                  # prefill + 2 tokens
        torch.cuda.synchronize()
        dtd = (t_all - (time.perf_counter() - t0)) / args.decode_tokens
        wbytes = sum(L[k].numel() * 2 for L in model.llama.layers for k in ("wqkv", "wo", "wgu", "wd")) + model.llama.lm_head.numel() * 2
        kvbytes = 2 * 2 * model.llama.hidden * len(model.llama.layers) * (prompt.size(1) + args.decode_tokens // 2)
        decode = {"ms_per_token": round(1e3 * dtd, 3), "tokens_per_s": round(1.0 / dtd, 1),
                  "weight_stream_GBps": round(wbytes / dtd / 1e9, 1),
                  "roofline": {"bound": "hbm", "achieved": round((wbytes + kvbytes) / dtd / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round((wbytes + kvbytes) / dtd / 8e12, 4),
                               "algorithmic_bytes_per_token": int(wbytes + kvbytes),
                               "what": "whole decode step (161 launches: 4 GEMVs + 1 attention per layer, lm_head, token selection); "
                                       "bytes = every bf16 weight once + the K/V rows attended"},
                  "note": "batch 1, KV cache, one hipGraph replay per token (token id and position stay on the device)"}

    if decode is not None and args.decode_batch > 1:
        # the same step for B equal-length sequences sharing one weight stream (SURVEY.md 8d config 5 decodes batches)
        try:
            Bd, nd = args.decode_batch, args.decode_tokens
            embB = emb.expand(Bd, -1, -1).contiguous()
            model.llama.decode_graph_batch(embB, nd + 2)                      # warm-up + graph capture
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.llama.decode_graph_batch(embB, nd + 2)
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            t0 = time.perf_counter()
            model.llama.decode_graph_batch(embB, 2)
            torch.cuda.synchronize()
            dtb = (t_all - (time.perf_counter() - t0)) / nd
            decode["batched"] = {"batch": Bd, "ms_per_step": round(1e3 * dtb, 3), "tokens_per_s": round(Bd / dtb, 1),
                                 "note": "B equal-length sequences, one hipGraph replay per step for the whole batch"}
            del embB
        except Exception as ex:
            decode["batched"] = {"error": repr(ex)}

    live_parity = None
    if rank == 0 and world == 1 and args.parity_tokens > 0:
        try:
            live_parity = live_parity_leg(args, model, image, boxes, prompt, device)
        except Exception as ex:                                      # never lose the headline
            live_parity = {"error": repr(ex)[:300]}
            torch.cuda.empty_cache()

    vit = None
    if rank == 0 and not args.no_roofline:
        try:
            vit = vit_roofline(model, args, device)
        except Exception as ex:
            vit = {"error": repr(ex)}
        if roofline is not None:
            roofline["vit"] = vit
    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        # extra legs, never part of `value`: (a) ONE batch-1 request, strictly serial (the per-request latency of configs[1]);
        # (b) the reference-native 224^2 resolution (its RoI extractor asserts a 16 x 16 grid, layers.py:289-291), same merge;
        # (c) the same merged launch sequence in the reference's SERVING dtype (fp16, app.py:74-98) when the headline is bf16
        extras = {}
        try:
            i1, b1, p1 = make_inputs(args, ids, device, seed=9000, batch=1)
            t1 = timed_replay(model, i1, b1, p1, steps=max(4, args.steps))
            extras["single_request"] = {"ms": round(1e3 * t1, 3), "region_tokens_per_s": round(args.rois / t1, 1),
                                        "what": "one batch-1 request, one stream, hipGraph replay (latency of configs[1])"}
            i2, b2, p2 = make_inputs(args, ids, device, seed=9001, image_size=224)
            t2 = timed_replay(model, i2, b2, p2, steps=max(4, args.steps // 2), streams=max(1, args.streams))
            extras["native_224"] = {"ms_per_step": round(1e3 * t2, 3), "requests_per_step": int(i2.size(0)),
                                    "region_tokens_per_s": round(args.rois * i2.size(0) / t2, 1), "prompt_tokens": int(p2.size(1)),
                                    "what": "the reference-native 224^2 image (P = 16), 32 RoIs, same merge and streams; "
                                            "position table of the 336^2 tower (first 257 rows), timing only"}
            del i1, b1, p1, i2, b2, p2
        except Exception as ex:
            extras["error"] = repr(ex)
        if args.mixed_tokens > 0:
            # (d) SURVEY.md 8d config 5 on ONE GPU: multi-region VCR-shape requests (224^2 crop, 64 RoIs), B of them served
            # together: batched vision + region module + prefill, then `mixed_tokens` greedy tokens each, one hipGraph replay per
            # decode step for the whole batch (tools/mixed_bench.py is the stand-alone form)
            try:
                from types import SimpleNamespace as _NS
                Bm, n_new = max(1, args.decode_batch), args.mixed_tokens
                a5 = _NS(**{**vars(args), "rois": 64})
                i5, b5, p5 = make_inputs(a5, ids, device, seed=9002, batch=Bm, image_size=224)
                boxes5 = model.prepare_boxes(b5, 224)

                def request(n):
                    emb5 = model.embed_inputs(p5, i5, boxes5)
                    return model.llama.decode_graph_batch(emb5, n)
                request(8)                                          # warm-up + decode-graph capture
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                request(2)
                torch.cuda.synchronize()
                t_pre = time.perf_counter() - t0
                t0 = time.perf_counter()
                outs5 = request(n_new)
                torch.cuda.synchronize()
                t_full = time.perf_counter() - t0
                extras["mixed_prefill_decode"] = {
                    "what": f"SURVEY.md 8d config 5 on one GPU: {Bm} requests of a 224^2 crop + 64 RoIs served together, prompt "
                            f"{int(p5.size(1))} tokens + {n_new} greedy tokens each (vision + region module + prefill, then one "
                            "hipGraph replay per decode step for the batch)",
                    "batch": Bm, "prefill_ms": round(1e3 * t_pre, 2), "batch_ms": round(1e3 * t_full, 2),
                    "decode_ms_per_step": round(1e3 * (t_full - t_pre) / max(1, n_new - 2), 3),
                    "generated_tokens_per_s": round(Bm * n_new / t_full, 1), "region_tokens_per_s": round(Bm * 64 / t_full, 1),
                    "requests_per_s": round(Bm / t_full, 3), "tokens_out": [len(o) for o in outs5][:2]}
                del i5, b5, p5, boxes5, outs5
            except Exception as ex:
                extras["mixed_prefill_decode"] = {"error": repr(ex)}
    train = None
    hard_exit = False                                    # a rank stuck in a collective of the training leg (see with_deadline)
    n_ctx, graph_ok = len(ctxs), all(g is not None for g in graphs)
    # the OTHER storage type of the same configuration (extras) and the training leg.  The reference trains in bf16
    # (train_stage1.sh:19) and serves in fp16 (app.py:74-98): with the fp16 headline the bf16 model built for the extras leg is
    # the one the training leg runs on (every rank builds it when the training leg is on).
    other_name = "bf16" if args.dtype == "fp16" else "fp16"
    other_dtype = torch.bfloat16 if other_name == "bf16" else torch.float16
    want_other_leg = rank == 0 and extras is not None and world == 1
    want_train = args.train_steps > 0
    del graphs, ctxs, reqs                               # release the captured inference pools
    graph_error = last.get("graph_error")
    last.clear()
    last["graph_error"] = graph_error
    torch.cuda.empty_cache()
    m_train = model if args.dtype == "bf16" else None
    want_stage2 = want_other_leg and args.stage2_steps > 0
    if want_train and m_train is not None:
        try:
            leg = lambda: train_leg(args, m_train, ids, device, rank, world, dist, device if backend == "nccl" else "cpu")   # noqa: E731
            train = leg() if world == 1 else with_deadline(leg, args.train_deadline, local)
        except _Deadline as ex:
            train, hard_exit = {"error": f"training leg: {ex}"}, True
        except Exception as ex:                                      # never lose the headline: every rank carries on to the
            train = {"error": repr(ex)}                              # end and exits 0 (no collective follows this leg); a
            #                                                          rank left waiting in one gets the group's timeout here
        if want_stage2:
            try:
                extras["stage2_step"] = stage2_leg(args, m_train, ids, device)
            except Exception as ex:
                extras["stage2_step"] = {"error": repr(ex)}
    if want_other_leg or (want_train and m_train is None):
        try:
            del model, m_train
            torch.cuda.empty_cache()
            m2, ids2 = build_model(args, device, seed=100 + rank, dtype=other_dtype)
            if want_other_leg:
                i3, b3, p3 = make_inputs(args, ids2, device, seed=rank)
                t3 = timed_replay(m2, i3, b3, p3, steps=max(4, args.steps // 2), streams=max(1, args.streams))
                extras[other_name] = {"ms_per_step": round(1e3 * t3, 3), "region_tokens_per_s": round(args.rois * i3.size(0) / t3, 1),
                                      "what": f"the headline configuration with {other_name} storage (bf16 = the reference's training "
                                              "dtype, train_stage1.sh:19; fp16 = its serving dtype, app.py:74-98): the other "
                                              "instantiation of the same kernels"}
                del i3, b3, p3
                torch.cuda.empty_cache()
        except Exception as ex:
            m2 = None
            if extras is not None:
                extras[other_name] = {"error": repr(ex)}
        if want_train and other_name == "bf16":
            try:
                if m2 is None:
                    raise RuntimeError("the bf16 model of the training leg could not be built")
                leg = lambda: train_leg(args, m2, ids2, device, rank, world, dist, device if backend == "nccl" else "cpu")   # noqa: E731
                train = leg() if world == 1 else with_deadline(leg, args.train_deadline, local)
            except _Deadline as ex:
                train, hard_exit = {"error": f"training leg: {ex}"}, True
            except Exception as ex:
                train = {"error": repr(ex)}
        if want_stage2 and other_name == "bf16" and m2 is not None:
            try:
                extras["stage2_step"] = stage2_leg(args, m2, ids2, device)
            except Exception as ex:
                extras["stage2_step"] = {"error": repr(ex)}
        m2 = None
        torch.cuda.empty_cache()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args)
        except Exception as ex:                                      # never lose the GPU number
            cpu = {"error": repr(ex)}

    greedy = None
    if rank == 0:
        # exact-id statistics of the committed GPU test for this storage type: written by
        # tests/test_fullwidth_gpu.py on the GPU box, merged into profiles/rNN_greedy_parity.json
        try:
            import glob
            import re as _re
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_greedy_parity.json")),
                           key=lambda f: int(_re.search(r"r(\d+)_", os.path.basename(f)).group(1)))
            if cands:
                runs = json.load(open(cands[-1]))["runs"]

                def summary(dt):
                    rs = [r for r in runs if r["dtype"] == dt]
                    if not rs:
                        return None
                    n = rs[0]["new_tokens"]
                    return {"seeds": [r["seed"] for r in rs], "new_tokens": n,
                            "exact": all(r["whole_path_generate_exact_len"] == n and r["teacher_forced_identical"] == n for r in rs),
                            "whole_path_exact_len": [r["whole_path_generate_exact_len"] for r in rs],
                            "decoder_free_running_exact_len": [r["free_running_exact_len"] for r in rs],
                            "teacher_forced_identical": [r["teacher_forced_identical"] for r in rs],
                            "first_divergence": [r["whole_path_first_divergence"] for r in rs if r["whole_path_first_divergence"]] or None}
                import hashlib
                doc = json.load(open(cands[-1]))
                greedy = {"source": os.path.relpath(cands[-1], ROOT),
                          "source_sha256_16": hashlib.sha256(open(cands[-1], "rb").read()).hexdigest()[:16],
                          "quoted": True, "scope": "batch1 (one request at a time): tests/test_fullwidth_gpu.py::test_bench_model_full_depth_...",
                          "against": runs[0]["against"],
                          args.dtype: summary(args.dtype), ("bf16" if args.dtype == "fp16" else "fp16"): summary("bf16" if args.dtype == "fp16" else "fp16")}
                m16 = doc.get("merged16")
                if m16:      # tests/test_fullwidth_gpu.py::test_sixteen_merged_requests_... (whole path incl. vision + region stages)
                    greedy["merged16_" + m16.get("dtype", "fp16")] = {k: m16[k] for k in (
                        "requests", "new_tokens", "merged_exact_len", "alone_exact_len", "merged_equals_alone_len",
                        "max_logit_err_of_range", "max_merged_vs_alone_logits_rel") if k in m16}
        except Exception as ex:
            greedy = {"error": repr(ex)}
    if rank == 0:
        P = args.image_size // 14
        total_regions = args.rois * args.batch * args.steps * world
        # forward FLOPs of one request: ViT 23 blocks + patch embed, 1x1 + 5 fuse rounds of 3x3 convs over the
        # 85 P^2 pyramid pixels, pconvs + flatten_linear + updims per RoI, projector, LLaMA prefill (12.95 GF/token incl. lm_head
        # + causal attention)
        flops_per_request = forward_flops_per_request(P, args.rois, int(prompt.size(1)), args.llama_layers)
        line = {
            "metric": f"region-tokens/sec (336^2 img, 32 RoIs, CLIP ViT-L/14 + region module + LLaMA-7B fwd; {args.batch} batch-1 requests "
                      "merged per launch sequence, value_batch1 = one request at a time)",
            "value": round(total_regions / dt, 2), "unit": "region-tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "value_per_gpu": round(total_regions / dt / world, 2),
            "ranks": world, "backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
            # configs[1] is a batch-1 request; `value` is the throughput of a server that merges `requests_per_step` of them into
            # one launch sequence (the weights stream once for all); the strictly serial batch-1 figure of the same run:
            "value_batch1": (extras or {}).get("single_request", {}).get("region_tokens_per_s"),
            "in_flight_requests_per_gpu": n_ctx, "hipMalloc_calls_in_timed_region": device_allocs_in_timed_region,
            "hipgraph": graph_ok, "hipgraph_error": last.get("graph_error"),
            "requests_per_step": args.batch,
            "single_stream": {"ms_per_image": round(1e3 * dt_serial / args.steps / args.batch, 3),
                              "region_tokens_per_s_per_gpu": round(args.rois * args.batch * args.steps / dt_serial, 2)},
            "config": {"workload": f"configs[1]: batch-1 requests of 1x{args.image_size}^2 image + {args.rois} RoIs, {args.batch} of them "
                                   f"merged per launch sequence: ViT-L/14(23 blocks) + SPI(P={P}) + LLaMA-7B({args.llama_layers} "
                                   f"layers) prefill T={prompt.size(1)} per request with full logits",
                       "image_size": args.image_size, "rois_per_image": args.rois, "prompt_tokens": int(prompt.size(1)),
                       "parallelism": f"replicas x{world} (no data-path collective); {n_ctx} launch sequences of {args.batch} "
                                      f"requests in flight per GPU on separate HIP streams",
                       "valid": args.llama_layers == 32 and args.image_size == 336 and args.rois == 32,
                       # why `value` is where it is against north_star's 5000 (SURVEY.md section 7 "hard parts")
                       "target_arithmetic": {
                           "north_star_target_region_tokens_per_s": 5000,
                           "forward_TFLOP_per_request": round(flops_per_request / 1e12, 2),
                           "PFLOP_per_s_needed_at_target": round(5000 / args.rois * flops_per_request / 1e15, 3),
                           "fraction_of_dense_16bit_peak_needed": round(5000 / args.rois * flops_per_request / 1e12 / PEAK_BF16_TFLOPS, 3),
                           "this_run_PFLOP_per_s": round(total_regions / dt / args.rois * flops_per_request / 1e15, 3),
                           "this_run_fraction_of_peak": round(total_regions / dt / args.rois * flops_per_request / 1e12 / PEAK_BF16_TFLOPS, 3),
                           "value_if_every_kernel_ran_at_the_vendor_gemm_fraction_0p62": round(0.62 * PEAK_BF16_TFLOPS * 1e12 / flops_per_request * args.rois, 0),
                           "note": "5000 region-tokens/s at 336^2 needs ~99 % of the dense MFMA peak for the whole forward; at the "
                                   "reference-native 224^2 it needs ~59 % (extras.native_224 is that configuration)"},
                       "headline_defaults": {"dtype": "fp16 since round 5 (bf16 rounds 1-4; the other type is extras." + other_name + ")",
                                             "requests_merged": "16 since round 4 (1 x 2 streams rounds 1-3, 4 x 2 early round 4)"}},
            # greedy ids against HF fp32: MEASURED in this run for the merged dispatch `value` is timed on (live_parity_leg) when
            # that leg ran; otherwise the quotation of the committed test record, whose scope is the single request
            "greedy_exact": (live_parity["exact"] if isinstance(live_parity, dict) and "exact" in live_parity else
                             ((greedy or {}).get(args.dtype, {}).get("exact") if isinstance((greedy or {}).get(args.dtype), dict) else None)),
            "greedy_exact_scope": (f"merged{args.batch}, measured in this run (decoder: merged prefill + batched decode, "
                                   f"{args.parity_tokens} tokens x {args.batch} requests vs HF fp32)"
                                   if isinstance(live_parity, dict) and "exact" in live_parity else
                                   ("batch1, quoted from " + str((greedy or {}).get("source")) + " sha256:" + str((greedy or {}).get("source_sha256_16"))
                                    if greedy and "source" in greedy else None)),
            "greedy_exact_len": live_parity.get("exact_len") if isinstance(live_parity, dict) else None,
            "greedy_live": live_parity,
            "greedy_parity": greedy,
            "train_leg_timed_out": bool(hard_exit),
            "roofline": roofline, "cpu_baseline": cpu, "decode": decode, "train": train, "extras": extras, "kernels": kernels,
        }
        print(json.dumps(line), flush=True)
    if hard_exit:
        sys.stdout.flush()
        os._exit(0)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
