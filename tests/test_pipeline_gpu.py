"""GPU: stage-by-stage and end-to-end parity of the HIP pipeline (gpt4roi_amd/{vit,layers,llama,
spi_llava}.py, everything through the C ABI) against the CPU oracles on the same seeded inputs.

Tolerances: the pipeline stores bf16 between kernels, the reference semantics are fp32.
  * vs the oracle run with bf16 rounding at the pipeline's storage points ("emulate"): tight --
    this is the check that catches kernel bugs;
  * vs the pure-fp32 oracle / the reference-code fixture: loose (bf16 noise through ~40 layers).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import spi_oracle as S  # noqa: E402
from oracle import transformer_oracle as T  # noqa: E402

if torch.cuda.is_available():
    from gpt4roi_amd import kernels as K
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.layers import MLVLROIQueryModule
    from gpt4roi_amd.llama import LlamaDecoder
    from gpt4roi_amd.spi_llava import SPILlavaLlamaModel, SPILlavaMPTForCausalLM
    from gpt4roi_amd.vit import ClipVisionTower

DEV = "cuda"


@pytest.fixture(autouse=True)
def _inference_mode():
    """This file checks the INFERENCE path (app.py runs it under torch.inference_mode, app.py:285); with grad enabled
    the same calls would take the autograd seam (tests/test_autograd_gpu.py)."""
    with torch.no_grad():
        yield


def relerr(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-6)).item()


def bf(x):
    return x.to(torch.bfloat16).float()


# ------------------------------------------------------------------------------------------ region module
def test_spi_module_against_reference_code_fixture(golden_dir):
    """fixture = output of the reference's own gpt4roi/models/layers.py (fp32, 224^2, embed 512)."""
    z = np.load(os.path.join(golden_dir, "spi_module_ref_c512.npz"))
    C, B, P = int(z["embed_dims"]), int(z["B"]), int(z["P"])
    m = MLVLROIQueryModule(embed_dims=C, out_dims=4096, num_levels=4)
    m.load_state_dict(S.synthetic_state(m, int(z["wseed"])))
    m.to(DEV)
    feats, boxes = S.synthetic_inputs(int(z["iseed"]), B, P, C, [int(n) for n in z["n_rois"]])
    out = torch.cat(m([f.to(DEV) for f in feats], [b.to(DEV) for b in boxes]), 0)
    want = torch.from_numpy(z["out"])
    e = relerr(out, want)
    print("spi module vs reference-code fixture: rel-to-max err", e)
    assert e < 4e-2, e
    # and against the restated oracle with bf16 rounding points (same weights, bf16-rounded inputs)
    o = S.MLVLROIQueryOracle(embed_dims=C, P=P)
    o.load_state_dict(S.synthetic_state(o, int(z["wseed"])))
    with torch.no_grad():
        emu = torch.cat(o([bf(f) for f in feats], boxes, emulate=True), 0)
    e2 = relerr(out, emu)
    print("spi module vs bf16-emulating oracle: rel-to-max err", e2)
    assert e2 < 1.5e-2, e2


def test_spi_module_intermediates_p8():
    """Smaller grid (P=8), two images, per-stage comparison against the emulating oracle."""
    C, P, B = 512, 8, 2
    m = MLVLROIQueryModule(embed_dims=C, out_dims=512, num_levels=4)
    o = S.MLVLROIQueryOracle(embed_dims=C, P=P)
    o.roi_align.updims = torch.nn.Linear(1024, 512)
    sd = S.synthetic_state(o, 5)
    o.load_state_dict(sd)
    m.load_state_dict(sd)
    m.to(DEV)
    feats, boxes = S.synthetic_inputs(6, B, P, C, [4, 1])
    with torch.no_grad():
        want, inter = o([bf(f) for f in feats], boxes, emulate=True, return_intermediates=True)
    toks = [f.to(DEV).to(torch.bfloat16) for f in feats]
    sizes = [P * 2 ** l for l in range(4)][::-1]
    maps, affs = m.mlvl_fuse(toks, P, sizes)
    for l in range(4):
        y = torch.relu(maps[l].float() * affs[l][:, 0][:, None, None, :] + affs[l][:, 1][:, None, None, :])
        e = relerr(y.permute(0, 3, 1, 2), inter["fused"][l])
        assert e < 2e-2, (l, e)
    out = m(toks, [b.to(DEV) for b in boxes])
    assert [x.shape for x in out] == [(4, 512), (1, 512)]
    assert relerr(torch.cat(out), torch.cat(want)) < 1.5e-2
    # no regions in any image -> empty outputs, no launch (layers.py:314-317 keeps the graph alive)
    empty = m(toks, [torch.zeros(0, 4, device=DEV), torch.zeros(0, 4, device=DEV)])
    assert [x.shape for x in empty] == [(0, 512), (0, 512)]


# ------------------------------------------------------------------------------------------ ViT
def _mini_vit(hidden=256, heads=4, layers=12, image=112):
    sd = syn.vit_state(hidden, 4 * hidden, layers, image, seed=3)
    return sd, ClipVisionTower(sd, heads=heads, device=DEV)


def test_vit_hidden_states():
    sd, tower = _mini_vit()
    assert tower.level_indices == [2, 5, 8, 11] and tower.image_feature_index == 11 and len(tower.layers) == 11
    img = torch.randn(2, 3, 112, 112, generator=torch.Generator().manual_seed(4))
    keep = tower.forward(img.to(DEV))
    sdb = {k: bf(v) for k, v in sd.items()}
    emu = T.clip_vit_hidden_states(sdb, img, heads=4, n_layers=11, emulate=True)
    ref = T.clip_vit_hidden_states(sd, img, heads=4, n_layers=11)
    for i in tower.level_indices:
        e_emu, e_ref = relerr(keep[i], emu[i]), relerr(keep[i], ref[i])
        print(f"vit hs[{i}] vs emulate {e_emu:.4f} vs fp32 {e_ref:.4f}")
        assert e_emu < 2e-2 and e_ref < 6e-2
    img_feat, lv = tower.select(keep)
    assert img_feat.shape == (2, 64, 256) and len(lv) == 4


# ------------------------------------------------------------------------------------------ LLaMA
def _mini_llama(hidden=512, heads=4, inter=1408, layers=4, vocab=1000):
    sd = syn.llama_state(hidden, inter, layers, vocab, seed=5)
    return sd, LlamaDecoder(sd, heads=heads, max_positions=256, device=DEV)


def test_llama_prefill_decode_and_greedy_ids():
    sd, dec = _mini_llama()
    sdb = {k: bf(v) for k, v in sd.items()}
    ids = torch.randint(0, 1000, (1, 37), generator=torch.Generator().manual_seed(6))
    emb = bf(sd["model.embed_tokens.weight"])[ids]
    dec.reset(1)
    logits = dec.forward(emb.to(DEV).to(torch.bfloat16))
    h, cache = T.llama_forward(sdb, emb, heads=4, emulate=True)
    want = T.lm_logits(sdb, h, emulate=True)
    e = relerr(logits, want)
    print("llama prefill logits vs emulate", e)
    assert logits.shape == (1, 37, 1000) and e < 2e-2
    # cached single-token step
    nxt = torch.tensor([[123]])
    l2 = dec.forward(bf(sd["model.embed_tokens.weight"])[nxt].to(DEV).to(torch.bfloat16))
    h2, _ = T.llama_forward(sdb, bf(sd["model.embed_tokens.weight"])[nxt], heads=4, kv_cache=cache, pos0=37, emulate=True)
    assert relerr(l2, T.lm_logits(sdb, h2, emulate=True)) < 2e-2
    # greedy decode: identical token ids (north_star) -- margins reported when a near-tie flips
    got = dec.greedy(emb.to(DEV).to(torch.bfloat16), 12)
    embed_fn = lambda t: bf(sd["model.embed_tokens.weight"])[t]
    want_ids, trace = T.greedy_decode(sdb, emb, embed_fn, heads=4, n_new=12, emulate=True)
    if got != want_ids:
        k = next(i for i, (a, b) in enumerate(zip(got, want_ids)) if a != b)
        top2 = trace[k].topk(2).values
        pytest.fail(f"greedy ids diverge at step {k}: {got} vs {want_ids}; oracle top-2 margin {float(top2[0]-top2[1]):.2e}")


def test_device_resident_greedy_decode_matches_host_loop():
    """greedy_graph (token id / position on the device, one hipGraph replay per token) == greedy (host loop)."""
    sd, dec = _mini_llama()
    emb = bf(sd["model.embed_tokens.weight"])[torch.randint(0, 1000, (1, 21), generator=torch.Generator().manual_seed(16))]
    emb = emb.to(DEV).to(torch.bfloat16)
    want = dec.greedy(emb, 20)
    assert dec.greedy_graph(emb, 20, use_graph=False) == want
    assert dec.greedy_graph(emb, 20) == want
    assert dec.greedy_graph(emb, 20) == want                      # second call reuses the captured graph
    stop = want[7]
    first = want.index(stop)
    assert dec.greedy_graph(emb, 20, stop_ids=(stop,), check_every=4) == want[:first + 1]


def test_llama_batched_greedy_decode():
    """greedy_batch (one decoder pass per step for B sequences): every token it emits must be the argmax -- up to
    bf16 summation-order noise -- of the single-sequence logits for the same prefix (teacher-forced check, robust
    against near-tie flips between the M = 1 and M = B kernels), and stop ids cut the rows."""
    sd, dec = _mini_llama(layers=2)
    B, T0, n_new = 3, 19, 6
    g = torch.Generator().manual_seed(70)
    emb_w = bf(sd["model.embed_tokens.weight"])
    prompts = emb_w[torch.randint(0, 1000, (B, T0), generator=g)].to(DEV).to(torch.bfloat16)
    got = dec.greedy_batch(prompts, n_new)
    assert len(got) == B and all(len(r) == n_new for r in got)
    same = 0
    for b in range(B):
        toks = torch.tensor(got[b][:-1], dtype=torch.int64)
        prefix = torch.cat([prompts[b:b + 1], emb_w[toks][None].to(DEV).to(torch.bfloat16)], 1)
        dec.reset(1)
        logits = dec.forward(prefix)[0, T0 - 1:]                       # predictions for the n_new generated positions
        top = logits.max(-1).values
        chosen = logits[torch.arange(n_new, device=DEV), torch.tensor(got[b], device=DEV)]
        assert float((top - chosen).max()) <= 2e-2 * float(logits.abs().max()), (b, (top - chosen).tolist())
        same += int((logits.argmax(-1).cpu() == torch.tensor(got[b])).sum())
    assert same >= int(0.8 * B * n_new)
    stop = got[1][2]
    cut = dec.greedy_batch(prompts, n_new, stop_ids=(stop,))
    assert cut[1] == got[1][:got[1].index(stop) + 1] and all(len(r) <= n_new for r in cut)


def test_llama_batch2_matches_batch1():
    sd, dec = _mini_llama(layers=2)
    dec.reset(2)
    emb = torch.randn(2, 20, 512, generator=torch.Generator().manual_seed(7)).to(DEV).to(torch.bfloat16)
    both = dec.forward(emb)
    dec.reset(1)
    one = dec.forward(emb[1:2])
    assert relerr(both[1], one[0]) < 1e-3


# ------------------------------------------------------------------------------------------ end to end
def test_end_to_end_embeds_logits_and_greedy_ids():
    """ViT -> levels -> region module -> projector -> splice -> LLaMA, mini widths, P = 8 (112^2),
    forward(input_ids, images, bboxes) signature of spi_llava.py:23-36."""
    H, P, image = 512, 8, 112
    ids = syn.token_ids(vocab_base=990)
    vsd = syn.vit_state(H, 4 * H, 12, image, seed=8)
    lsd = syn.llama_state(512, 1408, 3, ids.vocab, seed=9)
    tower = ClipVisionTower(vsd, heads=8, device=DEV)
    dec = LlamaDecoder(lsd, heads=4, max_positions=256, device=DEV)
    model = SPILlavaLlamaModel(tower, dec, ids, embed_dims=H)
    orc = S.MLVLROIQueryOracle(embed_dims=H, P=P)
    orc.roi_align.updims = torch.nn.Linear(1024, 512)
    spi_sd = S.synthetic_state(orc, 10)
    orc.load_state_dict(spi_sd)
    model.spi_module.load_state_dict(spi_sd)
    g = torch.Generator().manual_seed(11)
    pw, pb = torch.randn(512, H, generator=g) / H ** 0.5, torch.randn(512, generator=g) * 0.05
    with torch.no_grad():
        model.mm_projector.weight.copy_(pw)
        model.mm_projector.bias.copy_(pb)
    lm = SPILlavaMPTForCausalLM(model)
    img = torch.randn(1, 3, image, image, generator=g)
    boxes = [syn.boxes(3, g)]
    prompt = syn.prompt_ids(ids, P, 3, g, sys_len=6, question_len=5, vocab_base=990)[None]
    out = lm(input_ids=prompt.to(DEV), images=img.to(DEV), bboxes=[b.to(DEV) for b in boxes])
    model.check_status()
    # ---- oracle pipeline (bf16 rounding points) ----
    vb = {k: bf(v) for k, v in vsd.items()}
    lb = {k: bf(v) for k, v in lsd.items()}
    hs = T.clip_vit_hidden_states(vb, img, heads=8, n_layers=11, emulate=True)
    img_feat, lv = T.select_spi_levels(hs + [hs[-1]], -2, 4)   # pad: oracle ran 11 of the 12 layers
    with torch.no_grad():
        spi = orc(lv, boxes, emulate=True)
    proj = bf(bf(img_feat) @ bf(pw).t() + bf(pb))
    emb = bf(lsd["model.embed_tokens.weight"])[prompt]
    spliced = S.splice(prompt, emb, proj, spi, ids.im_start_token, ids.im_end_token, ids.bbox_token)
    got_emb = model.embed_inputs(prompt.to(DEV), img.to(DEV), [b.to(DEV) for b in boxes])
    e_emb = relerr(got_emb, spliced)
    print("inputs_embeds vs oracle", e_emb)
    assert e_emb < 2e-2
    h, _ = T.llama_forward(lb, spliced, heads=4, emulate=True)
    want = T.lm_logits(lb, h, emulate=True)
    e_log = relerr(out.logits, want)
    print("end-to-end logits vs oracle", e_log)
    assert out.logits.shape == (1, prompt.size(1), ids.vocab) and e_log < 3e-2
    # greedy decode from the SAME spliced embeddings on both sides
    got_ids = dec.greedy(spliced.to(DEV).to(torch.bfloat16), 8)
    want_ids, trace = T.greedy_decode(lb, spliced, lambda t: bf(lsd["model.embed_tokens.weight"])[t], heads=4,
                                      n_new=8, emulate=True)
    if got_ids != want_ids:
        k = next(i for i, (a, b) in enumerate(zip(got_ids, want_ids)) if a != b)
        top2 = trace[k].topk(2).values
        pytest.fail(f"greedy ids diverge at step {k}: {got_ids} vs {want_ids}; margin {float(top2[0]-top2[1]):.2e}")
    gen = lm.generate(prompt.to(DEV), images=img.to(DEV), bboxes=[b.to(DEV) for b in boxes], max_new_tokens=4)
    assert gen.shape == (1, prompt.size(1) + 4) and torch.equal(gen[:, :prompt.size(1)].cpu(), prompt)


def test_baseline_config0_full_size_vit_region_module_projector():
    """BASELINE.json configs[0]: ONE 336x336 image, 4 random boxes, full-width CLIP ViT-L/14 (23 blocks) + region module
    (C = 1024, P = 24) + projector, no LLaMA -- the HIP path against the CPU oracles at the real sizes."""
    H, P, image, heads = 1024, 24, 336, 16
    vsd = syn.vit_state(H, 4 * H, 24, image, seed=31)
    tower = ClipVisionTower(vsd, heads=heads, device=DEV)
    m = MLVLROIQueryModule(embed_dims=H, out_dims=4096, num_levels=4)
    orc = S.MLVLROIQueryOracle(embed_dims=H, P=P)
    sd = S.synthetic_state(orc, 32)
    orc.load_state_dict(sd)
    m.load_state_dict(sd)
    m.to(DEV)
    g = torch.Generator().manual_seed(33)
    img = torch.randn(1, 3, image, image, generator=g)
    boxes = [syn.boxes(4, g)]
    pw, pb = torch.randn(4096, H, generator=g) / H ** 0.5, torch.randn(4096, generator=g) * 0.05
    keep = tower.forward(img.to(DEV))
    feat, lv = tower.select(keep)
    got_spi = torch.cat(m(lv, [b.to(DEV) for b in boxes]), 0)
    got_proj = K.gemm(feat[0], pw.to(DEV).to(torch.bfloat16), bias=bf(pb).to(DEV))
    vb = {k: bf(v) for k, v in vsd.items()}
    with torch.no_grad():
        hs = T.clip_vit_hidden_states(vb, img, heads=heads, n_layers=23, emulate=True)
        img_feat, olv = T.select_spi_levels(hs + [hs[-1]], -2, 4)
        want_spi = torch.cat(orc(olv, boxes, emulate=True), 0)
        want_proj = bf(bf(img_feat[0]) @ bf(pw).t() + bf(pb))
    e_lv = max(relerr(a[0], b[0]) for a, b in zip(lv, olv))
    e_spi, e_proj = relerr(got_spi, want_spi), relerr(got_proj, want_proj)
    print("configs[0] at full size: ViT levels", e_lv, "region tokens", e_spi, "patch tokens", e_proj)
    assert got_spi.shape == (4, 4096) and got_proj.shape == (P * P, 4096)
    assert e_lv < 3e-2 and e_spi < 3e-2 and e_proj < 3e-2


def test_end_to_end_batch_of_two_ragged_regions():
    """B = 2 images with different numbers of regions (one has none): the batched call must equal
    the two single-image calls (every stage of the path is per-image)."""
    H, P, image = 512, 8, 112
    ids = syn.token_ids(vocab_base=990)
    tower = ClipVisionTower(syn.vit_state(H, 4 * H, 12, image, seed=8), heads=8, device=DEV)
    dec = LlamaDecoder(syn.llama_state(512, 1408, 2, ids.vocab, seed=9), heads=4, max_positions=256, device=DEV,
                       max_batch=2)
    model = SPILlavaLlamaModel(tower, dec, ids, embed_dims=H)
    model.spi_module.load_state_dict(syn.spi_state(model.spi_module, 3))
    g = torch.Generator().manual_seed(13)
    imgs = torch.randn(2, 3, image, image, generator=g).to(DEV)
    boxes = [syn.boxes(2, g).to(DEV), torch.zeros(0, 4, device=DEV)]
    p0 = syn.prompt_ids(ids, P, 2, g, sys_len=4, question_len=6, vocab_base=990)
    p1 = syn.prompt_ids(ids, P, 0, g, sys_len=4, question_len=6 + 8, vocab_base=990)   # same length, no regions
    assert p0.numel() == p1.numel()
    prompts = torch.stack([p0, p1]).to(DEV)
    both = model(input_ids=prompts, images=imgs, bboxes=boxes)
    model.check_status()
    for b in range(2):
        one = model(input_ids=prompts[b:b + 1], images=imgs[b:b + 1], bboxes=[boxes[b]])
        # not bit-equal: B changes the GEMM M, hence the tile / split-K choice and the fp32 summation order
        assert relerr(both[b], one[0]) < 1.5e-2, b


def test_two_request_contexts_on_two_streams_are_bit_identical():
    """clone_context(): shared weights, private KV cache; interleaving two requests on two HIP streams
    must not change either result (bench.py's default mode)."""
    H, P, image = 512, 8, 112
    ids = syn.token_ids(vocab_base=990)
    tower = ClipVisionTower(syn.vit_state(H, 4 * H, 12, image, seed=8), heads=8, device=DEV)
    dec = LlamaDecoder(syn.llama_state(512, 1408, 2, ids.vocab, seed=9), heads=4, max_positions=256, device=DEV)
    a = SPILlavaLlamaModel(tower, dec, ids, embed_dims=H)
    a.spi_module.load_state_dict(syn.spi_state(a.spi_module, 3))
    a.prepare()
    b = a.clone_context()
    assert b.llama.kc.data_ptr() != a.llama.kc.data_ptr() and b.llama.layers is a.llama.layers
    g = torch.Generator().manual_seed(14)
    reqs = []
    for n in (3, 1):
        reqs.append((syn.prompt_ids(ids, P, n, g, sys_len=4, question_len=6 + 4 * (3 - n), vocab_base=990)[None].to(DEV),
                     torch.randn(1, 3, image, image, generator=g).to(DEV), [syn.boxes(n, g).to(DEV)]))
    want = [a(input_ids=p, images=im, bboxes=bx).clone() for p, im, bx in reqs]
    torch.cuda.synchronize()
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    got = [None, None]
    for rep in range(3):
        with torch.cuda.stream(s0):
            got[0] = a(input_ids=reqs[0][0], images=reqs[0][1], bboxes=reqs[0][2])
        with torch.cuda.stream(s1):
            got[1] = b(input_ids=reqs[1][0], images=reqs[1][1], bboxes=reqs[1][2])
    torch.cuda.synchronize()
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_forward_is_hipgraph_capturable_with_prepared_boxes():
    """With PreparedBoxes the launch sequence has no host<->device traffic: capture it once, replay it
    with new image / token contents in the static input buffers, and get the eager result."""
    H, P, image = 512, 8, 112
    ids = syn.token_ids(vocab_base=990)
    tower = ClipVisionTower(syn.vit_state(H, 4 * H, 12, image, seed=8), heads=8, device=DEV)
    dec = LlamaDecoder(syn.llama_state(512, 1408, 2, ids.vocab, seed=9), heads=4, max_positions=256, device=DEV)
    m = SPILlavaLlamaModel(tower, dec, ids, embed_dims=H)
    m.spi_module.load_state_dict(syn.spi_state(m.spi_module, 3))
    g = torch.Generator().manual_seed(15)
    img = torch.randn(1, 3, image, image, generator=g).to(DEV)
    prompt = syn.prompt_ids(ids, P, 2, g, sys_len=4, question_len=5, vocab_base=990)[None].to(DEV)
    req = m.prepare_boxes([syn.boxes(2, g)], image)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        m(input_ids=prompt, images=img, bboxes=req)
        m(input_ids=prompt, images=img, bboxes=req)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        out = m(input_ids=prompt, images=img, bboxes=req)
    # new request contents in the same buffers
    img.copy_(torch.randn(1, 3, image, image, generator=g))
    prompt[0, -3:] = torch.tensor([17, 23, 5], device=DEV)
    with torch.cuda.stream(st):
        graph.replay()
    torch.cuda.synchronize()
    want = m(input_ids=prompt, images=img, bboxes=req)
    assert torch.equal(out, want)
    m.check_status()


def test_malformed_prompt_raises_like_the_reference():
    ids = syn.token_ids(vocab_base=990)
    vsd = syn.vit_state(256, 1024, 12, 112, seed=8)
    lsd = syn.llama_state(512, 1408, 1, ids.vocab, seed=9)
    model = SPILlavaLlamaModel(ClipVisionTower(vsd, heads=4, device=DEV), LlamaDecoder(lsd, heads=4, device=DEV,
                               max_positions=128), ids, embed_dims=256)
    g = torch.Generator().manual_seed(12)
    prompt = syn.prompt_ids(ids, 8, 1, g, sys_len=3, question_len=2, vocab_base=990)[None].clone()
    prompt[0, (prompt[0] == ids.im_end_token).nonzero()[0, 0]] = 5       # drop <im_end>
    img = torch.randn(1, 3, 112, 112, generator=g).to(DEV)
    model.embed_inputs(prompt.to(DEV), img, None)
    with pytest.raises(ValueError):
        model.check_status()


def test_batched_device_resident_decode_matches_host_loop():
    """decode_graph_batch (one hipGraph replay per step for B sequences, ids / position on the device, one attention
    launch for the batch with RoPE + cache append fused) emits exactly the ids of greedy_batch, the host-loop form over
    the same kernels; a second call reuses the captured graph; stop ids cut the rows."""
    sd, dec = _mini_llama(layers=2)
    B, T0, n_new = 3, 19, 9
    g = torch.Generator().manual_seed(71)
    emb_w = bf(sd["model.embed_tokens.weight"])
    prompts = emb_w[torch.randint(0, 1000, (B, T0), generator=g)].to(DEV).to(torch.bfloat16)
    want = dec.greedy_batch(prompts, n_new)
    got = dec.decode_graph_batch(prompts, n_new)
    assert got == want
    assert dec._bstate[(B, False)]["graph"] is not None        # the state pool is keyed by (batch, ragged)
    again = dec.decode_graph_batch(prompts, n_new)
    assert again == want
    short = dec.decode_graph_batch(prompts, 4)
    assert short == [r[:4] for r in want]
    eager = dec.decode_graph_batch(prompts, n_new, use_graph=False)
    assert eager == want
    stop = want[2][3]
    cut = dec.decode_graph_batch(prompts, n_new, stop_ids=(stop,))
    assert cut[2] == want[2][:want[2].index(stop) + 1]
    # the single-sequence device loop is not disturbed by the batch state (separate workspaces / graphs)
    one = dec.decode_graph(prompts[1:2], n_new)
    assert one == dec.greedy(prompts[1:2], n_new)
