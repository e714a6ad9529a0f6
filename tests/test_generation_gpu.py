"""GPU: generation glue of the serving caller (gpt4roi/app.py:285-301) -- the device-side sampler
against oracle/sampler_oracle.py (same uniform draws -> identical ids), HF-style generate() with a partial-bound
forward, stopping criteria, and HF-directory loading (8f-2)."""
import os
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sampler_oracle as SO  # noqa: E402

if torch.cuda.is_available():
    from gpt4roi_amd import checkpoint as ckpt
    from gpt4roi_amd import kernels as K
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.llama import LlamaDecoder
    from gpt4roi_amd.spi_llava import KeywordsStoppingCriteria, SPILlavaLlamaModel, SPILlavaMPTForCausalLM
    from gpt4roi_amd.vit import ClipVisionTower

DEV = "cuda"


def _state(n=64):
    return dict(tok=torch.zeros((1, 1), dtype=torch.int64, device=DEV), out=torch.zeros(n, dtype=torch.int64, device=DEV),
                step=torch.zeros(1, dtype=torch.int32, device=DEV), pos=torch.zeros(1, dtype=torch.int32, device=DEV),
                seed=torch.zeros(1, dtype=torch.int64, device=DEV), u=torch.zeros(n, dtype=torch.float32, device=DEV))


@pytest.mark.parametrize("N,T,k,p", [(32006, 0.2, 50, 1.0), (32006, 1.0, 50, 0.9), (1000, 0.7, 5, 1.0), (1000, 1.0, 0, 1.0),
                                     (300, 1.3, 1024, 0.5), (257, 1.0, 1, 1.0)])
def test_sample_advance_matches_the_oracle_draw_for_draw(N, T, k, p):
    g = torch.Generator().manual_seed(N + k)
    st = _state()
    seed = 0x1234_5678_9ABC + k
    st["seed"].fill_(seed)
    rows = [(torch.randn(N, generator=g) * 3).float() for _ in range(24)]
    if p == 1.0:
        rows[3][:] = 0.25                                               # all ties: every token kept (list overflow path)
    else:
        rows[3][:12] = 7.5                                              # a 12-way tie at the top
    rows[4][17] = 50.0                                                  # a near-deterministic row
    want, us = [], []
    for s, r in enumerate(rows):
        K.sample_advance(r.to(DEV), st["tok"], st["out"], st["step"], st["pos"], st["seed"], T, k, p, u_out=st["u"])
        us.append(float(SO.uniform(s, seed)))
        want.append(SO.sample(r.numpy(), s, seed, T, k, p))
    got = st["out"][:len(rows)].tolist()
    assert int(st["step"]) == len(rows) and int(st["pos"]) == len(rows) and int(st["tok"]) == got[-1]
    np.testing.assert_array_equal(st["u"][:len(rows)].cpu().numpy(), np.array(us, dtype=np.float32))   # Philox pinned
    # identical ids given the same uniforms; a draw may differ only if u*Z lands within float rounding of a CDF step
    diff = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    for i in diff:
        keep, e = SO.kept_and_weights(rows[i].numpy(), T, k, p)
        cdf = np.cumsum(e.astype(np.float64)) / e.astype(np.float64).sum()
        assert np.min(np.abs(cdf - us[i])) < 1e-6, (i, got[i], want[i])
    assert len(diff) <= 1, diff
    if k == 1:
        assert all(got[i] == int(r.argmax()) for i, r in enumerate(rows) if i != 3)   # top_k = 1 is greedy (row 3: ties stay)


def test_sampling_statistics_follow_the_distribution():
    probs = np.array([0.5, 0.25, 0.125, 0.125], dtype=np.float32)
    logits = torch.log(torch.from_numpy(probs)).to(DEV)
    n = 4000
    st = _state(n)
    st["seed"].fill_(99)
    for _ in range(n):
        K.sample_advance(logits, st["tok"], st["out"], st["step"], st["pos"], st["seed"], 1.0, 0, 1.0)
    freq = np.bincount(st["out"].cpu().numpy(), minlength=4) / n
    assert np.abs(freq - probs).max() < 0.03, freq


def _mini_lm():
    H, P, image = 512, 8, 112
    ids = syn.token_ids(vocab_base=990)
    tower = ClipVisionTower(syn.vit_state(H, 4 * H, 12, image, seed=8), heads=8, device=DEV)
    dec = LlamaDecoder(syn.llama_state(512, 1408, 2, ids.vocab, seed=9), heads=4, max_positions=256, device=DEV)
    model = SPILlavaLlamaModel(tower, dec, ids, embed_dims=H)
    model.spi_module.load_state_dict(syn.spi_state(model.spi_module, 3))
    g = torch.Generator().manual_seed(31)
    img = torch.randn(1, 3, image, image, generator=g).to(DEV)
    boxes = [syn.boxes(2, g).to(DEV)]
    prompt = syn.prompt_ids(ids, P, 2, g, sys_len=4, question_len=5, vocab_base=990)[None].to(DEV)
    return SPILlavaMPTForCausalLM(model), ids, prompt, img, boxes


class _Tok:
    """ids are their own words; '###' is token 5."""

    def __call__(self, text):
        from types import SimpleNamespace
        return SimpleNamespace(input_ids=[5] if text == "###" else [1, 2])

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join("###" if t == 5 else f"t{t}" for t in row) for row in ids.tolist()]


def test_generate_the_way_app_py_calls_it():
    """app.py:286-300: forward rebound with partial(img_metas, bboxes), inference_mode, generate(input_ids, images=...,
    do_sample=True, temperature=0.2, max_new_tokens=..., stopping_criteria=[...]) -> full id tensor."""
    lm, ids, prompt, img, boxes = _mini_lm()
    T = prompt.size(1)
    with torch.inference_mode():
        want = lm.generate(prompt, images=img, bboxes=boxes, do_sample=False, max_new_tokens=12)
        lm.orig_forward = lm.forward
        lm.forward = partial(lm.orig_forward, img_metas=[None], bboxes=boxes)
        greedy = lm.generate(prompt, images=img, do_sample=False, max_new_tokens=12)
        s1 = lm.generate(prompt, images=img, do_sample=True, temperature=0.2, max_new_tokens=12, seed=7)
        s2 = lm.generate(prompt, images=img, do_sample=True, temperature=0.2, max_new_tokens=12, seed=7)
        s3 = lm.generate(prompt, images=img, do_sample=True, temperature=5.0, top_k=0, max_new_tokens=12, seed=8)
        lm.forward = lm.orig_forward
    assert greedy.shape == (1, T + 12) and torch.equal(greedy[:, :T], prompt)
    assert torch.equal(greedy, want)                                   # the partial-bound boxes were honoured
    nobox = lm.generate(prompt, images=img, bboxes=[torch.zeros(0, 4, device=DEV)], do_sample=False, max_new_tokens=1)
    assert nobox.shape == (1, T + 1)
    assert torch.equal(s1, s2) and s1.shape == (1, T + 12)             # a fixed seed is reproducible
    assert not torch.equal(s3, greedy)                                 # hot sampling leaves the greedy path
    # greedy ids == the host-loop greedy decode of the same embeddings
    with torch.no_grad():
        emb = lm.model.embed_inputs(prompt, img, boxes)
        assert greedy[0, T:].tolist() == lm.model.llama.greedy(emb, 12)
    # stopping: token-id keyword and eos
    stop_tok = int(greedy[0, T + 4])
    first = greedy[0, T:].tolist().index(stop_tok)

    class Tok(_Tok):
        def __call__(self, text):
            from types import SimpleNamespace
            return SimpleNamespace(input_ids=[stop_tok])

        def batch_decode(self, ids_, skip_special_tokens=True):
            return ["x"]
    crit = KeywordsStoppingCriteria(["###"], Tok(), prompt)
    cut = lm.generate(prompt, images=img, bboxes=boxes, max_new_tokens=12, stopping_criteria=[crit])
    assert cut.shape == (1, T + first + 1) and torch.equal(cut, greedy[:, :T + first + 1])
    cut2 = lm.generate(prompt, images=img, bboxes=boxes, max_new_tokens=12, eos_token_id=stop_tok)
    assert torch.equal(cut2, cut)


def test_sampled_decode_replays_the_oracle_on_its_own_logits():
    """The device loop with do_sample: every emitted id equals the oracle's draw from the logits of that step with the
    Philox uniform of (seed, step) -- teacher-forced along the emitted sequence."""
    lm, ids, prompt, img, boxes = _mini_lm()
    dec = lm.model.llama
    with torch.no_grad():
        emb = lm.model.embed_inputs(prompt, img, boxes)
        n, seed, T, k = 10, 4242, 0.8, 20
        got = dec.decode_graph(emb, n, sampler=(T, k, 1.0), seed=seed)
        assert dec.decode_graph(emb, n, sampler=(T, k, 1.0), seed=seed, use_graph=False) == got
        dec.reset(1)
        logits = dec.forward(emb, all_logits=False)
        for s in range(n):
            want = SO.sample(logits.view(-1).cpu().numpy(), s, seed, T, k, 1.0)
            assert got[s] == want, (s, got, want)
            logits = dec.forward(dec.embed[torch.tensor([[got[s]]], device=DEV)].view(1, 1, -1), all_logits=False)


def test_from_pretrained_directory_round_trip_on_the_gpu(tmp_path):
    """save_pretrained -> from_pretrained(dir, vision tower from a local CLIP dir) -> identical logits and greedy ids."""
    lm, ids, prompt, img, boxes = _mini_lm()
    clip = str(tmp_path / "clip")
    ckpt.save_hf_state_dict({f"vision_model.{k}": v for k, v in syn.vit_state(512, 2048, 12, 112, seed=8).items()}, clip)
    import json
    with open(os.path.join(clip, "config.json"), "w") as f:
        json.dump({"vision_config": {"num_attention_heads": 8, "layer_norm_eps": 1e-5, "image_size": 112, "patch_size": 14}}, f)
    lm.model.config.mm_vision_tower = clip
    d = str(tmp_path / "gpt4roi-mini")
    lm.save_pretrained(d)
    back = SPILlavaMPTForCausalLM.from_pretrained(d, low_cpu_mem_usage=True, torch_dtype=torch.bfloat16, use_cache=True)
    with torch.no_grad():
        a = lm(input_ids=prompt, images=img, bboxes=boxes).logits
        b = back(input_ids=prompt, images=img, bboxes=boxes).logits
    # the vision tower was re-read from fp32 files, the decoder from its own bf16 export: same bf16 weights either way
    assert torch.equal(a, b)
    # the call shape of app.py:70-75 (torch_dtype=torch.float16) selects the fp16 instantiation of the kernels (round 4):
    # an fp16 model of the same weights -- logits within the bf16 model's own rounding of it
    half = SPILlavaMPTForCausalLM.from_pretrained(d, low_cpu_mem_usage=True, torch_dtype=torch.float16, use_cache=True)
    assert half.model.llama.dtype == torch.float16 and half.model.vision_tower[0].dtype == torch.float16
    with torch.no_grad():
        c = half(input_ids=prompt, images=img.half(), bboxes=[bx.half() for bx in boxes]).logits
    assert c.dtype == torch.float32 and ((c - a).abs().max() / a.abs().max()).item() < 3e-2
    assert torch.equal(lm.generate(prompt, images=img, bboxes=boxes, max_new_tokens=6),
                       back.generate(prompt, images=img, bboxes=boxes, max_new_tokens=6))
    info = back.model.initialize_vision_modules(clip, mm_vision_select_layer=-2)
    assert info["image_token_len"] == 64 and info["vision_config"].mm_hidden_size == 512


class _BotTok:
    """Word-level tokenizer with both call styles the bot's collaborators use: HF batch call (prompt.py) and the plain
    call + batch_decode of KeywordsStoppingCriteria.  Special tokens carry the model's configured ids."""

    def __init__(self, ids, model_max_length=2048):
        self.special = {"<im_patch>": ids.im_patch_token, "<bbox>": ids.bbox_token, "<point>": ids.point_token,
                        "<im_start>": ids.im_start_token, "<im_end>": ids.im_end_token}
        self.vocab = {"<pad>": 0, "<s>": 1, "###": 5}
        self.pad_token_id, self.model_max_length = 0, model_max_length

    def _ids(self, text):
        import re
        out = []
        for piece in re.split("(" + "|".join(re.escape(t) for t in self.special) + ")", text):
            if piece in self.special:
                out.append(self.special[piece])
            else:
                out += [self.vocab.setdefault(w, 10 + len(self.vocab)) for w in piece.split()]
        return out

    def __call__(self, text, return_tensors=None, padding=None, max_length=None, truncation=None):
        from types import SimpleNamespace
        if return_tensors is None:
            return SimpleNamespace(input_ids=self._ids(text))
        rows = [[1] + self._ids(t) for t in ([text] if isinstance(text, str) else text)]
        n = max(len(r) for r in rows)
        return SimpleNamespace(input_ids=torch.tensor([r + [0] * (n - len(r)) for r in rows], dtype=torch.int64))

    def batch_decode(self, ids, skip_special_tokens=True):
        inv = {v: k for k, v in self.vocab.items()}
        return [" ".join(inv.get(t, f"t{t}") for t in row if not (skip_special_tokens and t in self.special.values()))
                for row in ids.tolist()]


def test_headless_conversation_bot_rounds():
    """serve.ConversationBot.run (gpt4roi/app.py:243-328 without the UI): check -> prompt assembly -> image kernel ->
    partial-bound forward -> sampling generate with the '###' criterion -> answer + history; a fixed seed reproduces the
    answer, greedy equals a direct generate() on the same inputs, a second round reuses the boxes of the first."""
    from gpt4roi_amd.serve import ConversationBot
    lm, ids, _, _, _ = _mini_lm()
    tok = _BotTok(ids)
    bot = ConversationBot(lm, tok, image_size=112)
    g = torch.Generator().manual_seed(5)
    picture = torch.randint(0, 256, (300, 400, 3), generator=g, dtype=torch.uint8).numpy()
    image = {"image": picture, "boxes": [[20, 30, 220, 260], [100, 10, 390, 200]]}
    ans1, err, hist1 = bot.run("What is <region1> doing next to <region2> ?", image, [], max_new_tokens=10, seed=3)
    ans2, _, _ = bot.run("What is <region1> doing next to <region2> ?", image, [], max_new_tokens=10, seed=3)
    assert err is None and isinstance(ans1, str) and ans1 == ans2 and len(ans1) > 0
    conv = hist1[-1]["sources"]["conversations"]
    assert [t["from"] for t in conv] == ["human", "gpt"] and conv[1]["value"] == ans1.replace("Assistant: ", "")
    assert hist1[-1]["bboxes"].shape == (2, 4) and float(hist1[-1]["bboxes"].max()) <= 1.0
    assert lm.forward == lm.orig_forward                                   # the partial binding was undone
    # greedy through the bot == generate() called directly with the bot's own inputs
    hist = []
    data, hist = bot.init_inputs(image, "What is <region1> doing next to <region2> ?", hist)
    T = data["input_ids"].numel()
    assert int((data["input_ids"] == ids.bbox_token).sum()) == 2 and int((data["input_ids"] == ids.im_patch_token).sum()) == 64
    direct = lm.generate(data["input_ids"][None].to(DEV), images=data["image"][None], bboxes=[data["bboxes"].to(DEV)],
                         do_sample=False, max_new_tokens=6)
    ans_g, _, hist_g = bot.run("What is <region1> doing next to <region2> ?", image, [], do_sample=False, max_new_tokens=6)
    want = tok.batch_decode(direct[:, T:])[0].strip()
    want = want[:-3].strip() if want.endswith("###") else want
    assert ans_g == want
    # second round: no new box, an old region mentioned again -> same two boxes, longer prompt
    ans3, err3, hist3 = bot.run("And what colour is <region2> ?", {"image": picture, "boxes": []}, hist_g, do_sample=False,
                                max_new_tokens=4)
    assert err3 is None and len(hist3[-1]["sources"]["conversations"]) == 4 and hist3[-1]["bboxes"].shape == (2, 4)
    # malformed round: a region mentioned without a box
    none, err4, _ = bot.run("What about <region7> ?", {"image": picture, "boxes": []}, hist3)
    assert none is None and "region" in err4
