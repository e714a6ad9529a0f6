"""CPU: host-side rows around the path against outputs of the REFERENCE'S OWN code
(tests/golden/setup_ref.{npz,json}, written by tests/golden/make_setup_golden.py in the build container):
initialize_vision_tokenizer (spi_llava.py:242-306), KeywordsStoppingCriteria (llava/model/utils.py:26-46),
prepare_inputs_for_generation (llava.py:263-283), apply_delta (scripts/apply_delta.py:15-43); plus the HF-directory
round trip (save_pretrained / from_pretrained / make_delta o apply_delta) and the sampling oracle's pins (Philox
known-answer vectors, kept set == HF's logits warpers).  No kernel is launched here."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from make_setup_golden import delta_case, stopping_cases, tokenizer_case, toy_tokenizer  # noqa: E402

from gpt4roi_amd import checkpoint as ckpt  # noqa: E402
from gpt4roi_amd import synthetic as syn  # noqa: E402
from gpt4roi_amd.generation import (KeywordsStoppingCriteria, SamplingConfig, dense_or_mask,  # noqa: E402
                                    prepare_inputs_for_generation)
from gpt4roi_amd.llama import LlamaDecoder  # noqa: E402
from gpt4roi_amd.spi_llava import SPILlavaLlamaModel, SPILlavaMPTForCausalLM  # noqa: E402
from oracle import sampler_oracle as SO  # noqa: E402


@pytest.fixture(scope="module")
def ref(golden_dir):
    z = np.load(os.path.join(golden_dir, "setup_ref.npz"))
    with open(os.path.join(golden_dir, "setup_ref.json")) as f:
        return z, json.load(f)


def _tiny_lm(vocab=25, hidden=8, layers=1, embed_dims=64, seed=3):
    sd = syn.llama_state(hidden, 16, layers, vocab, seed=seed)
    dec = LlamaDecoder(sd, heads=2, max_positions=32, device="cpu")
    ids = syn.token_ids(vocab - 6)
    inner = SPILlavaLlamaModel(None, dec, ids, embed_dims=embed_dims)
    return SPILlavaMPTForCausalLM(inner), sd


def test_initialize_vision_tokenizer_matches_the_reference(ref):
    z, meta = ref
    embed, head = tokenizer_case()
    lm, _ = _tiny_lm(vocab=embed.size(0), hidden=embed.size(1))
    dec = lm.model.llama
    dec.embed, dec.lm_head = embed.clone(), head.clone()          # fp32 here so the comparison is exact
    tok = toy_tokenizer()
    lm.initialize_vision_tokenizer(True, tok, device="cpu")
    assert len(tok) == meta["tokenizer_len"] == dec.embed.size(0) == dec.vocab
    c = lm.model.config
    assert {k: int(getattr(c, k)) for k in meta["token_ids"]} == meta["token_ids"]
    np.testing.assert_allclose(dec.embed.numpy(), z["embed_after"], atol=1e-6)
    np.testing.assert_allclose(dec.lm_head.numpy(), z["head_after"], atol=1e-6)
    # the last 4 rows (<bbox>, <point>, <im_start>, <im_end>) are the mean of the rows before them; <im_patch> is not
    mean_before = dec.embed[:-4].numpy().mean(0)           # includes the (zero) <im_patch> row, as in the reference
    assert np.allclose(dec.embed[-4:].numpy(), mean_before[None].repeat(4, 0), atol=1e-6)
    assert np.allclose(dec.embed[-5].numpy(), 0.0)         # <im_patch>: a fresh row, never read (its positions are spliced)
    assert lm.model.tokenizer is tok


def test_keywords_stopping_criteria_matches_the_reference(ref):
    _, meta = ref
    tok, prompt, _ = stopping_cases()
    for case in meta["stopping"]:
        c = KeywordsStoppingCriteria(case["keywords"], tok, prompt)
        assert c(prompt, None) is False                    # first call records the prompt length (utils.py:36-37)
        got = [bool(c(torch.cat([prompt, torch.tensor([case["new"][:n]])], 1), None)) for n in range(1, len(case["new"]) + 1)]
        assert got == case["decisions"], case


def test_prepare_inputs_for_generation_matches_the_reference(ref):
    _, meta = ref
    ids = torch.tensor([[5, 6, 7, 8]])
    img = torch.zeros(1, 3, 2, 2)
    am = torch.ones(1, 4, dtype=torch.long)
    k = 0
    for past in (None, "cache"):
        for emb in (None, torch.zeros(1, 4, 2)):
            r = prepare_inputs_for_generation(ids, past_key_values=past, attention_mask=am, inputs_embeds=emb, images=img,
                                              use_cache=True)
            want = meta["prepare_inputs"][k]
            k += 1
            assert sorted(r.keys()) == want["keys"]
            assert (r["input_ids"].tolist() if "input_ids" in r else None) == want["input_ids"]
            assert (r["images"] is img) == want["has_images"] and r["use_cache"] == want["use_cache"]


def test_apply_delta_matches_the_reference_and_inverts_make_delta(ref, tmp_path):
    z, _ = ref
    base, delta = delta_case()
    got = ckpt._combine(delta, base, +1)
    want = {k[len("target::"):]: z[k] for k in z.files if k.startswith("target::")}
    assert set(got) == set(want)
    for k in want:
        np.testing.assert_allclose(got[k].numpy(), want[k], atol=1e-6, err_msg=k)
    with pytest.raises(NameError):
        ckpt._combine({"model.unknown": torch.zeros(2)}, base, +1)
    # directory level: make_delta then apply_delta gives the target back (sharded torch-pickle and safetensors files)
    for safe, shard in ((True, 5 << 30), (False, 200)):
        b, t, d, o = [str(tmp_path / f"{n}{int(safe)}") for n in "btdo"]
        target = {k: v.clone() for k, v in got.items()}
        ckpt.save_hf_state_dict(base, b, safe, shard)
        ckpt.save_hf_state_dict(target, t, safe, shard)
        for p in (b, t):
            with open(os.path.join(p, "config.json"), "w") as f:
                json.dump({"model_type": "llava"}, f)
        ckpt.make_delta(b, t, d, safe)
        back = ckpt.apply_delta(b, o, d, safe)
        for k, v in target.items():
            torch.testing.assert_close(back[k], v, atol=1e-6, rtol=0)
        assert os.path.exists(os.path.join(o, "config.json"))
        assert set(ckpt.load_hf_state_dict(o)) == set(target)


def test_save_pretrained_from_pretrained_round_trip(tmp_path):
    lm, _ = _tiny_lm()
    with torch.no_grad():
        for p in lm.model.spi_module.parameters():
            p.add_(0.01 * torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel() % 97)))
    d = str(tmp_path / "ckpt")
    lm.save_pretrained(d)
    sd0 = lm.state_dict()
    assert "model.spi_module.mlvl_fuse.input_conv.0.weight" in sd0 and "model.mm_projector.bias" in sd0
    assert "model.layers.0.self_attn.q_proj.weight" in sd0 and not any("vision_tower" in k for k in sd0)
    back = SPILlavaMPTForCausalLM.from_pretrained(d, device="cpu", torch_dtype=torch.float16, low_cpu_mem_usage=True,
                                                  use_cache=True)
    sd1 = back.state_dict()
    assert set(sd0) == set(sd1)
    for k in sd0:
        assert torch.equal(sd0[k].float(), sd1[k].float()), k
    c0, c1 = lm.model.config, back.model.config
    assert (c1.im_patch_token, c1.bbox_token, c1.im_start_token, c1.im_end_token) == \
        (c0.im_patch_token, c0.bbox_token, c0.im_start_token, c0.im_end_token)
    with pytest.raises(FileNotFoundError):       # a hub name cannot be resolved: there is no network
        ckpt.load_vision_tower("openai/clip-vit-large-patch14")


def test_vision_tower_directory_loader(tmp_path):
    sd = {f"vision_model.{k}": v for k, v in syn.vit_state(64, 128, 3, 28, seed=5).items()}
    d = str(tmp_path / "clip")
    ckpt.save_hf_state_dict(sd, d)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump({"vision_config": {"num_attention_heads": 4, "layer_norm_eps": 1e-5, "image_size": 28, "patch_size": 14}}, f)
    tower = ckpt.load_vision_tower(d, device="cpu")
    assert tower.heads == 4 and tower.hidden == 64 and tower.image_size == 28 and tower.num_layers == 3


def test_philox_known_answers_and_uniform_range():
    for ctr, key, out in SO.PHILOX_KAT:
        assert SO.philox4x32_10(ctr, key) == out
    us = [float(SO.uniform(s, 1234567890123)) for s in range(2000)]
    assert 0.0 <= min(us) and max(us) < 1.0 and abs(np.mean(us) - 0.5) < 0.03
    assert us[:3] != [float(SO.uniform(s, 1234567890124)) for s in range(3)]


def test_sampler_oracle_keeps_what_hf_warpers_keep():
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    g = torch.Generator().manual_seed(0)
    for (T, k, p) in ((0.2, 50, 1.0), (1.0, 5, 1.0), (0.7, 50, 0.9), (1.3, 8, 0.5)):
        logits = torch.randn(1, 500, generator=g) * 3
        s = TemperatureLogitsWarper(T)(None, logits.clone())
        s = TopKLogitsWarper(k)(None, s)
        if p < 1.0:
            s = TopPLogitsWarper(p)(None, s)
        hf_keep = torch.isfinite(s[0]).nonzero().flatten().numpy()
        keep, e = SO.kept_and_weights(logits[0].numpy(), T, k, p)
        assert np.array_equal(keep, hf_keep), (T, k, p)
        want = torch.softmax(s[0][torch.isfinite(s[0])], 0).numpy()
        np.testing.assert_allclose(e / e.sum(), want, rtol=2e-5)
    # the draw is the inverse CDF in ascending vocabulary order
    lg = np.log(np.array([0.1, 0.2, 0.3, 0.4], dtype=np.float32))
    assert [SO.sample(lg, 0, 0, 1.0, 0, 1.0, u=u) for u in (0.0, 0.09, 0.11, 0.29, 0.31, 0.59, 0.61, 0.999)] == \
        [0, 0, 1, 1, 2, 2, 3, 3]


def test_sampling_config_and_mask_validation():
    assert SamplingConfig().sampler() is None
    assert SamplingConfig(do_sample=True, temperature=0.2).sampler() == (0.2, 50, 1.0)
    with pytest.raises(ValueError):
        SamplingConfig(do_sample=True, temperature=0.0).sampler()
    with pytest.raises(NotImplementedError):
        SamplingConfig(do_sample=True, top_k=0, top_p=0.9).sampler()
    # training batches: right padding (the collator's) rides the dense kernels, anything else is handed on as a mask
    assert dense_or_mask(None) is None
    assert dense_or_mask(torch.tensor([[1, 1, 1, 0], [1, 1, 1, 1]])) is None
    m = dense_or_mask(torch.tensor([[0, 1, 1, 1], [1, 1, 1, 1]]))
    assert m.dtype == torch.bool and m.tolist() == [[False, True, True, True], [True] * 4]
    assert dense_or_mask(torch.tensor([[1, 0, 1, 1]])) is not None


def test_prompt_assembly_matches_the_reference(ref):
    """preprocess_multimodal + preprocess (gpt4roi/train/train.py:185-208, 354-386) on two conversations, with and
    without <im_start>/<im_end> and with the image moved to the front -- ids AND label masks equal to the reference's."""
    import copy
    from make_setup_golden import HFToyTokenizer, prompt_cases
    from gpt4roi_amd import prompt as PR
    _, meta = ref
    assert len(meta["prompt"]) == 4
    for case in meta["prompt"]:
        tok = HFToyTokenizer()
        src = copy.deepcopy(prompt_cases())
        if case["front"]:
            src = [s for s in src if '<image>' in s[0]['value']]
        src = PR.preprocess_multimodal(src, dict(is_multimodal=True, sep_image_conv_front=case["front"],
                                                 use_im_start_end=case["use_im_start_end"]), 4)
        assert [[t['value'] for t in s] for s in src] == case["after_multimodal"]
        out = PR.preprocess(src, tok)
        assert [t.tolist() for t in out["input_ids"]] == case["input_ids"]
        assert [t.tolist() for t in out["labels"]] == case["labels"]
        assert any(v == PR.IGNORE_INDEX for v in case["labels"][0]) and any(v != PR.IGNORE_INDEX for v in case["labels"][0])
    assert PR.region_question("what is <region1> and <2> next to <> ?") == "what is region1 <bbox> and region2 <bbox> next to <bbox> ?"
    one = PR.build_sample(prompt_cases()[0], HFToyTokenizer(), 4)
    assert one["input_ids"].dtype == torch.int64 and one["input_ids"].shape == one["labels"].shape
    # the result goes straight into the collator of the batch contract
    from gpt4roi_amd.data import DataCollatorForDetDataset
    batch = DataCollatorForDetDataset(pad_token_id=0)([
        dict(one, image=torch.zeros(3, 28, 28), bboxes=torch.zeros(2, 4), img_metas={}),
        dict(PR.build_sample(prompt_cases()[1], HFToyTokenizer(), 4), image=torch.zeros(3, 28, 28), bboxes=torch.zeros(1, 4),
             img_metas={})])
    assert batch["input_ids"].shape == batch["labels"].shape and batch["input_ids"].size(0) == 2


def test_ragged_layout_index_sets_on_the_host():
    """llama.RaggedLayout is plain index bookkeeping (no kernel): the front-packed slot index (-1 behind a sequence), its inverse with -1 on the pad rows,
    the cache rows of the kept positions and each sequence's last kept row, for left padding, holes and an empty row."""
    from gpt4roi_amd.llama import RaggedLayout
    m = torch.tensor([[0, 0, 1, 1, 1], [1, 1, 1, 1, 1], [1, 0, 1, 0, 0], [0, 0, 0, 0, 0]])
    r = RaggedLayout.of(m, 16)
    assert r.lens == [3, 5, 2, 0] and r.cu == [0, 3, 8, 10, 10] and r.nnz == 10
    assert r.Tc == 5 and r.idx.tolist() == [2, 3, 4, -1, -1, 5, 6, 7, 8, 9, 10, 12, -1, -1, -1] + [-1] * 5
    assert r.idx_cache.tolist() == [2, 3, 4, -1, -1, 16, 17, 18, 19, 20, 32, 34, -1, -1, -1] + [-1] * 5
    assert r.inv.tolist() == [-1, -1, 0, 1, 2, 5, 6, 7, 8, 9, 10, -1, 11] + [-1] * 7
    assert r.last.tolist() == [4, 9, 12, -1]
    assert RaggedLayout.of(torch.ones(2, 3), 16) is None and RaggedLayout.of(None, 16) is None
    # packing then padding back with `inv` is the identity on the kept rows and zero on the pad rows
    x = torch.arange(20.).view(20, 1) + 1
    packed = torch.where(r.idx[:, None] >= 0, x[r.idx.clamp(min=0).long()], torch.zeros(1))
    back = torch.where(r.inv[:, None] >= 0, packed[r.inv.clamp(min=0).long()], torch.zeros(1))
    assert torch.equal(back * m.reshape(-1, 1), back) and torch.equal(back[m.reshape(-1).bool()], x[m.reshape(-1).bool()])
