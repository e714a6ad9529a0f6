"""tests/golden/make_setup_golden.py -- golden outputs of the REFERENCE'S OWN setup / generation-glue / delta code.

Runs only in the build container (needs /root/reference); writes tests/golden/setup_ref.npz + setup_ref.json.
Every function below is imported from the reference unmodified; only names the reference pulls from un-importable
packages are stubbed (same technique as make_splice_golden.py):

  initialize_vision_tokenizer   gpt4roi/models/spi_llava.py:242-306   called as an unbound method on a stub `self` that
                                exposes resize_token_embeddings / get_{input,output}_embeddings over plain tensors
  KeywordsStoppingCriteria      llava/model/utils.py:26-46             (its `from llava.model import *` is stubbed)
  prepare_inputs_for_generation llava/model/llava.py:263-283           (AutoConfig.register is a no-op during import)
  apply_delta                   scripts/apply_delta.py:15-43           with from_pretrained / save_pretrained replaced by
                                in-memory state-dict holders (the arithmetic in between is the reference's)
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = "/root/reference"


class ToyTokenizer:
    """Whitespace tokenizer with `add_tokens`: the calls the reference makes (len, add_tokens, convert_tokens_to_ids,
    __call__().input_ids, batch_decode).  Shared with the tests (tests/test_setup_cpu.py imports it from here)."""

    def __init__(self, words):
        self.vocab = {w: i for i, w in enumerate(words)}
        self.special = set()

    def __len__(self):
        return len(self.vocab)

    def add_tokens(self, toks, special_tokens=False):
        n = 0
        for t in toks:
            if t not in self.vocab:
                self.vocab[t] = len(self.vocab)
                n += 1
            if special_tokens:
                self.special.add(t)
        return n

    def convert_tokens_to_ids(self, toks):
        return [self.vocab[t] for t in toks]

    def __call__(self, text):
        return types.SimpleNamespace(input_ids=[self.vocab[w] for w in text.split() if w in self.vocab])

    def batch_decode(self, ids, skip_special_tokens=True):
        inv = {i: w for w, i in self.vocab.items()}
        out = []
        for row in ids.tolist():
            ws = [inv[i] for i in row]
            if skip_special_tokens:
                ws = [w for w in ws if w not in self.special]
            out.append(" ".join(ws))
        return out


class HFToyTokenizer:
    """HF call signature for the prompt functions (train.py:125-148): tokenizer(text | [texts], return_tensors='pt',
    padding='longest', max_length=..., truncation=True).input_ids; a BOS id in front of EVERY tokenisation (as LLaMA's
    tokenizer does -- the reference's span arithmetic depends on it), special tokens split out of running text, ids handed
    out on first sight."""
    SPECIALS = ['<im_start>', '<im_end>', '<im_patch>', '<bbox>', '<point>', '<image>']

    def __init__(self, model_max_length=512):
        self.vocab = {'<pad>': 0, '<s>': 1}
        self.pad_token_id, self.bos_token_id = 0, 1
        self.model_max_length = model_max_length

    def _ids(self, text):
        import re
        out = [self.bos_token_id]
        pat = '(' + '|'.join(re.escape(t) for t in self.SPECIALS) + ')'
        for piece in re.split(pat, text):
            for w in ([piece] if piece in self.SPECIALS else piece.split()):
                out.append(self.vocab.setdefault(w, len(self.vocab)))
        return out[:self.model_max_length]

    def __call__(self, text, return_tensors=None, padding=None, max_length=None, truncation=None):
        rows = [self._ids(t) for t in ([text] if isinstance(text, str) else text)]
        n = max(len(r) for r in rows)
        ids = torch.tensor([r + [self.pad_token_id] * (n - len(r)) for r in rows], dtype=torch.int64)
        return types.SimpleNamespace(input_ids=ids)


def prompt_cases():
    return [
        [{'from': 'human', 'value': '<image>\nWhat is in region1 <bbox> and region2 <bbox> ?'},
         {'from': 'gpt', 'value': 'A cat sits next to a dog .'},
         {'from': 'human', 'value': 'What colour is region1 <bbox> ?'},
         {'from': 'gpt', 'value': 'It is black .'}],
        [{'from': 'human', 'value': 'Describe <bbox> briefly .\n<image>'},
         {'from': 'gpt', 'value': 'A red car .'}],
    ]


def run_ref_prompt():
    """preprocess_multimodal + preprocess of gpt4roi/train/train.py on the cases above (its module-level imports of the
    trainer and of llava.model are stubs; llava/conversation.py is the reference's own file)."""
    conv = _import(f"{REF}/llava/conversation.py", "llava.conversation", {})
    llava_pkg = types.ModuleType("llava")
    llava_pkg.conversation = conv
    model_pkg = types.ModuleType("llava.model")
    model_pkg.__all__ = []
    trainer = types.ModuleType("gpt4roi.train.llava_trainer")
    trainer.LLaVATrainer = object
    mod = _import(f"{REF}/gpt4roi/train/train.py", "ref_train",
                  {"llava": llava_pkg, "llava.conversation": conv, "llava.model": model_pkg,
                   "gpt4roi": types.ModuleType("gpt4roi"), "gpt4roi.train": types.ModuleType("gpt4roi.train"),
                   "gpt4roi.train.llava_trainer": trainer})
    out = []
    for front in (False, True):
        for use_se in (True, False):
            tok = HFToyTokenizer()
            import copy
            src = copy.deepcopy(prompt_cases())
            if front:
                src = [s for s in src if '<image>' in s[0]['value']]
            cfg = dict(is_multimodal=True, sep_image_conv_front=front, use_im_start_end=use_se)
            src = mod.preprocess_multimodal(src, cfg, 4)
            texts = [[t['value'] for t in s] for s in copy.deepcopy(src)]
            d = mod.preprocess(src, tok)
            out.append(dict(front=front, use_im_start_end=use_se, after_multimodal=texts,
                            input_ids=[t.tolist() for t in d['input_ids']], labels=[t.tolist() for t in d['labels']]))
    return out


def toy_tokenizer():
    return ToyTokenizer(["<unk>", "<s>", "</s>"] + [f"w{i}" for i in range(20)] + ["###", "stop"])


def _import(path, name, stubs):
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def ref_spi_llava():
    from make_splice_golden import import_reference
    return import_reference()


def ref_utils():
    pkg = types.ModuleType("llava.model")
    pkg.__all__ = []
    return _import(f"{REF}/llava/model/utils.py", "ref_llava_utils", {"llava": types.ModuleType("llava"), "llava.model": pkg})


def ref_llava():
    import transformers
    orig_c, orig_m = transformers.AutoConfig.register, transformers.AutoModelForCausalLM.register
    transformers.AutoConfig.register = staticmethod(lambda *a, **k: None)
    transformers.AutoModelForCausalLM.register = staticmethod(lambda *a, **k: None)
    try:
        return _import(f"{REF}/llava/model/llava.py", "ref_llava_llava", {})
    finally:
        transformers.AutoConfig.register, transformers.AutoModelForCausalLM.register = orig_c, orig_m


# ------------------------------------------------------------------------------------------ initialize_vision_tokenizer
def tokenizer_case(seed=7, V=25, C=8):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(V, C, generator=g), torch.randn(V, C, generator=g)


def run_ref_tokenizer_init(embed, head):
    ref = ref_spi_llava()
    tok = toy_tokenizer()
    assert len(tok) == embed.size(0)

    class Emb:
        def __init__(self, w):
            self.weight = nn.Parameter(w.clone())

        def parameters(self):
            return [self.weight]

    class Self:
        def __init__(self):
            self.inp, self.out = Emb(embed), Emb(head)
            self.vc = types.SimpleNamespace()
            self._model = types.SimpleNamespace(vision_tower=[types.SimpleNamespace(config=self.vc)])

        def get_model(self):
            return self._model

        def get_input_embeddings(self):
            return self.inp

        def get_output_embeddings(self):
            return self.out

        def resize_token_embeddings(self, n):          # HF semantics: keep the old rows; new rows here are ZERO
            for e in (self.inp, self.out):
                w = torch.zeros(n, e.weight.size(1))
                k = min(n, e.weight.size(0))
                w[:k] = e.weight.data[:k]
                e.weight = nn.Parameter(w)

        def modules(self):
            return []

    s = Self()
    ref.SPILlavaMPTForCausalLM.initialize_vision_tokenizer(s, True, tok, device="cpu")
    ids = {k: int(getattr(s.vc, k)) for k in ("im_patch_token", "bbox_token", "point_token", "im_start_token", "im_end_token")}
    return s.inp.weight.data.clone(), s.out.weight.data.clone(), ids, len(tok)


# ------------------------------------------------------------------------------------------ stopping criteria
def stopping_cases():
    tok = toy_tokenizer()
    prompt = torch.tensor([[1, 3, 4, 5]])
    seqs = [[3, 4], [3, tok.vocab["###"]], [6, 7, tok.vocab["stop"], 8], [9], [10, 11, 12, tok.vocab["###"], 3]]
    return tok, prompt, seqs


def run_ref_stopping():
    ref = ref_utils()
    tok, prompt, seqs = stopping_cases()
    out = []
    for kw in (["###"], ["stop", "###"], ["w6 w7"]):
        for new in seqs:
            c = ref.KeywordsStoppingCriteria(kw, tok, prompt)
            dec = []
            c(prompt, None)                                  # HF's first call: records the prompt length
            for n in range(1, len(new) + 1):
                full = torch.cat([prompt, torch.tensor([new[:n]])], 1)
                dec.append(bool(c(full, None)))
            out.append(dict(keywords=kw, new=new, decisions=dec))
    return out


# ------------------------------------------------------------------------------------------ prepare_inputs_for_generation
def run_ref_prepare_inputs():
    ref = ref_llava()
    f = ref.LlavaLlamaForCausalLM.prepare_inputs_for_generation
    ids = torch.tensor([[5, 6, 7, 8]])
    img = torch.zeros(1, 3, 2, 2)
    am = torch.ones(1, 4, dtype=torch.long)
    res = []
    for past in (None, "cache"):
        for emb in (None, torch.zeros(1, 4, 2)):
            r = f(None, ids, past_key_values=past, attention_mask=am, inputs_embeds=emb, images=img, use_cache=True)
            res.append(dict(past=past is not None, with_embeds=emb is not None, keys=sorted(r.keys()),
                            input_ids=r["input_ids"].tolist() if "input_ids" in r else None,
                            has_images=r["images"] is img, use_cache=r["use_cache"]))
    return res


# ------------------------------------------------------------------------------------------ apply_delta
def delta_case(seed=11):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    base = {"model.embed_tokens.weight": r(10, 4), "lm_head.weight": r(10, 4), "model.layers.0.w": r(4, 4),
            "model.norm.weight": r(4)}
    delta = {"model.embed_tokens.weight": r(15, 4), "lm_head.weight": r(15, 4), "model.layers.0.w": r(4, 4),
             "model.norm.weight": r(4), "model.mm_projector.weight": r(4, 3), "model.mm_projector.bias": r(4),
             "model.spi_module.roi_align.updims.weight": r(4, 2)}
    return base, delta


def run_ref_apply_delta(base, delta):
    stub = types.ModuleType("gpt4roi.models.spi_llava")
    captured = {}

    class Holder:
        def __init__(self, sd):
            self.sd = {k: nn.Parameter(v.clone(), requires_grad=False) for k, v in sd.items()}

        def state_dict(self):
            return self.sd

        def save_pretrained(self, path):
            captured["target"] = {k: v.data.clone() for k, v in self.sd.items()}

    class SPI:
        @staticmethod
        def from_pretrained(path, **kw):
            return Holder(delta)

    stub.SPILlavaMPTForCausalLM = SPI
    mod = _import(f"{REF}/scripts/apply_delta.py", "ref_apply_delta",
                  {"gpt4roi": types.ModuleType("gpt4roi"), "gpt4roi.models": types.ModuleType("gpt4roi.models"),
                   "gpt4roi.models.spi_llava": stub})
    mod.AutoModelForCausalLM = types.SimpleNamespace(from_pretrained=lambda path, **kw: Holder(base))
    mod.AutoTokenizer = types.SimpleNamespace(
        from_pretrained=lambda path, **kw: types.SimpleNamespace(save_pretrained=lambda p: None))
    mod.tqdm = lambda it, **kw: it
    mod.apply_delta("base", "target", "delta")
    return captured["target"]


if __name__ == "__main__":
    embed, head = tokenizer_case()
    e2, h2, ids, n_tok = run_ref_tokenizer_init(embed, head)
    base, delta = delta_case()
    target = run_ref_apply_delta(base, delta)
    np.savez_compressed(os.path.join(HERE, "setup_ref.npz"), embed_after=e2.numpy(), head_after=h2.numpy(),
                        **{f"target::{k}": v.numpy() for k, v in target.items()})
    with open(os.path.join(HERE, "setup_ref.json"), "w") as f:
        json.dump(dict(token_ids=ids, tokenizer_len=n_tok, stopping=run_ref_stopping(),
                       prepare_inputs=run_ref_prepare_inputs(), prompt=run_ref_prompt()), f, indent=1)
    print("token ids", ids, "len", n_tok)
    print("wrote setup_ref.npz / setup_ref.json")
