"""Generation glue of the region-feature path.

Mirrors what the reference's callers touch around `model.generate` (gpt4roi/app.py:285-301):
  * `KeywordsStoppingCriteria(keywords, tokenizer, input_ids)`      llava/model/utils.py:26-46
  * `prepare_inputs_for_generation(input_ids, past_key_values, ...)` llava/model/llava.py:263-283
  * the sampling defaults HF's `generate(do_sample=True, temperature=0.2)` runs with at the pinned transformers commit
    (GenerationConfig: top_k = 50, top_p = 1.0), executed by the device-side sampler g4r_sample_advance_f32.
Host-side logic only; the decode loop itself is LlamaDecoder.decode_graph (one hipGraph replay per token).
"""
from dataclasses import dataclass
from typing import Optional

import torch


class StoppingCriteria:
    """Call contract of transformers.StoppingCriteria: __call__(input_ids [B, T], scores) -> bool."""

    def __call__(self, input_ids, scores=None, **kwargs) -> bool:
        raise NotImplementedError


class KeywordsStoppingCriteria(StoppingCriteria):
    """Stops generation at a keyword -- the contract of llava/model/utils.py:26-46, which gpt4roi/app.py:289-300 builds with
    `['###']`: constructor `(keywords, tokenizer, input_ids)`; callable `(output_ids, scores) -> bool`.

    Behaviour kept from the reference, because callers (and HF's per-token calling convention) rely on it:
      * the first call arms the criterion (it only remembers how long the prompt was) and answers False;
      * afterwards a hit is EITHER the newest token being the id of a keyword that tokenises to exactly one id, OR the
        keyword text occurring in the decoded continuation (special tokens skipped);
      * only row 0 of the batch is looked at (the app is batch 1).
    Incremental use: `LlamaDecoder.decode_graph` shows it every prefix exactly once."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = list(keywords)
        self.tokenizer = tokenizer
        self.input_ids = input_ids
        self.start_len = None                       # armed by the first call
        single_token_ids = []
        for word in self.keywords:
            ids = tokenizer(word).input_ids
            if isinstance(ids, list) and len(ids) == 1:
                single_token_ids.append(ids[0])
        self.keyword_ids = single_token_ids

    def _prompt_length(self):
        return self.input_ids.shape[1]

    def __call__(self, output_ids, scores=None, **kwargs) -> bool:
        if self.start_len is None:
            self.start_len = self._prompt_length()
            return False
        newest = output_ids[0, -1]
        if any(newest == kid for kid in self.keyword_ids):
            return True
        continuation = self.tokenizer.batch_decode(output_ids[:, self.start_len:], skip_special_tokens=True)[0]
        return any(word in continuation for word in self.keywords)


def prepare_inputs_for_generation(input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kwargs):
    """What HF `generate` asks the model for before every step (contract of llava/model/llava.py:263-283):
    with a KV cache only the newest token id is fed; `inputs_embeds` may replace the ids on the cache-less first step only;
    `images` is forwarded on every step (the vision branch ignores single-token calls, spi_llava.py:47-48)."""
    first_step = past_key_values is None
    if past_key_values:
        input_ids = input_ids[:, -1:]
    feed = {"inputs_embeds": inputs_embeds} if (inputs_embeds is not None and first_step) else {"input_ids": input_ids}
    feed["past_key_values"] = past_key_values
    feed["use_cache"] = kwargs.get("use_cache")
    feed["attention_mask"] = attention_mask
    feed["images"] = kwargs.get("images", None)
    return feed


@dataclass
class SamplingConfig:
    """do_sample / temperature / top_k / top_p as HF's GenerationConfig names them (defaults of the pinned release)."""
    do_sample: bool = False
    temperature: float = 1.0
    top_k: int = 50
    top_p: float = 1.0
    seed: Optional[int] = None

    def sampler(self):
        if not self.do_sample:
            return None
        if not self.temperature > 0:
            raise ValueError("`temperature` has to be a strictly positive float")
        if not 0 < self.top_p <= 1.0:
            raise ValueError("`top_p` has to be a float > 0 and <= 1")
        k = int(self.top_k or 0)
        if k < 0 or k > 1024:
            raise ValueError("`top_k` has to be in [0, 1024] (0 disables it) for the device-side sampler")
        if self.top_p < 1.0 and k == 0:
            raise NotImplementedError("top_p < 1 without top_k is not implemented on the device-side sampler")
        return (float(self.temperature), k, float(self.top_p))


def dense_or_mask(attention_mask):
    """Training batches: None when the mask is absent, all ones or RIGHT-padded (what the collator produces,
    data_modules.py:22-56) -- causal attention already hides the pad keys from every real row and the pad rows carry no
    loss (labels -100), so the dense kernels give the reference's logits on the real rows and its gradients; any other
    pattern (left padding, holes) is returned as a bool mask for the unpad -> varlen -> pad path the reference's
    flash-attention patch takes (llama_flash_attn_monkey_patch.py:60-85).  One small device->host read."""
    if attention_mask is None or attention_mask.dim() != 2:
        return None
    m = attention_mask.to(torch.bool)
    if m.size(1) <= 1 or bool((m[:, :-1] | ~m[:, 1:]).all()):
        return None
    return m
