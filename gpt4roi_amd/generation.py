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
    """llava/model/utils.py:26-46, same constructor and the same two-stage test: a single-token keyword id at the end of
    the sequence, else the keyword string inside the decoded new tokens.  As in the reference, the FIRST call only
    records the prompt length (HF calls the criteria once per generated token)."""

    def __init__(self, keywords, tokenizer, input_ids):
        self.keywords = keywords
        self.keyword_ids = [tokenizer(keyword).input_ids for keyword in keywords]
        self.keyword_ids = [keyword_id[0] for keyword_id in self.keyword_ids
                            if type(keyword_id) is list and len(keyword_id) == 1]
        self.tokenizer = tokenizer
        self.start_len = None
        self.input_ids = input_ids

    def __call__(self, output_ids, scores=None, **kwargs) -> bool:
        if self.start_len is None:
            self.start_len = self.input_ids.shape[1]
        else:
            for keyword_id in self.keyword_ids:
                if output_ids[0, -1] == keyword_id:
                    return True
            outputs = self.tokenizer.batch_decode(output_ids[:, self.start_len:], skip_special_tokens=True)[0]
            for keyword in self.keywords:
                if keyword in outputs:
                    return True
        return False


def prepare_inputs_for_generation(input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kwargs):
    """llava/model/llava.py:263-283: after step 0 only the last token is fed; `images` rides along every step (the
    vision branch is skipped for single-token calls, spi_llava.py:47-48)."""
    if past_key_values:
        input_ids = input_ids[:, -1:]
    if inputs_embeds is not None and past_key_values is None:
        model_inputs = {"inputs_embeds": inputs_embeds}
    else:
        model_inputs = {"input_ids": input_ids}
    model_inputs.update({"past_key_values": past_key_values, "use_cache": kwargs.get("use_cache"),
                         "attention_mask": attention_mask, "images": kwargs.get("images", None)})
    return model_inputs


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


def check_right_padded(attention_mask):
    """The decoder kernels are causal and position-indexed from 0: a batch may be right-padded (the collator pads on
    the right and masks the pad labels with -100, data_modules.py:22-56) but not left-padded.  Raises otherwise instead
    of silently attending to pad tokens.  One small device->host read, batches of more than one row only."""
    if attention_mask is None or attention_mask.dim() != 2 or attention_mask.size(0) == 1 and bool(attention_mask.all()):
        return
    m = attention_mask.to(torch.bool)
    if m.size(1) > 1 and not bool((m[:, :-1] | ~m[:, 1:]).all()):
        raise ValueError("attention_mask must be all ones or right-padded: left padding / holes are not supported by the "
                         "position-indexed KV cache of this path")
