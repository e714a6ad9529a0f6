"""MI355X drop-in for `gpt4roi.models.spi_llava` (the model-level seam B3 of SURVEY.md 8b).

Mirrors /root/reference/gpt4roi/models/spi_llava.py:
  SPILlavaLlamaModel.forward(input_ids, attention_mask, img_metas, bboxes, past_key_values,
      inputs_embeds, use_cache, output_attentions, output_hidden_states, images, return_dict)  (:23-36)
  SPILlavaMPTForCausalLM.forward(*args, img_metas=None, bboxes=None, **kwargs)                (:226-240)
with the same control flow -- the vision branch runs only when `images` is given and the call is
not a single-token decode step (:47-48); `image_features = hs[-2][:,1:]`, levels
`hs[-2::-3][::-1][-4:]` (:58-82); `spi_module(mlvl, bboxes)` (:83-85); `mm_projector` (:89-97);
patch splice + `<bbox>` injection (:99-196); decoder + lm_head (:198-205, llava.py:235-249) --
but every stage is a hand-written gfx950 kernel sequence (gpt4roi_amd/{vit,layers,llama}.py) and the
per-sample host splice loop is ONE gather kernel (g4r_splice_embed_bf16).

What is intentionally not reproduced: HF PreTrainedModel plumbing (from_pretrained, generate's
sampling modes -- `generate()` here is the greedy path used for parity), the dummy projector
call the reference makes for text-only samples (:94-97, a zero contribution).  `forward(labels=...)`
returns the shifted-label loss; the training STEP (backward, exchange, AdamW) is gpt4roi_amd/train.py.
"""
import itertools
from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.nn as nn

from . import kernels as K
from .layers import MLVLROIQueryModule, PreparedBoxes
from .llama import LlamaDecoder
from .vit import ClipVisionTower

DEFAULT_IMAGE_PATCH_TOKEN = '<im_patch>'
DEFAULT_IM_START_TOKEN = '<im_start>'
DEFAULT_IM_END_TOKEN = '<im_end>'


@dataclass
class CausalLMOutputWithPast:
    logits: torch.Tensor
    past_key_values: Optional[object] = None
    hidden_states: Optional[torch.Tensor] = None
    loss: Optional[torch.Tensor] = None


class SPILlavaLlamaModel(nn.Module):
    """Holds the four stages; `forward` returns the decoder's final hidden states / logits."""

    def __init__(self, vision_tower: ClipVisionTower, llama: LlamaDecoder, token_ids: SimpleNamespace,
                 embed_dims=1024, mm_projector: Optional[nn.Linear] = None):
        super().__init__()
        self.num_level_spi_features = 4
        self.vision_tower = [vision_tower]          # a list, as in the reference (llava.py:48)
        self.llama = llama
        self.config = token_ids                     # im_patch_token, im_start_token, im_end_token, bbox_token
        self.spi_module = MLVLROIQueryModule(embed_dims=embed_dims, out_dims=llama.hidden, num_levels=4)
        self.mm_projector = mm_projector if mm_projector is not None else nn.Linear(embed_dims, llama.hidden)
        self._proj = None
        self.last_status = None

    def prepare(self):
        bf = torch.bfloat16
        dev = self.llama.device
        self.spi_module.to(dev)
        self.spi_module.prepare()
        self._proj = (self.mm_projector.weight.detach().to(device=dev, dtype=bf).contiguous(),
                      self.mm_projector.bias.detach().to(device=dev, dtype=bf).float().contiguous())

    @torch.no_grad()
    def embed_inputs(self, input_ids, images=None, bboxes=None):
        """Stages a4-a15 of SURVEY.md 8a: returns inputs_embeds [B,T,C] bf16 for the decoder."""
        if self._proj is None:
            self.prepare()
        cfg = self.config
        B, T = input_ids.shape
        run_vision = images is not None and T != 1
        img_tok, spi, off = None, None, None
        n_patch = 0
        if run_vision:
            tower = self.vision_tower[0]
            if isinstance(images, (list, tuple)):
                images = torch.stack(list(images), 0)
            keep = tower.forward(images)
            image_features, mlvl = tower.select(keep)
            if bboxes is not None and (isinstance(bboxes, PreparedBoxes) or len(bboxes) > 0):
                if not isinstance(bboxes, PreparedBoxes):      # reference contract: list[B] of [n_i, 4]
                    bboxes = PreparedBoxes(bboxes, images.size(-1), images.device)
                feats = self.spi_module(mlvl, bboxes)
                spi = torch.cat(feats, 0).contiguous() if len(feats) > 1 else feats[0]
                off = bboxes.offsets
            n_patch = image_features.size(1)
            C = image_features.size(2)
            # mm_projector over the patch tokens (strided view of the hidden state, CLS skipped)
            img_tok = torch.empty((B, n_patch, self.llama.hidden), dtype=torch.bfloat16, device=images.device)
            for b in range(B):
                K.gemm(image_features[b], self._proj[0], bias=self._proj[1], out=img_tok[b])
        embeds, status = K.splice_embed(input_ids.contiguous(), self.llama.embed, img_tok, spi, off, n_patch,
                                        cfg.im_patch_token, cfg.bbox_token, cfg.im_start_token, cfg.im_end_token)
        self.last_status = status
        return embeds

    def prepare_boxes(self, bboxes, image_size):
        """Do the host-side part of a request once (see layers.PreparedBoxes); the returned object can be
        passed as `bboxes=` and makes forward() free of host<->device traffic (hipGraph-capturable)."""
        return PreparedBoxes(bboxes, image_size, self.llama.device)

    def clone_context(self):
        """A second request context: shares every weight / prepared buffer with `self`, owns its KV
        cache and status.  Two contexts driven on two HIP streams keep the 256 CUs busy through the
        wave-quantisation tails and the latency-bound small kernels of a batch-1 request
        (+30 % region-tokens/s on MI355X)."""
        import copy
        other = copy.copy(self)                      # nn.Module shallow copy: parameters are shared
        other.llama = copy.copy(self.llama)
        other.llama._alloc_cache(self.llama.kc.size(1))
        other.last_status = None
        return other

    def check_status(self):
        """Raise like the reference (spi_llava.py:115-128) if the last splice saw a malformed prompt.
        Costs a device sync; call it where the reference would have raised."""
        st = self.last_status
        if st is not None and bool((st != 0).any()):
            codes = st.tolist()
            if any(c & 12 for c in codes):
                raise ValueError('The image end token should follow the image start token.')
            if any(c & 2 for c in codes):
                raise ValueError('The number of <bbox> tokens does not match the number of regions.')
            raise ValueError(f'malformed multimodal prompt (status {codes})')

    @torch.no_grad()
    def forward(self, input_ids: torch.LongTensor = None, attention_mask: Optional[torch.Tensor] = None,
                img_metas=None, bboxes=None, past_key_values=None, inputs_embeds: Optional[torch.Tensor] = None,
                use_cache: Optional[bool] = None, output_attentions: Optional[bool] = None,
                output_hidden_states: Optional[bool] = None, images: Optional[torch.Tensor] = None,
                return_dict: Optional[bool] = None, all_logits=True):
        if inputs_embeds is None:
            inputs_embeds = self.embed_inputs(input_ids, images, bboxes)
        if past_key_values is None:
            self.llama.reset(inputs_embeds.size(0))
        return self.llama.forward(inputs_embeds, all_logits=all_logits)


class SPILlavaMPTForCausalLM(nn.Module):
    """Same call shape as the reference class of that name (despite "MPT" it is the LLaMA model)."""

    def __init__(self, model: SPILlavaLlamaModel):
        super().__init__()
        self.model = model

    def get_model(self):
        return self.model

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, return_dict=None,
                *, img_metas=None, bboxes=None):
        logits = self.model(input_ids=input_ids, attention_mask=attention_mask, img_metas=img_metas, bboxes=bboxes,
                            past_key_values=past_key_values, inputs_embeds=inputs_embeds, images=images)
        loss = None
        if labels is not None:
            # shifted-label token cross entropy, llava/model/llava.py:240-252 (evaluation of the loss only; the
            # training step with its hand-written backward is gpt4roi_amd/train.py::RegionTrainer)
            B, T, V = logits.shape
            lab = torch.full((B, T), -100, dtype=torch.int64, device=logits.device)
            lab[:, :-1] = labels[:, 1:]
            lab = lab.reshape(-1).contiguous()
            cnt = (lab >= 0).sum().clamp(min=1).float()
            loss_sum = torch.zeros(1, dtype=torch.float32, device=logits.device)
            K.cross_entropy(logits.view(B * T, V), lab, loss_sum)
            loss = (loss_sum / cnt).reshape(())
        return CausalLMOutputWithPast(logits=logits, past_key_values=self.model.llama, loss=loss)

    @torch.no_grad()
    def generate(self, input_ids, images=None, bboxes=None, max_new_tokens=64, do_sample=False, stop_ids=(), **_):
        """Greedy decode (the parity mode; app.py:294-300 samples with T=0.2 instead)."""
        if do_sample:
            raise NotImplementedError("only greedy decoding is implemented")
        embeds = self.model.embed_inputs(input_ids, images, bboxes)
        return self.model.llama.greedy(embeds, max_new_tokens, stop_ids)
