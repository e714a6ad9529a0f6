"""MI355X drop-in for `gpt4roi.models.spi_llava` (the model-level seam B3 of SURVEY.md 8b).

Mirrors /root/reference/gpt4roi/models/spi_llava.py and the parts of llava/model/llava.py its callers touch:
  SPILlavaLlamaModel.forward(input_ids, attention_mask, img_metas, bboxes, past_key_values,
      inputs_embeds, use_cache, output_attentions, output_hidden_states, images, return_dict)  (:23-36)
  SPILlavaMPTForCausalLM.forward(*args, img_metas=None, bboxes=None, **kwargs)                (:226-240)
  SPILlavaMPTForCausalLM.initialize_vision_tokenizer(...)                                     (:242-306)
  LlavaLlamaModel.initialize_vision_modules(...)                                              (llava.py:54-86)
  LlavaLlamaForCausalLM.prepare_inputs_for_generation / loss                                  (llava.py:235-283)
with the same control flow -- the vision branch runs only when `images` is given and the call is
not a single-token decode step (:47-48); `image_features = hs[-2][:,1:]`, levels
`hs[-2::-3][::-1][-4:]` (:58-82); `spi_module(mlvl, bboxes)` (:83-85); `mm_projector` (:89-97);
patch splice + `<bbox>` injection (:99-196); decoder + lm_head (:198-205, llava.py:235-249) --
but every stage is a hand-written gfx950 kernel sequence (gpt4roi_amd/{vit,layers,llama}.py) and the
per-sample host splice loop is ONE gather kernel (g4r_splice_embed_bf16).

How the reference's callers run unchanged against it:
  * training (gpt4roi/train/train.py:698-712, HF Trainer.training_step): `model(**batch)` with grad enabled returns a
    loss / logits that carry an autograd node; `loss.backward()` runs the hand-written backward of every stage
    (LlamaDecoder.backward, MLVLROIQueryModule.backward) and fills `.grad` of the reference-keyed nn.Parameters
    (`model.spi_module.*`, `model.mm_projector.*`: the stage-1 trainables, train.py:685-696); any torch optimizer then
    steps them and the bf16 kernel copies are refreshed lazily on the next forward (parameter version stamps).
    `gradient_checkpointing_enable()` (train_stage1.sh:36) recomputes each decoder layer in the backward.
  * serving (gpt4roi/app.py:286-300): `model.forward = partial(model.orig_forward, img_metas=[None], bboxes=bboxes)`
    followed by `model.generate(input_ids, images=..., do_sample=True, temperature=0.2, max_new_tokens=1024,
    stopping_criteria=[KeywordsStoppingCriteria(...)])` -- generate() drives `self.forward` (so the bound boxes are
    seen), samples on the device (g4r_sample_advance_f32) and returns the full id tensor [1, T + new] like HF.
Not reproduced: the dummy projector call the reference makes for text-only samples (:94-97, a zero contribution), beam
search and the other HF generation modes the callers never use.  Stage-2 (decoder weights trainable) goes through
gpt4roi_amd/train.py::FullTrainer, whose masters are plain tensors in kernel layout.
"""
import itertools
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from . import kernels as K
from .generation import (KeywordsStoppingCriteria, SamplingConfig, StoppingCriteria,  # noqa: F401 (re-exported)
                         dense_or_mask, prepare_inputs_for_generation)
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
    attentions: Optional[object] = None

    def __getitem__(self, i):          # HF outputs index as tuples with the None entries dropped: (loss?, logits, ...)
        return [v for v in (self.loss, self.logits, self.past_key_values) if v is not None][i]


def add_spatial_token(tokenizer):
    """gpt4roi/models/spi_llava.py:208-212."""
    spi_tokens = ['<bbox>', '<point>']
    num_spi_tokens = tokenizer.add_tokens(spi_tokens, special_tokens=True)
    return tokenizer, num_spi_tokens


# ---------------------------------------------------------------------------------------------------- autograd seams
class _RegionPathFn(torch.autograd.Function):
    """logits = path(input_ids, images, bboxes; trainable parameters).  Forward = `forward_train` of every stage (keeps
    what the hand-written backward needs), backward = LlamaDecoder.backward -> gather of the <bbox> / <im_patch> rows ->
    region-module / projector parameter gradients, returned to autograd in the order the parameters were passed."""

    @staticmethod
    def forward(ctx, model, input_ids, images, bboxes, attention_mask, names, *params):
        with torch.no_grad():
            logits, pctx = model.forward_train(input_ids, images, bboxes, attention_mask=attention_mask)
        ctx.model, ctx.pctx, ctx.names = model, pctx, names
        ctx.shape = (input_ids.size(0), input_ids.size(1))
        ctx.input_ids = input_ids
        return logits.view(input_ids.size(0), input_ids.size(1), -1)

    @staticmethod
    def backward(ctx, dlogits):
        model, (B, T) = ctx.model, ctx.shape
        dec = model.llama
        with torch.no_grad():
            dl = torch.zeros((B * T, dec.v_pad), dtype=torch.bfloat16, device=dlogits.device)
            dl[:, :dec.vocab] = dlogits.reshape(B * T, dec.vocab)
            grads = model.backward(ctx.pctx, dl, train_projector=any(n.startswith("mm_projector.") for n in ctx.names))
            if any(n.startswith("llama.") for n in ctx.names):       # stage 2: the decoder's own weights (and embedding rows)
                grads.update(model.decoder_param_grads(ctx.input_ids))
        ctx.pctx = None
        return (None, None, None, None, None, None) + tuple(grads.get(n) for n in ctx.names)


class _ShiftedCrossEntropyFn(torch.autograd.Function):
    """llava/model/llava.py:240-252 (shift, flatten, CrossEntropyLoss with ignore_index -100) as the fused kernel: loss and
    (softmax - onehot) / n_valid in one pass."""

    @staticmethod
    def forward(ctx, logits, labels, dec):
        B, T, V = logits.shape
        with torch.no_grad():
            loss, dl = dec.loss_and_dlogits(logits.reshape(B * T, V), labels)
        ctx.dl, ctx.shape = dl, (B, T, V)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        B, T, V = ctx.shape
        return (ctx.dl[:, :V].float() * g).view(B, T, V), None, None


class _DecoderLayerMasters(nn.Module):
    pass


class _DecoderMasters(nn.Module):
    """The decoder's weights as fp32 nn.Parameters over its KERNEL-layout tensors (fused q|k|v rows, interleaved gate / up rows:
    `LlamaDecoder.trainable_tensors`) -- what makes stage 2 drivable through the reference's own caller
    (gpt4roi/train/train.py:698-712: HF Trainer.training_step = model(**batch).loss.backward() + a torch optimizer over
    `model.parameters()`; stage 1 freezes them by name, train.py:685-697; train_stage2.sh leaves them trainable).  The bf16
    tensors the kernels read are roundings of these masters, re-derived when an optimizer step bumps a version counter
    (`SPILlavaLlamaModel._maybe_prepare`); the fp32 norm weights ARE the parameters' storage."""

    def __init__(self, dec):
        super().__init__()
        live = dec.trainable_tensors()

        def P(t):
            return nn.Parameter(t if t.dtype == torch.float32 else t.float())
        self.embed_tokens, self.norm, self.lm_head = P(live["embed_tokens"]), P(live["norm"]), P(live["lm_head"])
        self.layers = nn.ModuleList()
        for i in range(len(dec.layers)):
            m = _DecoderLayerMasters()
            for nm in ("wqkv", "wo", "wgu", "wd", "n1", "n2"):
                setattr(m, nm, P(live[f"{i}.{nm}"]))
            self.layers.append(m)

    def named_kernel_tensors(self):
        """(name as LlamaDecoder.trainable_tensors() spells it, parameter), in that order"""
        yield "embed_tokens", self.embed_tokens
        for i, m in enumerate(self.layers):
            for nm in ("wqkv", "wo", "wgu", "wd", "n1", "n2"):
                yield f"{i}.{nm}", getattr(m, nm)
        yield "norm", self.norm
        yield "lm_head", self.lm_head


class SPILlavaLlamaModel(nn.Module):
    """Holds the four stages; `forward` returns the decoder's logits."""

    def __init__(self, vision_tower: Optional[ClipVisionTower], llama: LlamaDecoder, token_ids: SimpleNamespace,
                 embed_dims=1024, mm_projector: Optional[nn.Linear] = None):
        super().__init__()
        self.num_level_spi_features = 4
        self.vision_tower = [vision_tower]          # a list, as in the reference (llava.py:48): not in the state_dict
        self.llama = llama
        self.config = token_ids                     # im_patch_token, im_start_token, im_end_token, bbox_token
        # the 16-bit storage type of the whole path follows the decoder's: bf16 (training dtype) or fp16 (the reference's
        # serving dtype, app.py:74-98: model, boxes :271 and image :296 are all .half())
        self.dtype = getattr(llama, "dtype", torch.bfloat16)
        assert vision_tower is None or vision_tower.dtype == self.dtype, "vision tower and decoder must share the storage type"
        self.spi_module = MLVLROIQueryModule(embed_dims=embed_dims, out_dims=llama.hidden, num_levels=4)
        self.spi_module.set_compute_dtype(self.dtype)
        self.mm_projector = mm_projector if mm_projector is not None else nn.Linear(embed_dims, llama.hidden)
        self._proj = None
        self._stamp = None
        self.last_status = None
        self.gradient_checkpointing = False
        self.llama_master = None                    # stage 2 through autograd: enable_decoder_training()
        self._llama_stamp = None

    # ---- kernel-ready copies of the nn.Parameters, refreshed when a parameter changed -------------------------
    def _param_stamp(self):
        return tuple((p.data_ptr(), p._version) for p in itertools.chain(self.spi_module.parameters(),
                                                                         self.mm_projector.parameters()))

    def prepare(self):
        bf = self.dtype
        dev = self.llama.device
        self.spi_module.to(dev)
        self.mm_projector.to(dev)
        self.spi_module.prepare()
        self._proj = (self.mm_projector.weight.detach().to(device=dev, dtype=bf).contiguous(),
                      self.mm_projector.bias.detach().to(device=dev, dtype=bf).float().contiguous())
        self._stamp = self._param_stamp()

    def _maybe_prepare(self):
        """An optimizer step (or load_state_dict) bumps the parameters' version counters: re-derive the bf16 copies."""
        if self._proj is None or self._stamp != self._param_stamp():
            self.prepare()
        if self.llama_master is not None:
            live = self.llama.trainable_tensors()
            for k, prm in self.llama_master.named_kernel_tensors():          # (a table replaced behind the masters' back)
                if tuple(live[k].shape) != tuple(prm.shape):
                    raise RuntimeError(f"decoder tensor {k} changed shape {tuple(prm.shape)} -> {tuple(live[k].shape)} after "
                                       "enable_decoder_training(): call enable_decoder_training() again (and re-create the optimizer)")
            st = tuple((p.data_ptr(), p._version) for _, p in self.llama_master.named_kernel_tensors())
            if st != self._llama_stamp:
                self.sync_decoder_from_masters()

    # ---- stage 2 under autograd ----------------------------------------------------------------------------------------------
    def enable_decoder_training(self):
        """Expose the decoder's weights as fp32 nn.Parameters (`llama_master.*` in named_parameters(), requires_grad=True: the
        state the reference's model is in when train_stage2.sh starts; the stage-1 freeze loop of train.py:685-697 turns them
        off again by name).  27 GB of masters for the 7 B model, on top of which a torch optimizer keeps its own state."""
        if self.llama_master is None:
            self.llama.prepare_training(train_weights=True)
            self.llama_master = _DecoderMasters(self.llama)
            self.sync_decoder_from_masters()
        return self.llama_master

    @torch.no_grad()
    def sync_decoder_from_masters(self):
        """bf16 kernel tensors <- fp32 masters (one rounding, as FullTrainer's fused AdamW writes them), W^T refreshed"""
        live = self.llama.trainable_tensors()
        for k, prm in self.llama_master.named_kernel_tensors():
            t = live[k]
            if t.data_ptr() != prm.data_ptr():
                t.copy_(prm.data)
        self.llama.refresh_transposes()
        self._llama_stamp = tuple((p.data_ptr(), p._version) for _, p in self.llama_master.named_kernel_tensors())

    @torch.no_grad()
    def decoder_param_grads(self, input_ids):
        """After backward(): 'llama.<kernel tensor>' -> fp32 gradient of every decoder weight (LlamaDecoder.backward left them
        in `llama.grads`) + the embedding rows: every position that took embed[id] in the splice (not <im_patch>, not <bbox>)."""
        dec, cfg = self.llama, self.config
        out = {f"llama.{k}": g for k, g in dec.grads.items()}
        flat = input_ids.reshape(-1)
        idx = torch.where((flat == cfg.im_patch_token) | (flat == cfg.bbox_token), torch.full_like(flat, -1), flat)
        ge = torch.zeros(dec.embed.shape, dtype=torch.float32, device=dec.embed.device)
        K.scatter_add_rows(self._d_emb, idx.to(torch.int32).contiguous(), ge)
        out["llama.embed_tokens"] = ge
        dec.grads = {}
        return out

    def initialize_vision_modules(self, vision_tower, mm_vision_select_layer=-2, pretrain_mm_mlp_adapter=None,
                                  tune_mm_mlp_adapter=False):
        """llava/model/llava.py:54-86.  `vision_tower`: a directory holding an HF CLIPVisionModel checkpoint, an HF-keyed
        state dict, or a ready ClipVisionTower (there is no hub access here, so a hub name cannot be resolved).  The
        tower stays frozen and outside the state_dict; `mm_projector` is (re)loaded from `pretrain_mm_mlp_adapter`."""
        from . import checkpoint as ckpt
        tower = vision_tower if isinstance(vision_tower, ClipVisionTower) else \
            ckpt.load_vision_tower(vision_tower, device=self.llama.device, select_layer=mm_vision_select_layer)
        self.vision_tower = [tower]
        cfg = self.config
        cfg.mm_vision_tower = vision_tower if isinstance(vision_tower, str) else getattr(cfg, "mm_vision_tower", None)
        cfg.use_mm_proj = True
        cfg.mm_hidden_size = tower.hidden
        cfg.mm_vision_select_layer = mm_vision_select_layer
        if self.mm_projector.in_features != tower.hidden:
            self.mm_projector = nn.Linear(tower.hidden, self.llama.hidden)
        if pretrain_mm_mlp_adapter is not None:
            w = pretrain_mm_mlp_adapter if isinstance(pretrain_mm_mlp_adapter, dict) else \
                torch.load(pretrain_mm_mlp_adapter, map_location='cpu')
            self.mm_projector.load_state_dict({k.split('.')[-1]: v for k, v in w.items() if 'mm_projector' in k})
        self._proj = None
        image_size = getattr(tower, "image_size", None)
        num_patches = (image_size // tower.patch) ** 2 if image_size else None
        return dict(image_processor=ckpt.ClipImageProcessor(image_size or 224), image_token_len=num_patches,
                    vision_config=cfg)

    # ---- stages a4-a15 ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def embed_inputs(self, input_ids, images=None, bboxes=None):
        """Stages a4-a15 of SURVEY.md 8a: returns inputs_embeds [B,T,C] bf16 for the decoder."""
        self._maybe_prepare()
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
                    bboxes = PreparedBoxes(bboxes, images.size(-1), images.device, dtype=self.dtype)
                feats = self.spi_module(mlvl, bboxes)
                spi = torch.cat(feats, 0).contiguous() if len(feats) > 1 else feats[0]
                off = bboxes.offsets
            n_patch = image_features.size(1)
            # mm_projector over the patch tokens (strided view of the hidden state, CLS skipped)
            img_tok = torch.empty((B, n_patch, self.llama.hidden), dtype=self.dtype, device=images.device)
            for b in range(B):
                K.gemm(image_features[b], self._proj[0], bias=self._proj[1], out=img_tok[b])
        embeds, status = K.splice_embed(input_ids.contiguous(), self.llama.embed, img_tok, spi, off, n_patch,
                                        cfg.im_patch_token, cfg.bbox_token, cfg.im_start_token, cfg.im_end_token)
        self.last_status = status
        return embeds

    def prepare_boxes(self, bboxes, image_size):
        """Do the host-side part of a request once (see layers.PreparedBoxes); the returned object can be
        passed as `bboxes=` and makes forward() free of host<->device traffic (hipGraph-capturable)."""
        return PreparedBoxes(bboxes, image_size, self.llama.device, dtype=self.dtype)

    def clone_context(self):
        """A second request context: shares every weight / prepared buffer with `self`, owns its KV
        cache and status.  Two contexts driven on two HIP streams keep the 256 CUs busy through the
        wave-quantisation tails and the latency-bound small kernels of a batch-1 request
        (+30 % region-tokens/s on MI355X)."""
        import copy
        self._maybe_prepare()
        other = copy.copy(self)                      # nn.Module shallow copy: parameters are shared
        other.llama = copy.copy(self.llama)
        other.llama._alloc_cache(self.llama.kc.size(1))
        other.llama._attn_ws = None                  # per-context decode workspaces (the contexts run on different streams)
        other.llama._dstate = None
        other.llama._bstate = None
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

    # ---- training rows: the forward that keeps what the backward needs, and that backward ----------------------
    @torch.no_grad()
    def forward_train(self, input_ids, images, bboxes, attention_mask=None):
        """-> (logits fp32 [B*T, V], ctx).  Raises on a malformed prompt (per-sample <bbox>/region count,
        <im_start>/<im_end>) exactly where the reference does (spi_llava.py:115-157).  attention_mask: the collator's
        key padding mask (data_modules.py:22-56), handed to the decoder as the reference's flash-attention patch receives it
        (llama_flash_attn_monkey_patch.py:60-85); right padding takes the dense path (same real rows, same gradients)."""
        self._maybe_prepare()
        cfg = self.config
        B, T = input_ids.shape
        tower = self.vision_tower[0]
        if isinstance(images, (list, tuple)):
            images = torch.stack(list(images), 0)
        keep = tower.forward(images)
        image_features, mlvl = tower.select(keep)
        if bboxes is None:
            bboxes = [images.new_zeros((0, 4)) for _ in range(B)]
        if not isinstance(bboxes, PreparedBoxes):
            bboxes = PreparedBoxes(bboxes, images.size(-1), images.device)
        # a batch without any region still trains the projector (the reference keeps the graph alive with a dummy zero
        # term, layers.py:314-317 / spi_llava.py:94-108); the region module then simply receives no gradient
        spi, sctx = self.spi_module.forward_train(mlvl, bboxes) if bboxes.n > 0 else (None, None)
        n_patch = image_features.size(1)
        img_tok = torch.empty((B, n_patch, self.llama.hidden), dtype=torch.bfloat16, device=images.device)
        for b in range(B):
            K.gemm(image_features[b], self._proj[0], bias=self._proj[1], out=img_tok[b])
        embeds, status = K.splice_embed(input_ids.contiguous(), self.llama.embed, img_tok, spi, bboxes.offsets, n_patch,
                                        cfg.im_patch_token, cfg.bbox_token, cfg.im_start_token, cfg.im_end_token)
        self.last_status = status
        self.check_status()
        if not hasattr(self.llama, "lm_head_t"):
            self.llama.prepare_training(train_weights=False)
        self.llama.reset(B)
        logits, lctx = self.llama.forward_train(embeds, checkpoint=self.gradient_checkpointing,
                                                key_padding_mask=dense_or_mask(attention_mask))
        return logits, dict(sctx=sctx, lctx=lctx, input_ids=input_ids, boxes=bboxes, image_features=image_features)

    @torch.no_grad()
    def backward(self, ctx, dlogits, train_projector=False, on_grad=None, grad_slot=None):
        """dlogits bf16 [B*T, v_pad] -> {"spi_module.<key>": fp32 grad, ("mm_projector.weight"/".bias")} in the
        reference's state_dict layouts.  `on_grad(name, grad)` is called as each gradient is produced (head of the
        model first), which is what lets the bucketed exchange overlap with the rest of the backward."""
        cfg = self.config
        dec_cb = (lambda n, g: on_grad(f"llama.{n}", g)) if (on_grad is not None and self.llama.train_weights) else None
        dec_slot = (lambda n: grad_slot(f"llama.{n}")) if (grad_slot is not None and self.llama.train_weights) else None
        d_emb = self.llama.backward(ctx["lctx"], dlogits, on_grad=dec_cb, grad_slot=dec_slot)       # [B*T, C] bf16
        self._d_emb = d_emb
        flat = ctx["input_ids"].reshape(-1)
        grads = {}
        if train_projector:
            image_features = ctx["image_features"]
            idx_patch = (flat == cfg.im_patch_token).nonzero().flatten().to(torch.int32)
            d_img = K.gather_rows(d_emb, idx_patch)                             # [B*n_patch, C]
            x = image_features.reshape(-1, image_features.size(-1))
            grads["mm_projector.weight"] = K.linear_wgrad(d_img, x)
            grads["mm_projector.bias"] = K.colsum(d_img)
            if on_grad is not None:
                for k in ("mm_projector.bias", "mm_projector.weight"):
                    on_grad(k, grads[k])
        if ctx["sctx"] is None:
            # a batch without regions: the reference keeps every region-module parameter in the graph through a zero
            # dummy term (gpt4roi/models/layers.py:314-317, spi_llava.py:94-108), so the step -- and with world > 1 the
            # gradient exchange, which waits for every registered gradient on every rank -- goes on with ZERO gradients
            named = list(self.spi_module.named_parameters())
            for k, prm in reversed(named):                                      # the order the real backward reports in
                g = torch.zeros(prm.shape, dtype=torch.float32, device=d_emb.device)
                grads[f"spi_module.{k}"] = g
                if on_grad is not None:
                    on_grad(f"spi_module.{k}", g)
            return grads
        idx_bbox = (flat == cfg.bbox_token).nonzero().flatten().to(torch.int32)
        assert idx_bbox.numel() == ctx["boxes"].n, "number of <bbox> tokens != number of regions"
        cb = (lambda k, g: on_grad(f"spi_module.{k}", g)) if on_grad is not None else None
        for k, g in self.spi_module.backward(ctx["sctx"], K.gather_rows(d_emb, idx_bbox), on_grad=cb).items():
            grads[f"spi_module.{k}"] = g
        return grads

    def trainable_named(self):
        """name -> nn.Parameter for the stage-1 trainables that currently require grad (train.py:685-696)."""
        out = {f"spi_module.{k}": p for k, p in self.spi_module.named_parameters() if p.requires_grad}
        out.update({f"mm_projector.{k}": p for k, p in self.mm_projector.named_parameters() if p.requires_grad})
        if self.llama_master is not None:          # stage 2: the decoder's masters (enable_decoder_training)
            out.update({f"llama.{k}": p for k, p in self.llama_master.named_kernel_tensors() if p.requires_grad})
        return out

    def forward(self, input_ids: torch.LongTensor = None, attention_mask: Optional[torch.Tensor] = None,
                img_metas=None, bboxes=None, past_key_values=None, inputs_embeds: Optional[torch.Tensor] = None,
                use_cache: Optional[bool] = None, output_attentions: Optional[bool] = None,
                output_hidden_states: Optional[bool] = None, images: Optional[torch.Tensor] = None,
                return_dict: Optional[bool] = None, all_logits=True):
        if torch.is_grad_enabled() and inputs_embeds is None and images is not None and input_ids.size(1) != 1:
            named = self.trainable_named()
            if named:
                names = tuple(named)
                if self.llama_master is not None:   # weight gradients only when a decoder parameter asks for one
                    self.llama.train_weights = any(n.startswith("llama.") for n in names)
                return _RegionPathFn.apply(self, input_ids, images, bboxes, attention_mask, names, *named.values())
        with torch.no_grad():
            if inputs_embeds is None:
                inputs_embeds = self.embed_inputs(input_ids, images, bboxes)
            if past_key_values is None:
                self.llama.reset(inputs_embeds.size(0))
            # the prompt's mask: any pattern (HF LlamaModel semantics: masked keys unseen, positions of the padded layout);
            # on the later one-token calls HF's mask only grows by ones, which the ragged cache already encodes
            mask = attention_mask if (past_key_values is None and inputs_embeds.size(1) > 1) else None
            return self.llama.forward(inputs_embeds, all_logits=all_logits, key_padding_mask=mask)


class _EmbeddingView:
    """What `get_input_embeddings()` / `get_output_embeddings()` hand to initialize_vision_tokenizer: `.weight.data`."""

    def __init__(self, dec, attr):
        self._dec, self._attr = dec, attr

    @property
    def weight(self):
        return getattr(self._dec, self._attr)

    def parameters(self):
        return iter(())          # frozen bf16 tensors on this path (stage 2 trains them through FullTrainer)


class SPILlavaMPTForCausalLM(nn.Module):
    """Same call shape as the reference class of that name (despite "MPT" it is the LLaMA model)."""

    def __init__(self, model: SPILlavaLlamaModel, config: Optional[SimpleNamespace] = None):
        super().__init__()
        self.model = model
        self.config = config if config is not None else SimpleNamespace()
        self.generation_config = SamplingConfig()

    def get_model(self):
        return self.model

    def get_input_embeddings(self):
        return _EmbeddingView(self.model.llama, "embed")

    def get_output_embeddings(self):
        return _EmbeddingView(self.model.llama, "lm_head")

    def gradient_checkpointing_enable(self, *_, **__):
        """`--gradient_checkpointing True` (train_stage1.sh:36): keep only each decoder layer's input and re-run the
        layer in the backward (LlamaDecoder.forward_train(checkpoint=True))."""
        self.model.gradient_checkpointing = True

    def gradient_checkpointing_disable(self):
        self.model.gradient_checkpointing = False

    def enable_decoder_training(self):
        """Stage 2 (train_stage2.sh) through the reference's own caller: after this call `named_parameters()` lists the LLaMA
        weights (`model.llama_master.*`, fp32, requires_grad) beside the region module and the projector, `model(**batch).loss
        .backward()` fills their `.grad` with the hand-written backward's weight gradients, and any torch optimizer steps them;
        the stage-1 freeze loop (train.py:685-697: requires_grad = 'spi_module' in name) turns them off by name.  Call it once,
        after `resize_token_embeddings` / `initialize_vision_tokenizer` (the masters are built from the final tables)."""
        return self.model.enable_decoder_training()

    def resize_token_embeddings(self, new_num_tokens):
        """HF `resize_token_embeddings`: grow (or cut) the embedding table and lm_head; new rows are zero until
        initialize_vision_tokenizer gives them the mean of the old rows (spi_llava.py:262-272)."""
        dec = self.model.llama
        old = dec.embed.size(0)
        if new_num_tokens == old:
            return self.get_input_embeddings()
        for attr in ("embed", "lm_head"):
            w = getattr(dec, attr)
            nw = torch.zeros((new_num_tokens, w.size(1)), dtype=w.dtype, device=w.device)
            n = min(old, new_num_tokens)
            nw[:n] = w[:n]
            setattr(dec, attr, nw)
        dec.vocab = new_num_tokens
        if hasattr(dec, "lm_head_t"):
            dec.prepare_training(train_weights=getattr(dec, "train_weights", False))
        dec._dstate = None
        dec._bstate = None
        self.config.vocab_size = new_num_tokens
        if self.model.llama_master is not None:
            # the fp32 masters of stage 2 were built from the OLD tables; left alone, the next optimizer step would
            # write wrongly-shaped rows back.  Rebuild them from the resized tensors (an optimizer created before this call holds
            # the old Parameters and must be re-created, as with HF's own resize after optimizer construction).
            self.model.llama_master = None
            self.model._llama_stamp = None
            self.model.enable_decoder_training()
        return self.get_input_embeddings()

    def initialize_vision_tokenizer(self, mm_use_im_start_end, tokenizer, device=None, tune_mm_mlp_adapter=False,
                                    pretrain_mm_mlp_adapter=None):
        """gpt4roi/models/spi_llava.py:242-306, same order of operations: add <im_patch>, resize; add <bbox>, <point>;
        add <im_start>, <im_end>, resize; the last `num_new_tokens` (= 2 + the spatial tokens) rows of the input and
        output embeddings become the mean of the rows before them; ids are written into the vision config; the
        tokenizer is attached to every module."""
        vision_config = self.model.config
        vision_config.use_im_start_end = mm_use_im_start_end
        tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
        self.resize_token_embeddings(len(tokenizer))
        tokenizer, num_spi_tokens = add_spatial_token(tokenizer)
        if mm_use_im_start_end:
            num_new_tokens = tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
            self.resize_token_embeddings(len(tokenizer))
            vision_config.im_start_token, vision_config.im_end_token = tokenizer.convert_tokens_to_ids(
                [DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN])
            num_new_tokens = num_new_tokens + num_spi_tokens
            if num_new_tokens > 0:
                input_embeddings = self.get_input_embeddings().weight.data
                output_embeddings = self.get_output_embeddings().weight.data
                input_embeddings_avg = input_embeddings[:-num_new_tokens].float().mean(dim=0, keepdim=True)
                output_embeddings_avg = output_embeddings[:-num_new_tokens].float().mean(dim=0, keepdim=True)
                input_embeddings[-num_new_tokens:] = input_embeddings_avg.to(input_embeddings.dtype)
                output_embeddings[-num_new_tokens:] = output_embeddings_avg.to(output_embeddings.dtype)
            if tune_mm_mlp_adapter:
                self.model.orig_embeds_params = [self.get_input_embeddings().weight.data.clone()]
            if pretrain_mm_mlp_adapter:
                w = pretrain_mm_mlp_adapter if isinstance(pretrain_mm_mlp_adapter, dict) else \
                    torch.load(pretrain_mm_mlp_adapter, map_location='cpu')
                embed_tokens_weight = w['model.embed_tokens.weight']
                input_embeddings = self.get_input_embeddings().weight.data
                num_new_tokens = num_new_tokens - num_spi_tokens
                if input_embeddings.shape == embed_tokens_weight.shape:
                    input_embeddings[-num_new_tokens:] = embed_tokens_weight[-num_new_tokens:].to(input_embeddings)
                elif embed_tokens_weight.shape[0] == num_new_tokens:
                    input_embeddings[-num_new_tokens:] = embed_tokens_weight.to(input_embeddings)
                else:
                    raise ValueError(f'Unexpected embed_tokens_weight shape. Pretrained: {embed_tokens_weight.shape}. '
                                     f'Current: {input_embeddings.shape}. Numer of new tokens: {num_new_tokens}.')
            if hasattr(self.model.llama, "lm_head_t"):
                self.model.llama.refresh_transposes()
        vision_config.im_patch_token = tokenizer.convert_tokens_to_ids([DEFAULT_IMAGE_PATCH_TOKEN])[0]
        vision_config.bbox_token = tokenizer.convert_tokens_to_ids(['<bbox>'])[0]
        vision_config.point_token = tokenizer.convert_tokens_to_ids(['<point>'])[0]
        for m in self.modules():          # "broadcast the tokenizer to all modules" (:304-306)
            m.tokenizer = tokenizer

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        from . import checkpoint as ckpt
        return ckpt.from_pretrained(cls, path, **kwargs)

    def save_pretrained(self, path, **kwargs):
        from . import checkpoint as ckpt
        return ckpt.save_pretrained(self, path, **kwargs)

    def state_dict(self, *args, **kwargs):
        """The HF-keyed state dict the reference checkpoints carry (`model.layers.*`, `lm_head.weight`,
        `model.spi_module.*`, `model.mm_projector.*`; no vision tower: llava.py:48 keeps it in a Python list)."""
        if self.model.llama_master is not None:
            self.model._maybe_prepare()            # the kernel tensors follow the masters an optimizer may just have stepped
        sd = self.model.llama.export_hf_state_dict()
        sd.update({f"model.spi_module.{k}": v.detach() for k, v in self.model.spi_module.state_dict().items()})
        sd.update({f"model.mm_projector.{k}": v.detach() for k, v in self.model.mm_projector.state_dict().items()})
        return sd

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None,
                                      **kwargs):
        return prepare_inputs_for_generation(input_ids, past_key_values, attention_mask, inputs_embeds, **kwargs)

    def forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, output_attentions=None, output_hidden_states=None, images=None, return_dict=None,
                *, img_metas=None, bboxes=None, _last_only=False):
        logits = self.model(input_ids=input_ids, attention_mask=attention_mask, img_metas=img_metas, bboxes=bboxes,
                            past_key_values=past_key_values, inputs_embeds=inputs_embeds, images=images,
                            all_logits=not _last_only)
        loss = None
        if labels is not None:
            # shifted-label token cross entropy, llava/model/llava.py:240-252
            if logits.requires_grad:
                loss = _ShiftedCrossEntropyFn.apply(logits, labels, self.model.llama)
            else:
                with torch.no_grad():
                    B, T, V = logits.shape
                    lab = torch.full((B, T), -100, dtype=torch.int64, device=logits.device)
                    lab[:, :-1] = labels[:, 1:]
                    lab = lab.reshape(-1).contiguous()
                    cnt = ((lab >= 0) & (lab < V)).sum().clamp(min=1).float()
                    loss_sum = torch.zeros(1, dtype=torch.float32, device=logits.device)
                    K.cross_entropy(logits.view(B * T, V), lab, loss_sum)
                    loss = (loss_sum / cnt).reshape(())
        return CausalLMOutputWithPast(logits=logits, past_key_values=self.model.llama, loss=loss)

    @torch.no_grad()
    def generate(self, input_ids=None, images=None, max_new_tokens=64, do_sample=None, temperature=None, top_k=None,
                 top_p=None, stopping_criteria=None, eos_token_id=None, seed=None, bboxes=None, stop_ids=(),
                 check_every=None, return_new_tokens=False, attention_mask=None, pad_token_id=None, **_):
        """HF-style `generate`.  Batch 1 (what gpt4roi/app.py:293-300 calls): prefill through `self.forward` -- so a
        `partial(orig_forward, img_metas=..., bboxes=...)` bound by the caller is honoured (app.py:286-291; `bboxes=`
        may also be passed here) -- then the device-resident decode loop.  do_sample=False: greedy (the parity mode);
        do_sample=True: temperature / top-k / top-p sampling on the device with a Philox stream keyed by `seed`
        (drawn from torch's generator when None).  `stopping_criteria`: callables criteria(ids [1, T+n], scores) -> bool
        evaluated per generated token like HF (KeywordsStoppingCriteria); `eos_token_id` / `stop_ids` end the sequence
        (inclusive).  Returns the full sequence LongTensor [1, T + n] (or the list of new ids if return_new_tokens).
        Batch B > 1 (greedy only): prompts padded to one length with `attention_mask` marking the real tokens (left padding
        as HF batch generation uses it, or right padding) decode together, every sequence continuing from its own length
        with the positions of the padded layout (LlamaDecoder.forward); rows that stopped are filled with pad_token_id."""
        if input_ids.size(0) > 1:
            return self._generate_batch(input_ids, images, max_new_tokens, do_sample, stopping_criteria, eos_token_id,
                                        bboxes, stop_ids, return_new_tokens, attention_mask, pad_token_id)
        gc = self.generation_config
        cfg = SamplingConfig(do_sample=gc.do_sample if do_sample is None else bool(do_sample),
                             temperature=gc.temperature if temperature is None else temperature,
                             top_k=gc.top_k if top_k is None else top_k, top_p=gc.top_p if top_p is None else top_p)
        sampler = cfg.sampler()
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if sampler is not None else 0
        fwd = self.forward                                   # instance attribute first: the caller may have rebound it
        kw = dict(bboxes=bboxes) if bboxes is not None else {}
        model_inputs = self.prepare_inputs_for_generation(input_ids, images=images)
        dec = self.model.llama
        # step 0 = prefill through the (possibly partial-bound) forward; only its embeddings are needed here: the
        # decode loop owns the KV cache, so hand it the spliced embeddings
        embeds = self._prefill_embeds(fwd, model_inputs, kw)
        stops = set(int(s) for s in stop_ids)
        if eos_token_id is None:                            # HF generate stops on the config's EOS id by default
            eos_token_id = getattr(self.config, "eos_token_id", None)
        if eos_token_id is not None:
            stops.update([int(eos_token_id)] if not isinstance(eos_token_id, (list, tuple)) else map(int, eos_token_id))
        on_tokens = None
        if stopping_criteria:
            for c in stopping_criteria:                      # HF calls every criterion once per generated token; the
                c(input_ids, None)                           # reference's first call only records the prompt length
            prompt = input_ids

            def on_tokens(new_ids):
                seq = torch.cat([prompt, torch.tensor([new_ids], dtype=prompt.dtype, device=prompt.device)], 1)
                return any(bool(c(seq, None)) for c in stopping_criteria)
            # the first call above consumed HF's "first token" call of the reference criteria (start_len recorded),
            # so token 1 is tested like every later one
        every = check_every if check_every is not None else (1 if stopping_criteria else 32)
        new = dec.decode_graph(embeds, max_new_tokens, stop_ids=stops, check_every=every, sampler=sampler, seed=seed,
                               on_tokens=on_tokens)
        if return_new_tokens:
            return new
        return torch.cat([input_ids, torch.tensor([new], dtype=input_ids.dtype, device=input_ids.device)], 1)

    def _generate_batch(self, input_ids, images, max_new_tokens, do_sample, stopping_criteria, eos_token_id, bboxes,
                        stop_ids, return_new_tokens, attention_mask, pad_token_id):
        if (self.generation_config.do_sample if do_sample is None else do_sample) or stopping_criteria:
            raise NotImplementedError("batched generate() is greedy and stops on ids (sampling / criteria: one request at a time)")
        kw = dict(bboxes=bboxes) if bboxes is not None else {}
        embeds = self._prefill_embeds(self.forward, self.prepare_inputs_for_generation(input_ids, images=images), kw)
        stops = set(int(s) for s in stop_ids)
        if eos_token_id is None:
            eos_token_id = getattr(self.config, "eos_token_id", None)
        if eos_token_id is not None:
            stops.update([int(eos_token_id)] if not isinstance(eos_token_id, (list, tuple)) else map(int, eos_token_id))
        new = self.model.llama.decode_graph_batch(embeds, max_new_tokens, stop_ids=stops, key_padding_mask=attention_mask)
        if return_new_tokens:
            return new
        pad = pad_token_id if pad_token_id is not None else (getattr(self.config, "pad_token_id", None) or 0)
        n = max(len(r) for r in new)
        tail = torch.tensor([r + [pad] * (n - len(r)) for r in new], dtype=input_ids.dtype, device=input_ids.device)
        return torch.cat([input_ids, tail], 1)

    def _prefill_embeds(self, fwd, model_inputs, kw):
        """The spliced prompt embeddings of step 0.  When `forward` was rebound with partial(bboxes=...) the boxes live
        in its keywords (app.py:286-291); they are read from there, the rest of the call is `embed_inputs`."""
        bound = getattr(fwd, "keywords", None) or {}
        bboxes = kw.get("bboxes", bound.get("bboxes"))
        return self.model.embed_inputs(model_inputs["input_ids"], model_inputs.get("images"), bboxes)
