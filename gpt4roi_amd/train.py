"""One training step of the region-feature path on the gfx950 kernels.

What the reference runs per step under HF Trainer (/root/reference/gpt4roi/train/train.py:698-712 with
train_stage1.sh: `ONLY_SPI=1`, bf16, per-device batch 1, AdamW lr 2e-5, weight decay 0, cosine schedule, warm-up
ratio 0.003, max_grad_norm 1.0 (HF default), DDP over the GPUs of one node):

    forward (ViT frozen -> region module -> projector -> splice -> LLaMA frozen -> lm_head -> shifted CE)
    backward to the trainable parameters (autograd), gradient all-reduce (DDP), clip, AdamW.

Here the same step is an explicit launch sequence: `forward_train` of every stage keeps what its hand-written
backward needs, the frozen decoder only propagates the activation gradient (W^T GEMMs, flash-attention backward),
the region module produces its parameter gradients in the reference's state_dict layout, the exchange step is
grad_reduce.GradBucketReducer (reduce-scatter + all-gather over xGMI, one process per GPU), and AdamW is one fused
kernel per tensor on the fp32 master weights (the nn.Parameters themselves, so checkpoints keep the reference's
keys).  There is no autograd graph and no CPU fallback.
"""
import math

import torch
import torch.distributed as dist

from . import kernels as K
from .fsdp import FullShardManager
from .sharded import ShardedAdamW
from .grad_reduce import GradBucketReducer
from .layers import PreparedBoxes


def cosine_lr(step, total_steps, base_lr, warmup_ratio=0.003):
    """HF `get_cosine_schedule_with_warmup` (lr_scheduler_type "cosine", train_stage1.sh): step counts from 0."""
    warm = math.ceil(total_steps * warmup_ratio)
    if step < warm:
        return base_lr * step / max(1, warm)
    prog = (step - warm) / max(1, total_steps - warm)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))


def exchange_gradients(reducer, tensors, grads):
    """Average `grads` (name -> tensor) over the ranks with a GradBucketReducer built over `tensors` (name -> the
    parameter tensor each gradient belongs to).  Gradients are reported in reverse registration order -- the order
    the backward produces them (head of the model first) -- so that the first buckets are on the wire while the
    rest is still being computed.  Returns name -> averaged gradient (views into the reducer's flat buckets).
    Backend agnostic: RCCL on the node, gloo in tests/test_train_exchange_gloo.py."""
    names = list(tensors)
    reducer.reset()
    for k in reversed(names):
        reducer.ready(tensors[k], grads[k])
    red = reducer.finish()
    return {k: red[id(tensors[k])] for k in names}


class RegionTrainer:
    """Stage-1 trainer: `model.spi_module` (and optionally `model.mm_projector`) are updated, the vision tower and
    the decoder stay frozen.  `step()` returns the mean token loss as a device tensor."""

    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0,
                 train_projector=False, group=None, bucket_bytes=256 << 20, _build_reducer=True, exchange_algo="rs_ag"):
        """exchange_algo: "rs_ag" (reduce-scatter + all-gather per bucket: every xGMI link carries 1 / world of it) or
        "all_reduce" (one ring all-reduce per bucket, the measured alternative) -- grad_reduce.GradBucketReducer."""
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.train_projector = train_projector
        dev = model.llama.device
        model.spi_module.to(dev)
        model.mm_projector.to(dev)
        model.prepare()
        model.llama.prepare_training(train_weights=False)
        self.params = {f"spi_module.{k}": p for k, p in model.spi_module.named_parameters()}
        if train_projector:
            self.params.update({f"mm_projector.{k}": p for k, p in model.mm_projector.named_parameters()})
        for p in self.params.values():
            assert p.dtype == torch.float32 and p.is_contiguous()
            p.requires_grad_(True)
        self.opt = K.MultiTensorAdamW([p.data for p in self.params.values()], None, betas, eps, weight_decay)
        self.steps = 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.reducer = GradBucketReducer(list(self.params.values()), bucket_bytes=bucket_bytes, group=group, comm_dtype=torch.float32,
                                         algo=exchange_algo) if (self.world > 1 and _build_reducer) else None
        self.last_grad_norm = None

    # ---- forward + backward: parameter gradients in the reference layout ------------------------------------
    def _exchange_tensors(self):
        """name -> the tensor object the reducer was built over (nn.Parameters here; FullTrainer adds its masters)."""
        return self.params

    @torch.no_grad()
    def loss_and_grads(self, input_ids, images, bboxes, labels, exchange=False, attention_mask=None):
        """One forward + backward.  exchange=True (world > 1): every gradient is handed to the bucketed reducer the
        moment its kernel sequence has been issued (`on_grad`), so a bucket's reduce-scatter + all-gather runs on the
        communication stream while the backward of the earlier layers is still executing; the returned gradients are
        then the rank-averaged views into the reducer's flat buckets."""
        m = self.model
        logits, ctx = m.forward_train(input_ids, images, bboxes, attention_mask=attention_mask)
        loss, dlogits = m.llama.loss_and_dlogits(logits, labels)
        on_grad = None
        live = exchange and self.reducer is not None
        if live:
            tensors = self._exchange_tensors()
            self.reducer.reset()
            on_grad = lambda name, g: self.reducer.ready(tensors[name], g.reshape(tensors[name].shape))  # noqa: E731
            slot = lambda name: self.reducer.slot(tensors[name]) if name in tensors else None           # noqa: E731
        grads = m.backward(ctx, dlogits, train_projector=self.train_projector, on_grad=on_grad, grad_slot=slot if live else None)
        self._d_emb = m._d_emb
        self._last_input_ids = input_ids
        grads = self._extra_grads(grads, on_grad)
        if live:
            red = self.reducer.finish()
            grads = {k: red[id(t)] for k, t in tensors.items()}
        return loss, grads

    def _extra_grads(self, grads, on_grad):
        return grads

    # ---- exchange, clip, AdamW ---------------------------------------------------------------------------------
    @torch.no_grad()
    def apply(self, grads, lr=None, exchanged=False):
        """clip_grad_norm_(max_grad_norm) + AdamW over every trainable tensor in two launches (kernels.MultiTensorAdamW):
        the global norm and the clip coefficient stay on the device (`last_grad_norm` is a device tensor; reading it is
        the only host sync, and nothing here does)."""
        names = list(self.params)
        if self.reducer is not None and not exchanged:
            grads = exchange_gradients(self.reducer, self.params, grads)
        gl = [grads[k].reshape(self.params[k].shape) for k in names]
        gl = [g if g.is_contiguous() else g.contiguous() for g in gl]
        self.steps += 1
        total_sq = self.opt.step(gl, self.lr if lr is None else lr, self.max_grad_norm)
        self.last_grad_norm = total_sq.sqrt() if total_sq is not None else None
        self.model.prepare()                                # refresh the bf16 kernel copies of the updated weights

    def state_dict(self):
        """Optimizer state under the reference's parameter names (resume: train.py:708-712 reloads optimizer.pt)."""
        sd = self.opt.state_dict()
        names = list(self._all_names())
        return {"step": sd["step"], "exp_avg": dict(zip(names, sd["exp_avg"])), "exp_avg_sq": dict(zip(names, sd["exp_avg_sq"]))}

    def load_state_dict(self, sd):
        names = list(self._all_names())
        self.opt.load_state_dict({"step": sd["step"], "exp_avg": [sd["exp_avg"][k] for k in names],
                                  "exp_avg_sq": [sd["exp_avg_sq"][k] for k in names]})
        self.steps = int(sd["step"])

    def _all_names(self):
        return self.params.keys()

    def step(self, input_ids, images, bboxes, labels, lr=None, attention_mask=None):
        loss, grads = self.loss_and_grads(input_ids, images, bboxes, labels, exchange=self.reducer is not None,
                                          attention_mask=attention_mask)
        self.apply(grads, lr, exchanged=self.reducer is not None)
        return loss


class FullTrainer(RegionTrainer):
    """Stage 2 (train_stage2.sh: everything but the vision tower is trained): on top of the
    stage-1 step the decoder produces every weight gradient (NT GEMMs over the token axis), the embedding table
    gets a scatter-add of d(inputs_embeds), and AdamW runs on fp32 masters of the decoder's kernel-layout tensors
    (fused q|k|v, interleaved gate/up; `LlamaDecoder.export_hf_state_dict()` gives the HF layout back) writing the
    bf16 copy the kernels read in the same pass.  Memory for the 7B model: 13.5 GB bf16 weights + 13.5 GB W^T +
    27 GB masters + 54 GB Adam moments + 27 GB fp32 gradients -- a replica per GPU fits the 288 GB of an MI355X,
    which is why the exchange is a plain gradient all-reduce rather than the reference's FSDP sharding
    (train_stage2.sh:51-52)."""

    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, group=None,
                 bucket_bytes=256 << 20, exchange_algo="rs_ag"):
        super().__init__(model, lr, betas, eps, weight_decay, max_grad_norm, train_projector=True, group=group,
                         bucket_bytes=bucket_bytes, _build_reducer=False)   # one reducer over ALL tensors, below
        dec = model.llama
        dec.prepare_training(train_weights=True)
        self.dec_live = {f"llama.{k}": v for k, v in dec.trainable_tensors().items()}
        # fp32 masters (norm weights are fp32 already and are their own master)
        self.dec_master = {k: (v if v.dtype == torch.float32 else v.float()) for k, v in self.dec_live.items()}
        # ONE fused optimizer over the region module / projector parameters and the decoder masters; the decoder's bf16
        # kernel tensors are written by the same kernel (no separate refresh pass)
        masters = [p.data for p in self.params.values()] + list(self.dec_master.values())
        copies = [None] * len(self.params) + [(self.dec_live[k] if self.dec_live[k].dtype == torch.bfloat16 else None)
                                               for k in self.dec_master]
        self.opt = K.MultiTensorAdamW(masters, copies, betas, eps, weight_decay)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.reducer = None
        if self.world > 1:
            tensors = list(self.params.values()) + list(self.dec_master.values())
            self.reducer = GradBucketReducer(tensors, bucket_bytes=bucket_bytes, group=group, comm_dtype=torch.float32,
                                             trainable_only=False, algo=exchange_algo)

    def _exchange_tensors(self):
        return {**self.params, **self.dec_master}

    def _extra_grads(self, grads, on_grad):
        """Decoder weight gradients (produced inside LlamaDecoder.backward and already reported through `on_grad`) and the
        embedding rows: every position that took embed[id] in the splice (not <im_patch>, not <bbox>)."""
        m = self.model
        dec, cfg = m.llama, m.config
        for k, g in dec.grads.items():
            grads[f"llama.{k}"] = g
        flat = self._last_input_ids.reshape(-1)
        idx = torch.where((flat == cfg.im_patch_token) | (flat == cfg.bbox_token), torch.full_like(flat, -1), flat)
        ge = torch.zeros(dec.embed.shape, dtype=torch.float32, device=dec.embed.device)
        K.scatter_add_rows(self._d_emb, idx.to(torch.int32).contiguous(), ge)
        grads["llama.embed_tokens"] = ge
        if on_grad is not None:
            on_grad("llama.embed_tokens", ge)
        return grads

    def _all_names(self):
        return list(self.params) + list(self.dec_master)

    def state_dict(self):
        """Optimizer moments (reference parameter names for the region module / projector, `llama.<kernel tensor>` for the
        decoder) PLUS the decoder's fp32 master weights: `export_hf_state_dict()` only carries their bf16 roundings, so a
        resume from it alone would lose the master precision."""
        sd = super().state_dict()
        sd["masters"] = {k: v.clone() for k, v in self.dec_master.items()}
        return sd

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        for k, v in sd.get("masters", {}).items():
            self.dec_master[k].copy_(v)
            live = self.dec_live[k]
            if live.dtype == torch.bfloat16:
                live.copy_(v)                        # the kernels read the bf16 rounding of the master
        self.model.llama.refresh_transposes()

    @torch.no_grad()
    def apply(self, grads, lr=None, exchanged=False):
        names = list(self.params) + list(self.dec_master)
        tensors = {**{k: p.data for k, p in self.params.items()}, **self.dec_master}
        if self.reducer is not None and not exchanged:
            grads = exchange_gradients(self.reducer, {**self.params, **self.dec_master}, grads)
        gl = [grads[k].reshape(tensors[k].shape) for k in names]
        gl = [g if g.is_contiguous() else g.contiguous() for g in gl]
        self.steps += 1
        total_sq = self.opt.step(gl, self.lr if lr is None else lr, self.max_grad_norm)
        self.last_grad_norm = total_sq.sqrt() if total_sq is not None else None
        self.model.prepare()
        self.model.llama.refresh_transposes()


class ShardedFullTrainer(FullTrainer):
    """Stage 2 with the optimizer state sharded over the data-parallel ranks (gpt4roi_amd/sharded.py; SURVEY.md 8f-3,
    the role of `--fsdp "full_shard auto_wrap"` in train_stage2.sh:51-52).  Same forward/backward as FullTrainer; the
    exchange is reduce-scatter(gradients) -> clip + AdamW on the owned 1/world slice -> in-place all-gather(parameters),
    and the fp32 master / exp_avg / exp_avg_sq exist only for that slice: 81 GB -> 81/world GB for the 7B model, which is
    what lets the per-GPU batch of config 4 grow instead of the optimizer state filling the HBM."""

    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, group=None,
                 bucket_bytes=256 << 20, update_fn=None):
        RegionTrainer.__init__(self, model, lr, betas, eps, weight_decay, max_grad_norm, train_projector=True, group=group,
                               bucket_bytes=bucket_bytes, _build_reducer=False)
        self.opt = None                                     # (the stage-1 optimizer the base class built)
        dec = model.llama
        dec.prepare_training(train_weights=True)
        entries = [(k, p.data) for k, p in self.params.items()]
        entries += [(f"llama.{k}", v) for k, v in dec.trainable_tensors().items()]
        self.reducer = None

        def rebind(name, view):
            if name.startswith("llama."):
                dec.rebind_tensor(name[len("llama."):], view)
            else:
                self.params[name].data = view

        self.sharded = ShardedAdamW(entries, rebind, bucket_bytes=bucket_bytes, group=group, betas=betas, eps=eps,
                                    weight_decay=weight_decay, update_fn=update_fn)
        self.world = self.sharded.world
        model.prepare()
        dec.refresh_transposes()

    @torch.no_grad()
    def loss_and_grads(self, input_ids, images, bboxes, labels, exchange=True, attention_mask=None):
        """Forward + backward; every gradient goes straight into its flat bucket (`ShardedAdamW.ready`), full buckets are
        reduce-scattered on the communication stream while the backward continues.  Returns (loss, None): the gradients
        live in the buckets, `apply()` consumes them."""
        m = self.model
        logits, ctx = m.forward_train(input_ids, images, bboxes, attention_mask=attention_mask)
        loss, dlogits = m.llama.loss_and_dlogits(logits, labels)
        self.sharded.reset()
        grads = m.backward(ctx, dlogits, train_projector=True, on_grad=self.sharded.ready, grad_slot=self.sharded.slot)
        self._d_emb = m._d_emb
        self._last_input_ids = input_ids
        self._extra_grads(grads, self.sharded.ready)
        m.llama.grads = {}                                  # the buckets hold them now
        return loss, None

    @torch.no_grad()
    def apply(self, grads=None, lr=None, exchanged=True):
        self.steps += 1
        total_sq = self.sharded.step(self.lr if lr is None else lr, self.max_grad_norm)
        self.last_grad_norm = total_sq.sqrt() if total_sq is not None else None
        self.model.prepare()
        self.model.llama.refresh_transposes()

    def step(self, input_ids, images, bboxes, labels, lr=None, attention_mask=None):
        loss, _ = self.loss_and_grads(input_ids, images, bboxes, labels, attention_mask=attention_mask)
        self.apply(None, lr)
        return loss

    def state_dict(self):
        """THIS rank's shard of the optimizer state (per bucket: fp32 master slice + moments), as FSDP's sharded state
        dict does; the weights themselves are whole on every rank (`save_pretrained`)."""
        return {"step": self.steps, "rank": self.sharded.rank, "world": self.sharded.world,
                "buckets": [dict(names=[n for n, _ in b.entries], master=b.master.clone(), exp_avg=b.exp_avg.clone(),
                                 exp_avg_sq=b.exp_avg_sq.clone()) for b in self.sharded.buckets]}

    def load_state_dict(self, sd):
        assert sd["world"] == self.sharded.world and sd["rank"] == self.sharded.rank, "sharded state is per (rank, world)"
        self.steps = self.sharded.steps = int(sd["step"])
        if self.sharded._fused is not None:
            self.sharded._fused.steps = self.steps
        for b, s in zip(self.sharded.buckets, sd["buckets"]):
            assert [n for n, _ in b.entries] == s["names"]
            b.master.copy_(s["master"])
            b.exp_avg.copy_(s["exp_avg"])
            b.exp_avg_sq.copy_(s["exp_avg_sq"])
        self.sharded.load_masters()                         # live bf16 / fp32 parameters <- restored masters, all ranks
        self.model.prepare()
        self.model.llama.refresh_transposes()


class FSDPFullTrainer(FullTrainer):
    """Stage 2 with parameters, gradients and optimizer state fully sharded per LlamaDecoderLayer (gpt4roi_amd/fsdp.py) --
    what `--fsdp "full_shard auto_wrap" --fsdp_transformer_layer_cls_to_wrap LlamaDecoderLayer` does in the reference's own
    launch script (train_stage2.sh:51-52).  Units, as FSDP's auto-wrap forms them: unit 1 + li = decoder layer li (wqkv, wo,
    wgu, wd, n1, n2); unit 0 = the root, everything outside the layers (region module, projector, embedding table, final
    norm, lm_head), which stays gathered for the whole step as FSDP's root module does.  The forward gathers layer li+1 on the
    communication stream while layer li computes and releases li afterwards (reshard after forward); the backward gathers
    again in reverse, makes the layer's W^T buffers from the gathered weights (four transposes into reused scratch), and
    reduce-scatters the layer's fp32 gradients as soon as its six tensors exist.  Per rank and for the 7 B decoder: 13.5 / w GB
    of bf16 shards + 81 / w GB of master and moments + 27 / w GB of gradient slices, plus ~1.6 GB of transient pool, against
    13.5 + 81 + 27 GB unsharded."""

    def __init__(self, model, lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=1.0, group=None,
                 prefetch=1, update_fn=None):
        model.llama.lazy_transposes = True                    # no W^T for all layers up front: made per gathered layer
        RegionTrainer.__init__(self, model, lr, betas, eps, weight_decay, max_grad_norm, train_projector=True, group=group,
                               _build_reducer=False)
        self.opt = None
        self.reducer = None
        dec = model.llama
        dec.prepare_training(train_weights=True)
        live = dec.trainable_tensors()
        root = [(k, p.data) for k, p in self.params.items()] + [(f"llama.{k}", live[k]) for k in ("embed_tokens", "norm", "lm_head")]
        units = [root]
        for li in range(len(dec.layers)):
            units.append([(f"llama.{li}.{nm}", live[f"{li}.{nm}"]) for nm in ("wqkv", "wo", "wgu", "wd", "n1", "n2")])
        self._scratch = None                                  # W^T buffers of the layer in the backward, allocated once
        self._empty = torch.empty(0, device=dec.device)

        def rebind(name, tensor):
            if name.startswith("llama."):
                dec.set_tensor(name[len("llama."):], tensor)
            else:
                self.params[name].data = tensor if tensor is not None else self._empty

        self.fsdp = FullShardManager(units, rebind, group=group, betas=betas, eps=eps, weight_decay=weight_decay,
                                     prefetch=prefetch, update_fn=update_fn)
        self.world = self.fsdp.world
        for L in dec.layers:                                  # stale W^T of the unsharded construction
            for nm in ("wqkv", "wo", "wgu", "wd"):
                L.pop(nm + "_t", None)
        dec.lm_head_t = None
        dec.unit_hook = self._unit_hook
        if torch.cuda.is_available():
            torch.cuda.empty_cache()                          # the unsharded construction-time tensors are gone

    def _unit_hook(self, phase, li):
        f, dec = self.fsdp, self.model.llama
        if phase == "fwd_pre":
            f.use(1 + li)
        elif phase == "fwd_post":
            f.release(1 + li)
        elif phase == "bwd_pre":
            f.use(1 + li)
            if self._scratch is None:
                L = dec.layers[li]
                self._scratch = {nm: torch.empty((L[nm].size(1), L[nm].size(0)), dtype=L[nm].dtype, device=L[nm].device)
                                 for nm in ("wqkv", "wo", "wgu", "wd")}
            dec.layer_transposes(li, self._scratch)
        else:
            for nm in ("wqkv", "wo", "wgu", "wd"):
                dec.layers[li].pop(nm + "_t", None)
            f.release(1 + li)

    @torch.no_grad()
    def loss_and_grads(self, input_ids, images, bboxes, labels, exchange=True, attention_mask=None):
        m, f = self.model, self.fsdp
        dec = m.llama
        f.begin_step()
        f.direction(+1, root=0)
        f.use(0)                                              # the root unit: region module, projector, embed, norm, lm_head
        m.prepare()                                           # kernel-ready bf16 copies of the region module / projector from
        #                                                       the gathered fp32 parameters (the pool buffer may be the same
        #                                                       address as last step: the parameter stamp cannot tell)
        dec.refresh_transposes()                              # lazy: only lm_head^T, from the gathered lm_head
        logits, ctx = m.forward_train(input_ids, images, bboxes, attention_mask=attention_mask)
        loss, dlogits = dec.loss_and_dlogits(logits, labels)
        f.direction(-1, root=0)
        grads = m.backward(ctx, dlogits, train_projector=True, on_grad=f.grad_ready, grad_slot=f.grad_slot)
        self._d_emb = m._d_emb
        self._last_input_ids = input_ids
        self._extra_grads(grads, f.grad_ready)
        dec.grads = {}
        dec.lm_head_t = None
        f.release(0)
        return loss, None

    @torch.no_grad()
    def apply(self, grads=None, lr=None, exchanged=True):
        self.steps += 1
        total_sq = self.fsdp.step(self.lr if lr is None else lr, self.max_grad_norm)
        self.last_grad_norm = total_sq.sqrt() if total_sq is not None else None

    def step(self, input_ids, images, bboxes, labels, lr=None, attention_mask=None):
        loss, _ = self.loss_and_grads(input_ids, images, bboxes, labels, attention_mask=attention_mask)
        self.apply(None, lr)
        return loss

    def full_state_dict(self):
        """name -> full tensor (gathers unit by unit and clones: for checkpoints and tests)."""
        out = {}
        for ui in range(len(self.fsdp.units)):
            for n, v in self.fsdp.full_state(ui).items():
                out[n] = v.detach().clone()
            self.fsdp.release(ui)
        return out

    # ---- checkpoints: what the reference's stage 2 ends with is an HF-named FULL state dict
    #      (safe_save_model_for_hf_trainer, gpt4roi/train/train.py:86-95) plus a resumable optimizer; under FSDP the optimizer
    #      state is saved per rank (torch FSDP's sharded state dict) ----
    def state_dict(self):
        """THIS rank's shards: per unit and dtype the parameter shard, its fp32 master and the Adam moments, + the step count.
        Resuming needs the same (rank, world) layout, as FSDP's sharded state dict does."""
        return {"step": self.steps, **self.fsdp.shard_state()}

    def load_state_dict(self, sd):
        self.fsdp.load_shard_state(sd)
        self.steps = self.fsdp.steps = int(sd["step"])
        if self.fsdp._fused is not None:
            self.fsdp._fused.steps = self.steps

    def summon_full_params(self):
        """Context manager: every unit gathered and bound, the kernel-ready copies of the region module / projector re-derived
        -- the model can then run `generate` / evaluation or be exported between two steps (torch FSDP's
        `summon_full_params`).  Costs the full 13.5 GB of bf16 weights for its duration; everything is released on exit."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            f, dec = self.fsdp, self.model.llama
            f.direction(1)
            keep, f.prefetch = f.prefetch, 0
            hook, dec.unit_hook = dec.unit_hook, None          # (the per-layer gather / release of a training step stays out of it)
            try:
                for ui in range(len(f.units)):
                    f.use(ui)
                self.model.prepare()
                dec.refresh_transposes()                       # lazy: lm_head^T only
                yield self.model
            finally:
                dec.lm_head_t = None
                for ui in range(len(f.units)):
                    f.release(ui)
                f.prefetch = keep
                dec.unit_hook = hook
        return ctx()

    def export_hf_state_dict(self):
        """The reference-format checkpoint content: HF-named FULL tensors of the decoder (q|k|v and gate/up de-fused) + the
        region module and projector under their reference keys.  Gathers one unit at a time."""
        f, dec = self.fsdp, self.model.llama
        C = dec.hidden
        out = {}
        root = f.full_state(0)
        for n, v in root.items():
            if n == "llama.embed_tokens":
                out["model.embed_tokens.weight"] = v.detach().clone()
            elif n == "llama.norm":
                out["model.norm.weight"] = v.detach().clone()
            elif n == "llama.lm_head":
                out["lm_head.weight"] = v.detach().clone()
            else:
                out["model." + n] = v.detach().clone()           # spi_module.* / mm_projector.* (self.params keys)
        f.release(0)
        for li in range(len(dec.layers)):
            t = f.full_state(1 + li)
            g = lambda nm: t[f"llama.{li}.{nm}"]                 # noqa: E731
            p = f"model.layers.{li}."
            out[p + "self_attn.q_proj.weight"] = g("wqkv")[:C].detach().clone()
            out[p + "self_attn.k_proj.weight"] = g("wqkv")[C:2 * C].detach().clone()
            out[p + "self_attn.v_proj.weight"] = g("wqkv")[2 * C:].detach().clone()
            out[p + "self_attn.o_proj.weight"] = g("wo").detach().clone()
            out[p + "mlp.gate_proj.weight"] = g("wgu")[0::2].detach().clone()
            out[p + "mlp.up_proj.weight"] = g("wgu")[1::2].detach().clone()
            out[p + "mlp.down_proj.weight"] = g("wd").detach().clone()
            out[p + "input_layernorm.weight"] = g("n1").detach().clone()
            out[p + "post_attention_layernorm.weight"] = g("n2").detach().clone()
            f.release(1 + li)
        return out

    def save_pretrained(self, path, safe_serialization=True, max_shard_bytes=5 << 30):
        """HF-layout directory of the FULL model (every rank gathers; rank 0 writes) -- what train.py:86-95 leaves behind."""
        import json as _json
        import os as _os

        from . import checkpoint as ckpt
        sd = self.export_hf_state_dict()
        if self.fsdp.rank == 0:
            _os.makedirs(path, exist_ok=True)
            ckpt.save_hf_state_dict(sd, path, safe_serialization, max_shard_bytes)
            with open(_os.path.join(path, "config.json"), "w") as fh:
                _json.dump(ckpt.model_config(type("_Wrapped", (), {"model": self.model})()), fh, indent=2)
        if dist.is_initialized() and self.fsdp.world > 1:
            dist.barrier(group=self.fsdp.group)
