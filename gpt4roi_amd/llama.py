"""LLaMA decoder stack (prefill + KV-cache decode) on the gfx950 kernels.

Carries the arithmetic the reference delegates to HF `LlamaModel.forward(inputs_embeds=...)` at
/root/reference/gpt4roi/models/spi_llava.py:198-205 and the `lm_head` of
llava/model/llava.py:235-238: RMSNorm -> fused QKV GEMM -> rotary (rotate_half) + KV-cache append
-> causal attention -> O GEMM (+residual) -> RMSNorm -> fused gate|up GEMM with SiLU*up epilogue -> down GEMM
(+residual); final RMSNorm; logits.  Weights from an HF-named state dict
(`model.layers.N.self_attn.q_proj.weight`, ..., `lm_head.weight`).  One KV cache per batch element,
sized once for `max_positions` (288 GB of HBM make whole-sequence residency the default).
"""
import math
import os

import torch

from . import kernels as K


class RaggedLayout:
    """Which positions of a padded [B, T] batch are real tokens, as the index sets the kernels consume.

    The reference's training attention unpads the batch with the mask, runs varlen flash attention over the kept tokens and
    pads the result back with zeros (llava/train/llama_flash_attn_monkey_patch.py:60-85: unpad_input -> cu_seqlens ->
    pad_input); its serving path hands the same mask to HF's LlamaModel (llava/model/llava.py:263-283).  Here every
    sequence's kept rows are gathered to the FRONT of its slot of one [B, Tc, C] buffer (Tc = the longest sequence, order
    preserved = the causal order, the slots behind a sequence's last token are zero rows), ONE causal attention launch runs
    over the B slots, and `inv` scatters the result back (pad rows = zero rows).  Under the causal mask a kept row i only sees
    rows <= i of its own slot, all of them kept rows, so the zero rows behind need no length argument: their own outputs are
    never read, and in the backward their dO rows are zero, which makes their contribution to dK / dV exactly zero.  RoPE
    has already been applied at the positions of the PADDED layout, as the reference applies it before unpadding (:46-50).
    The index sets are built on the device; building the layout reads the B lengths once on the host (for Tc)."""

    def __init__(self, mask, max_positions):
        m = mask.to(torch.bool)
        assert m.dim() == 2
        B, T = m.shape
        dev = m.device
        n = m.sum(1)
        order = torch.argsort((~m).to(torch.int8), dim=1, stable=True)          # kept positions first, in their own order
        self.B, self.T = B, T
        self.lens = [int(v) for v in n.tolist()]
        self.cu = [0]
        for v in self.lens:
            self.cu.append(self.cu[-1] + v)
        self.nnz = self.cu[-1]
        self.Tc = Tc = max(max(self.lens), 1)
        row = torch.arange(B, device=dev)[:, None]
        t_of = order[:, :Tc]
        live = torch.arange(Tc, device=dev)[None, :] < n[:, None]
        none = torch.full((), -1, dtype=torch.int64, device=dev)
        # slot (b, i) -> row of the [B*T, C] activations / of a layer's [B_alloc*maxpos, C] cache (-1: behind the sequence)
        self.idx = torch.where(live, row * T + t_of, none).to(torch.int32).reshape(-1).contiguous()
        self.idx_cache = torch.where(live, row * max_positions + t_of, none).to(torch.int32).reshape(-1).contiguous()
        rank = torch.cumsum(m, 1) - 1
        self.inv = torch.where(m, row * Tc + rank, none).to(torch.int32).reshape(-1).contiguous()   # padded row -> slot (-1: pad)
        self.pos0 = torch.cat([n, torch.full((1,), T, dtype=n.dtype, device=dev)]).to(torch.int32)   # the ragged decode state
        #                                                              after the prompt: B cache lengths + the RoPE position
        last_t = order.gather(1, (n - 1).clamp(min=0)[:, None])[:, 0]
        self.last = torch.where(n > 0, row[:, 0] * T + last_t, none).to(torch.int32)   # padded row of each sequence's last token

    @staticmethod
    def of(mask, max_positions):
        """None for `no mask` / all ones (the dense path), a layout otherwise.  A layout passes through: a serving loop
        prepares it once per batch, outside the launch sequence, which then has no host <-> device traffic and can be captured
        in a hipGraph (the constructor reads the mask on the host)."""
        if mask is None or isinstance(mask, RaggedLayout):
            return mask
        m = mask.to(torch.bool)
        if bool(m.all()):
            return None
        return RaggedLayout(m, max_positions)


class LlamaDecoder:
    def __init__(self, state_dict, heads, eps=1e-6, theta=10000.0, max_positions=2048, device="cuda",
                 num_layers=None, max_batch=1, dtype=torch.bfloat16):
        """dtype: the 16-bit storage type of weights, activations and the KV cache -- torch.bfloat16 (training dtype,
        train_stage1.sh:19) or torch.float16 (the reference's serving dtype, app.py:74-98); fp32 accumulation either way.
        Training (forward_train / backward) is bf16 only."""
        sd = state_dict
        assert dtype in K.H16
        self.dtype = bf = dtype

        def g(name, dtype=bf):
            return sd[name].detach().to(device=device, dtype=dtype).contiguous()

        self.embed = g("model.embed_tokens.weight")
        self.vocab, self.hidden = self.embed.shape
        self.heads, self.eps, self.theta = heads, eps, theta
        self.head_dim = self.hidden // heads
        if num_layers is None:
            num_layers = sum(1 for k in sd if k.endswith("input_layernorm.weight"))
        self.layers = []
        for i in range(num_layers):
            p = f"model.layers.{i}."
            self.layers.append(dict(
                n1=g(p + "input_layernorm.weight", torch.float32),
                wqkv=torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"),
                                g(p + "self_attn.v_proj.weight")], 0).contiguous(),
                wo=g(p + "self_attn.o_proj.weight"),
                n2=g(p + "post_attention_layernorm.weight", torch.float32),
                wgu=K.interleave_gate_up(g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight")),
                wd=g(p + "mlp.down_proj.weight")))
        self.inter = self.layers[0]['wd'].size(1) if self.layers else 0
        self.norm = g("model.norm.weight", torch.float32)
        self.lm_head = g("lm_head.weight")
        inv = 1.0 / (theta ** (torch.arange(0, self.head_dim, 2, dtype=torch.float32) / self.head_dim))
        ang = torch.arange(max_positions, dtype=torch.float32)[:, None] * inv[None, :]
        self.cos, self.sin = ang.cos().to(device).contiguous(), ang.sin().to(device).contiguous()
        self.max_positions = max_positions
        self.device = device
        self.train_weights = False
        # full sharding (gpt4roi_amd/fsdp.py): `unit_hook(phase, li)` is called around every decoder layer of the training
        # forward ("fwd_pre" / "fwd_post") and backward ("bwd_pre" / "bwd_post"); with `lazy_transposes` the W^T buffers of a
        # layer are made by that hook when the layer's weights are gathered, not for all layers up front
        self.unit_hook = None
        self.lazy_transposes = False
        self._alloc_cache(max_batch)

    def _alloc_cache(self, batch):
        L = len(self.layers)
        self.kc = torch.zeros((L, batch, self.max_positions, self.hidden), dtype=self.dtype, device=self.device)
        self.vc = torch.zeros_like(self.kc)
        self.pos = 0
        self.ragged = None             # RaggedLayout of the prompt in the cache (pad rows squeezed out), or None
        self._dstate = None            # a captured decode graph points into the old cache
        self._bstate = None

    def reset(self, batch=1):
        if self.kc.size(1) < batch:
            self._alloc_cache(batch)
        self.pos = 0
        self.ragged = None

    def _attn_ragged(self, li, q, rag, want_lse=False, compact=False):
        """Causal attention of a masked batch (see RaggedLayout): q [B, T, C] rotated, K/V of this layer already in the
        cache at the padded positions.  -> (a [B, T, C] with zero pad rows, the front-packed operands for the backward).
        One launch for the whole batch, whatever B.  compact: each sequence's kept K/V rows move to the front of its cache
        slot (rows [0, n_b)), which is what the ragged decode launch continues from."""
        B, T, C = q.shape
        H, D, Tc = self.heads, self.head_dim, rag.Tc
        qp = K.gather_rows(q.view(B * T, C), rag.idx).view(B, Tc, C)
        kp = K.gather_rows(self.kc[li].view(-1, C), rag.idx_cache).view(B, Tc, C)
        vp = K.gather_rows(self.vc[li].view(-1, C), rag.idx_cache).view(B, Tc, C)
        ap = torch.empty_like(qp)
        lse = torch.empty((B, H, Tc), dtype=torch.float32, device=q.device) if want_lse else None
        K.flash_attn(qp, kp, vp, H, 1.0 / math.sqrt(D), True, out=ap, lse=lse)
        if compact:
            self.kc[li, :B, :Tc].copy_(kp)
            self.vc[li, :B, :Tc].copy_(vp)
        a = K.gather_rows(ap.view(B * Tc, C), rag.inv).view(B, T, C)
        return a, dict(qp=qp, kp=kp, vp=vp, ap=ap, lse=lse)

    def _attn_ragged_bwd(self, pk, da, rag):
        """Backward of `_attn_ragged`: da [B, T, C] -> dq, dk, dv [B, T, C] (zero pad rows).  One launch."""
        B, T, C = da.shape
        Tc = rag.Tc
        dap = K.gather_rows(da.reshape(B * T, C), rag.idx).view(B, Tc, C)
        dqp, dkp, dvp = K.flash_attn_bwd(pk['qp'], pk['kp'], pk['vp'], pk['ap'], dap, pk['lse'], self.heads,
                                         1.0 / math.sqrt(self.head_dim), True)
        return tuple(K.gather_rows(t.view(B * Tc, C), rag.inv).view(B, T, C) for t in (dqp, dkp, dvp))

    @torch.no_grad()
    def forward(self, inputs_embeds, all_logits=True, return_hidden=False, key_padding_mask=None):
        """inputs_embeds [B,T,C] bf16, appended at the current cache position.  Returns logits fp32
        [B,T,V] (all_logits) or [B,1,V] (last position only).
        key_padding_mask (bool [B,T], True = real token; with the PROMPT only): any mask -- right / left padding, holes.
        Masked keys are attended by nobody, masked query rows leave attention as zeros, positions are those of the padded
        layout (RaggedLayout); the kept K/V rows are squeezed to the front of each cache slot, later one-token calls continue
        every sequence at its own length, and `all_logits=False` returns each sequence's LAST KEPT position."""
        B, T, C = inputs_embeds.shape
        assert self.pos + T <= self.max_positions and B <= self.kc.size(1)
        H, D, pos0 = self.heads, self.head_dim, self.pos
        rag = RaggedLayout.of(key_padding_mask, self.max_positions)
        if rag is not None:
            assert pos0 == 0 and T > 1 and (rag.B, rag.T) == (B, T), "a padding mask enters with the prompt (empty cache)"
            self.ragged = rag
            self._rag_pos = rag.pos0.clone()
        elif pos0 == 0:
            self.ragged = None
        elif self.ragged is not None:
            assert T == 1, "after a masked prompt the sequences continue one token at a time"
        x = inputs_embeds.reshape(B * T, C)
        if x.dtype != self.dtype:
            x = x.to(self.dtype)
        x = x.contiguous()
        scale = 1.0 / math.sqrt(D)
        q = torch.empty((B, T, C), dtype=self.dtype, device=x.device)
        h = None                                             # the next RMSNorm output when the down_proj reduce produced it
        for li, L in enumerate(self.layers):
            if h is None:
                h = K.rmsnorm(x, L['n1'], self.eps)
            if T == 1:                                       # decode step: RoPE + cache append + split-key attention,
                qkv = K.gemm(h, L['wqkv']).view(B, T, 3 * C)
                a = torch.empty_like(q)                      # one launch for the B sequences of the batch
                if self.ragged is not None:                  # every sequence at its own length, RoPE at the padded position
                    K.attn_decode(None, self.kc[li, :B], self.vc[li, :B], H, scale, self._attn_work(B), out=a.view(B, C),
                                  qkv=qkv.view(B, 3 * C), cos=self.cos, sin=self.sin, kv_lens_dev=self._rag_pos[:B],
                                  rope_pos_dev=self._rag_pos[B:])
                else:
                    K.attn_decode(None, self.kc[li, :B], self.vc[li, :B], H, scale, self._attn_work(B), kv_len=pos0 + 1,
                                  out=a.view(B, C), qkv=qkv.view(B, 3 * C), cos=self.cos, sin=self.sin)
            else:
                # prefill: RoPE and the cache append ride in the projection's epilogue (one launch instead of 1 + B)
                if K.gemm_qkv_rope(h, L['wqkv'], B, T, H, D, q, self.kc[li, :B], self.vc[li, :B], self.cos, self.sin,
                                   pos0) is None:
                    qkv = K.gemm(h, L['wqkv']).view(B, T, 3 * C)
                    for b in range(B):
                        K.rope_qkv(qkv[b], self.cos, self.sin, q[b], self.kc[li, b], self.vc[li, b], H, D, pos0)
                if rag is not None:
                    a, _ = self._attn_ragged(li, q, rag, compact=True)
                else:
                    a = K.flash_attn(q, self.kc[li, :B, :pos0 + T], self.vc[li, :B, :pos0 + T], H, scale, True)
            x = K.gemm(a.view(B * T, C), L['wo'], residual=x)
            h = K.rmsnorm(x, L['n2'], self.eps)
            f = K.gemm(h, L['wgu'], act="swiglu")            # gate|up GEMM with the SiLU*up epilogue
            plan = K.long_k_plan(B * T, C, f.size(1))
            if plan is not None:
                # prefill: the down_proj runs as K slices; their reduce (+ residual) and the NEXT RMSNorm (the next layer's
                # input_layernorm, or the final norm) are one pass over the row instead of two launches
                part, ns = K.gemm_partials(f, L['wd'], plan[1], plan[0])
                g_next = self.layers[li + 1]['n1'] if li + 1 < len(self.layers) else self.norm
                x, h = K.rmsnorm_splitk(part, ns, x, g_next, self.eps)
            else:
                x = K.gemm(f, L['wd'], residual=x)
                h = None
        self.pos = pos0 + T
        if T == 1 and self.ragged is not None:
            self._rag_pos += 1
        xn = (h if h is not None else K.rmsnorm(x, self.norm, self.eps)).view(B, T, C)
        if return_hidden:
            return xn
        if not all_logits:
            xn = K.gather_rows(xn.view(B * T, C), rag.last).view(B, 1, C) if rag is not None else xn[:, -1:, :].contiguous()
        logits = K.gemm(xn.reshape(-1, C), self.lm_head, out_dtype=torch.float32)
        return logits.view(B, -1, self.vocab)

    # ---- training rows: forward that keeps what the backward needs ---------
    def prepare_training(self, train_weights=False):
        """Materialise W^T for every projection (the input-gradient GEMMs are NT GEMMs against W^T;
        13 GB for the 7B model, a one-off for frozen weights, refreshed by `refresh_transposes` after an
        optimizer step when the decoder itself is trained)."""
        assert self.dtype is torch.bfloat16, "training runs in bf16 (train_stage1.sh:19); the fp16 instantiation is inference only"
        self.train_weights = train_weights
        self.v_pad = -(-self.vocab // 64) * 64
        self.refresh_transposes()

    def refresh_transposes(self):
        if not self.lazy_transposes:
            for L in self.layers:
                for nm in ("wqkv", "wo", "wgu", "wd"):
                    L[nm + "_t"] = K.transpose(L[nm])
        if self.lm_head is not None:
            self.lm_head_t = K.transpose(self.lm_head, self.v_pad)

    def layer_transposes(self, li, scratch=None):
        """W^T of ONE layer (the input-gradient GEMMs of its backward), into `scratch` (name -> buffer) when given."""
        L = self.layers[li]
        for nm in ("wqkv", "wo", "wgu", "wd"):
            L[nm + "_t"] = K.transpose(L[nm], out=None if scratch is None else scratch.get(nm))

    def set_tensor(self, name, tensor):
        """Point the kernels at `tensor` for `name` (a key of trainable_tensors()); None = released (a sharded unit whose
        parameters are not gathered: touching it then fails loudly instead of reading stale weights)."""
        if name == "embed_tokens":
            self.embed = tensor
        elif name == "norm":
            self.norm = tensor
        elif name == "lm_head":
            self.lm_head = tensor
        else:
            li, nm = name.split(".")
            self.layers[int(li)][nm] = tensor

    def trainable_tensors(self):
        """name -> tensor the kernels read, for stage-2 training (kernel layouts: fused q|k|v rows, interleaved
        gate/up rows; norm weights are fp32 already)."""
        # registration order = REVERSE of the order `backward` produces the gradients (lm_head, norm, last layer ...
        # first layer, embedding rows), because the bucketed exchange fills its buckets from the end of this list
        out = {"embed_tokens": self.embed}
        for i, L in enumerate(self.layers):
            for nm in ("wqkv", "wo", "wgu", "wd", "n1", "n2"):
                out[f"{i}.{nm}"] = L[nm]
        out["norm"] = self.norm
        out["lm_head"] = self.lm_head
        return out

    def rebind_tensor(self, name, tensor):
        """Replace the tensor the kernels read for `name` (a key of trainable_tensors()) by `tensor` -- same shape, dtype
        and values, different storage: the sharded optimizer moves every weight into its flat all-gather buckets."""
        cur = self.trainable_tensors()[name]
        assert tensor.shape == cur.shape and tensor.dtype == cur.dtype and tensor.is_contiguous()
        if name == "embed_tokens":
            self.embed = tensor
        elif name == "norm":
            self.norm = tensor
        elif name == "lm_head":
            self.lm_head = tensor
        else:
            li, nm = name.split(".")
            self.layers[int(li)][nm] = tensor

    def export_hf_state_dict(self):
        """The HF-named state dict of the current weights (undoes the q|k|v fusion and the gate/up interleave)."""
        C = self.hidden
        sd = {"model.embed_tokens.weight": self.embed, "model.norm.weight": self.norm, "lm_head.weight": self.lm_head}
        for i, L in enumerate(self.layers):
            p = f"model.layers.{i}."
            sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"] = \
                L['wqkv'][:C], L['wqkv'][C:2 * C], L['wqkv'][2 * C:]
            sd[p + "self_attn.o_proj.weight"] = L['wo']
            sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = L['wgu'][0::2], L['wgu'][1::2]
            sd[p + "mlp.down_proj.weight"] = L['wd']
            sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = L['n1'], L['n2']
        return {k: v.detach().clone() for k, v in sd.items()}

    def _layer_forward_train(self, li, x, B, T, rag=None):
        """One decoder layer of `forward_train`: x [B*T, C] -> (x_out, everything its backward reads).
        rag: RaggedLayout of a masked batch (the reference's unpad -> varlen attention -> pad_input) or None."""
        if self.unit_hook is not None:
            self.unit_hook("fwd_pre", li)
        L = self.layers[li]
        C, H, D = self.hidden, self.heads, self.head_dim
        h = K.rmsnorm(x, L['n1'], self.eps)
        q = torch.empty((B, T, C), dtype=torch.bfloat16, device=x.device)
        # RoPE and the K/V writes ride in the projection's epilogue (as in `forward`: one launch instead of 1 + B, same bits)
        if K.gemm_qkv_rope(h, L['wqkv'], B, T, H, D, q, self.kc[li, :B], self.vc[li, :B], self.cos, self.sin, 0) is None:
            qkv = K.gemm(h, L['wqkv']).view(B, T, 3 * C)
            for b in range(B):
                K.rope_qkv(qkv[b], self.cos, self.sin, q[b], self.kc[li, b], self.vc[li, b], H, D, 0)
        pk = lse = None
        if rag is not None:
            a, pk = self._attn_ragged(li, q, rag, want_lse=True)
        else:
            lse = torch.empty((B, H, T), dtype=torch.float32, device=x.device)
            a = K.flash_attn(q, self.kc[li, :B, :T], self.vc[li, :B, :T], H, 1.0 / math.sqrt(D), True, lse=lse)
        x1 = K.gemm(a.view(B * T, C), L['wo'], residual=x)
        h2 = K.rmsnorm(x1, L['n2'], self.eps)
        gu = K.gemm(h2, L['wgu'])
        f = K.swiglu_il(gu)
        x2 = K.gemm(f, L['wd'], residual=x1)
        rec = dict(x=x, q=q, a=a, lse=lse, x1=x1, gu=gu, pk=pk)
        if self.train_weights:
            rec.update(h=h, h2=h2, f=f)
        if self.unit_hook is not None:
            self.unit_hook("fwd_post", li)
        return x2, rec

    def forward_train(self, inputs_embeds, checkpoint=False, key_padding_mask=None):
        """inputs_embeds [B,T,C] bf16 at positions 0..T-1 -> (logits fp32 [B*T, V], ctx).  key_padding_mask (bool [B,T]):
        the `attention_mask` the reference's flash-attention patch unpads with (llama_flash_attn_monkey_patch.py:60-85), any
        pattern; a right-padded batch may also come without it (causal attention hides the pad keys from the real rows and
        the pad rows carry no loss: same logits on the real rows, same gradients).  Same kernels as
        `forward` except that the gate|up GEMM keeps its pre-activation output for the SwiGLU backward and
        the attention kernel also returns the log-sum-exp.
        checkpoint=True (`--gradient_checkpointing True`, train_stage1.sh:36; torch.utils.checkpoint per decoder layer in
        HF): only each layer's INPUT is kept (2 B x C per token per layer instead of ~(6 C + 3 F) x 2 B) and `backward`
        re-runs the layer forward before differentiating it -- one extra forward of the decoder per step, which is what
        lets config 4's B = 16 x T = 767 tokens per GPU train inside 288 GB next to the 7 B replica and its Adam state."""
        B, T, C = inputs_embeds.shape
        assert T <= self.max_positions and B <= self.kc.size(1)
        x = inputs_embeds.reshape(B * T, C).contiguous()
        rag = RaggedLayout.of(key_padding_mask, self.max_positions)
        assert rag is None or (rag.B, rag.T) == (B, T)
        saved = []
        for li in range(len(self.layers)):
            x2, rec = self._layer_forward_train(li, x, B, T, rag)
            saved.append(dict(x=x) if checkpoint else rec)
            x = x2
        xn = K.rmsnorm(x, self.norm, self.eps)
        logits = K.gemm(xn, self.lm_head, out_dtype=torch.float32)
        self.pos = T
        return logits, dict(B=B, T=T, saved=saved, x_final=x, xn=xn, checkpoint=checkpoint, rag=rag)

    def backward(self, ctx, dlogits, on_grad=None, grad_slot=None):
        """dlogits bf16 [B*T, v_pad] (zero in the pad columns) -> d(inputs_embeds) [B*T, C] bf16.
        With `train_weights`, self.grads[name] receives the fp32 weight gradients (layer weights in the
        kernel layout: fused qkv, interleaved gate|up); `on_grad(name, grad)` is called as soon as a layer's gradients
        exist (last layer first), so a bucketed exchange can start while the earlier layers are still differentiating;
        `grad_slot(name)` (optional) returns the dense fp32 tensor the consumer wants that gradient written INTO (a view of
        its flat exchange bucket) or None: the weight-gradient kernels then write there and no copy follows."""
        B, T = ctx["B"], ctx["T"]
        C, H, D = self.hidden, self.heads, self.head_dim
        scale = 1.0 / math.sqrt(D)
        tw = self.train_weights
        grads = {}

        def wgrad(name, dy, x):
            out = grad_slot(name) if grad_slot is not None else None
            grads[name] = K.linear_wgrad(dy, x, out=out.view(dy.size(1), x.size(1)) if out is not None else None)

        def emit(*names):
            if on_grad is not None:
                for n in names:
                    if on_grad(n, grads[n]):                # truthy = the consumer copied it (sharded buckets): free it now
                        del grads[n]
        if tw:
            grads["lm_head"] = K.linear_wgrad(dlogits[:, :self.vocab], ctx["xn"])
            grads["norm"] = torch.zeros_like(self.norm)
        dxn = K.gemm(dlogits, self.lm_head_t)
        dx = K.rmsnorm_bwd(ctx["x_final"], self.norm, dxn, dgamma=grads.get("norm"), eps=self.eps)
        if tw:
            emit("lm_head", "norm")
        for li in range(len(self.layers) - 1, -1, -1):
            if self.unit_hook is not None:
                self.unit_hook("bwd_pre", li)
            L, S = self.layers[li], ctx["saved"][li]
            if ctx.get("checkpoint"):
                hook, self.unit_hook = self.unit_hook, None               # (the recompute runs inside this layer's gather)
                _, S = self._layer_forward_train(li, S['x'], B, T, ctx.get("rag"))   # recompute (also re-fills this layer's K/V)
                self.unit_hook = hook
            df = K.gemm(dx, L['wd_t'])
            dgu = K.swiglu_il_bwd(S['gu'], df)
            dh2 = K.gemm(dgu, L['wgu_t'])
            if tw:
                wgrad(f"{li}.wd", dx, S['f'])
                wgrad(f"{li}.wgu", dgu, S['h2'])
                grads[f"{li}.n2"] = torch.zeros_like(L['n2'])
                grads[f"{li}.n1"] = torch.zeros_like(L['n1'])
            dx1 = K.rmsnorm_bwd(S['x1'], L['n2'], dh2, dres=dx, dgamma=grads.get(f"{li}.n2"), eps=self.eps)
            da = K.gemm(dx1, L['wo_t']).view(B, T, C)
            if ctx.get("rag") is not None:
                dq, dk, dv = self._attn_ragged_bwd(S['pk'], da, ctx["rag"])
            else:
                dq, dk, dv = K.flash_attn_bwd(S['q'], self.kc[li, :B, :T], self.vc[li, :B, :T], S['a'], da, S['lse'], H,
                                              scale, True)
            # the whole batch in one launch, every sequence's rows written in place (positions restart every T rows)
            dqkv = K.rope_qkv_bwd(dq.view(B * T, C), dk.view(B * T, C), dv.view(B * T, C), self.cos, self.sin, H, D, 0,
                                  period=T)
            dh = K.gemm(dqkv, L['wqkv_t'])
            if tw:
                wgrad(f"{li}.wo", dx1, S['a'].view(B * T, C))
                wgrad(f"{li}.wqkv", dqkv, S['h'])
            dx = K.rmsnorm_bwd(S['x'], L['n1'], dh, dres=dx1, dgamma=grads.get(f"{li}.n1"), eps=self.eps)
            if tw:
                emit(f"{li}.n2", f"{li}.n1", f"{li}.wd", f"{li}.wgu", f"{li}.wo", f"{li}.wqkv")
            if self.unit_hook is not None:
                self.unit_hook("bwd_post", li)
        self.grads = grads
        return dx

    def loss_and_dlogits(self, logits, labels):
        """Shifted-label token cross entropy (llava/model/llava.py:240-252): logits fp32 [B*T, V], labels int64
        [B, T] with -100 = ignored.  Returns (mean loss fp32 [1] on the device, dlogits bf16 [B*T, v_pad])."""
        B, T = labels.shape
        lab = torch.full((B, T), -100, dtype=torch.int64, device=logits.device)
        lab[:, :-1] = labels[:, 1:]
        lab = lab.reshape(-1).contiguous()
        cnt = ((lab >= 0) & (lab < self.vocab)).sum().clamp(min=1)     # the kernel ignores labels outside [0, V)
        gs = (1.0 / cnt.float()).reshape(1).contiguous()
        loss_sum = torch.zeros(1, dtype=torch.float32, device=logits.device)
        dlogits = torch.empty((B * T, self.v_pad), dtype=torch.bfloat16, device=logits.device)
        K.cross_entropy(logits, lab, loss_sum, gs, dlogits, self.v_pad)
        return loss_sum * gs, dlogits

    # ---- device-resident decode (greedy or sampled), one hipGraph replay per token ---------------
    def _decode_state(self, max_new):
        st = getattr(self, "_dstate", None)
        if st is None or st["out"].numel() < max_new:
            dev = self.device
            n = max(max_new, 64)
            # the state outlives the call: keep it an ordinary tensor even when the caller runs under inference_mode
            # (app.py:285), or a later in-place update outside that mode would be refused
            with torch.inference_mode(False):
                st = self._new_decode_state(dev, n)
            self._dstate = st
        return st

    @staticmethod
    def _new_decode_state(dev, n):
        return dict(tok=torch.zeros((1, 1), dtype=torch.int64, device=dev),
                    pos=torch.zeros(1, dtype=torch.int32, device=dev),
                    step=torch.zeros(1, dtype=torch.int32, device=dev),
                    seed=torch.zeros(1, dtype=torch.int64, device=dev),
                    u=torch.zeros(n, dtype=torch.float32, device=dev),
                    out=torch.zeros(n, dtype=torch.int64, device=dev), graphs={})

    def _attn_work(self, batch=1):
        """Decode-attention workspace for `batch` sequences; one per batch size, never freed (captured decode graphs hold
        its addresses)."""
        pool = getattr(self, "_attn_ws", None)
        if pool is None:
            pool = self._attn_ws = {}
        if batch not in pool:
            # keys of a (sequence, head) split over `splits` workgroups: 8 for one sequence (32 heads -> 256 workgroups), fewer
            # as the batch itself fills the chip (fewer partials to merge, longer runs of keys per workgroup)
            splits = int(os.environ.get("G4R_ATTN_SPLITS", 0)) or max(1, min(8, 512 // (self.heads * batch)))
            with torch.inference_mode(False):
                pool[batch] = K.DecodeAttnWorkspace(self.heads, self.head_dim, self.device, splits=splits, batch=batch)
        return pool[batch]

    def _advance(self, logits_row, st, sampler):
        """Token selection on the device: argmax (generate(do_sample=False)) or one temperature / top-k / top-p draw
        (do_sample=True, app.py:293-300); sampler = (temperature, top_k, top_p)."""
        if sampler is None:
            K.greedy_advance(logits_row, st["tok"], st["out"], st["step"], st["pos"])
        else:
            K.sample_advance(logits_row, st["tok"], st["out"], st["step"], st["pos"], st["seed"], sampler[0], sampler[1],
                             sampler[2], u_out=st["u"])

    def _decode_step_device(self, st, sampler=None):
        """One token: embedding of st['tok'] at position st['pos'] -> 32 layers (K/V appended at *pos,
        attention over *pos + 1 keys) -> logits -> token selection -> st['tok'], st['out'][step]; counters
        advance on the device.  No host value enters the launch sequence."""
        C, H, D = self.hidden, self.heads, self.head_dim
        x, _ = K.splice_embed(st["tok"], self.embed, None, None, None, 0, -1, -1, -1, -1)
        x = x.view(1, C)
        scale = 1.0 / math.sqrt(D)
        work = self._attn_work()
        # 5 launches per layer: the two RMSNorms ride in the prologue of the GEMVs they feed; RoPE and the cache append
        # ride in the attention launch, which splits the cached keys of a head over 8 workgroups (the tiled prefill
        # kernel would run this case with one workgroup per head); their partials are merged by the o_proj's staging
        for li, L in enumerate(self.layers):
            qkv = K.gemv(x, L['wqkv'], norm_weight=L['n1'], eps=self.eps)
            K.attn_decode(None, self.kc[li, 0], self.vc[li, 0], H, scale, work, kv_len_dev=st["pos"], qkv=qkv,
                          cos=self.cos, sin=self.sin, defer_merge=True)
            x = K.gemv_attn_merge(work, H, D, L['wo'], residual=x)
            f = K.gemv(x, L['wgu'], norm_weight=L['n2'], eps=self.eps, act="swiglu")
            x = K.gemv(f, L['wd'], residual=x)
        logits = K.gemv(x, self.lm_head, norm_weight=self.norm, eps=self.eps, out_dtype=torch.float32)
        self._advance(logits.view(-1), st, sampler)

    @torch.no_grad()
    def decode_graph(self, inputs_embeds, max_new_tokens, stop_ids=(), check_every=32, use_graph=True, sampler=None,
                     seed=0, on_tokens=None):
        """generate() for batch 1 with the per-token loop on the device: prefill eagerly, then replay one captured
        hipGraph per token (token id, position, output slot, sampling seed and step live in device memory).
        sampler = None (greedy) or (temperature, top_k, top_p); the uniform of step s is Philox(seed, s).
        The host looks at the ids every `check_every` tokens: `stop_ids` ends the sequence at the first hit
        (inclusive); `on_tokens(list of new ids) -> bool` is the hook for HF-style stopping criteria."""
        assert inputs_embeds.size(0) == 1
        self.reset(1)
        logits = self.forward(inputs_embeds, all_logits=False)
        T = self.pos
        assert T + max_new_tokens <= self.max_positions
        st = self._decode_state(max_new_tokens)
        st["pos"].fill_(T - 1)
        st["step"].zero_()
        st["seed"].fill_(int(seed))
        self._advance(logits.view(-1), st, sampler)                                       # token 1, pos -> T
        done = 1
        key = None if sampler is None else tuple(float(x) for x in sampler)
        graph = st["graphs"].get(key) if use_graph else None

        checked = [0]                               # tokens already shown to stop_ids / the criteria

        def finished(n):
            ids = st["out"][:n].tolist()
            for m in range(checked[0] + 1, n + 1):  # each prefix once, token by token, like HF's loop: n calls in all
                if ids[m - 1] in stop_ids or (on_tokens is not None and on_tokens(ids[:m])):
                    return m
            checked[0] = n
            return None

        end = finished(done) if (stop_ids or on_tokens) else None
        if end is None and max_new_tokens > 1:
            self._decode_step_device(st, sampler)                                           # token 2 (also warms up)
            done = 2
            if use_graph and graph is None and max_new_tokens > 2:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()      # capture records the launches, it does not run them
                side = torch.cuda.Stream()
                # capture outside inference_mode even if the caller is inside it (app.py:285): torch's capture registers
                # generator state tensors, which must stay ordinary tensors for later captures in the process
                with torch.inference_mode(False), torch.no_grad():
                    with torch.cuda.graph(g, stream=side):
                        self._decode_step_device(st, sampler)
                st["graphs"][key] = graph = g
        while end is None and done < max_new_tokens:
            n = min(check_every, max_new_tokens - done)
            for _ in range(n):
                if graph is not None:
                    graph.replay()
                else:
                    self._decode_step_device(st, sampler)
            done += n
            if stop_ids or on_tokens:
                end = finished(done)
        if end is not None:
            done = end
        self.pos = T + done
        return st["out"][:done].tolist()

    def greedy_graph(self, inputs_embeds, max_new_tokens, stop_ids=(), check_every=32, use_graph=True):
        """generate(do_sample=False): decode_graph without a sampler."""
        return self.decode_graph(inputs_embeds, max_new_tokens, stop_ids, check_every, use_graph)

    def _decode_step_batch(self, st):
        """One token for each of B equal-length sequences: the weight stream of the step is shared by the batch (small-M
        GEMM tiles, profiles/r02_gemm_small_m.txt), RoPE + cache append + attention of all sequences in one launch."""
        B, C, H, D = st["tok32"].numel(), self.hidden, self.heads, self.head_dim
        scale = 1.0 / math.sqrt(D)
        work = self._attn_work(B)
        x = K.gather_rows(self.embed, st["tok32"])
        fused = K.GEMV_BATCH_FUSED_NORM and K.gemv_batch_wins(B, 3 * C, C) and x.stride(0) % 8 == 0

        def norm_proj(x, gamma, w, h=None, **kw):
            """RMSNorm + projection: the normalised rows h when the previous launch already made them, else one launch on the
            weight-streaming kernel (few rows), else two"""
            if h is not None:
                return K.gemm(h, w, **kw)
            if fused and K.gemv_batch_wins(B, w.size(0), C):
                return K.gemv_batch(x, w, norm_weight=gamma, eps=self.eps, variant=K.GEMV_BATCH_VARIANT, **kw)
            return K.gemm(K.rmsnorm(x, gamma, self.eps), w, **kw)

        def proj_res_norm(a, w, x, gamma_next):
            """x + a . w^T and, where that projection runs as K slices (more rows than the weight-streaming kernel takes), the NEXT
            RMSNorm in the slices' reduce launch (round 6: -64 launches per step; same bits as reduce, then norm)"""
            plan = K.decode_split_plan(B, w.size(0), a.size(1)) if K.DECODE_SPLITK_NORM else None
            if plan is None:
                return K.gemm(a, w, residual=x), None
            part, ns = K.gemm_partials(a, w, plan[1], plan[0])
            return K.rmsnorm_splitk(part, ns, x, gamma_next, self.eps)
        h = None
        nl = len(self.layers)
        for li, L in enumerate(self.layers):
            qkv = norm_proj(x, L['n1'], L['wqkv'], h=h)
            if st.get("ragged"):            # pos = [B cache lengths | RoPE position], all advanced by batch_advance
                a = K.attn_decode(None, self.kc[li, :B], self.vc[li, :B], H, scale, work, qkv=qkv, cos=self.cos,
                                  sin=self.sin, kv_lens_dev=st["pos"][:B], rope_pos_dev=st["pos"][B:])
            else:
                a = K.attn_decode(None, self.kc[li, :B], self.vc[li, :B], H, scale, work, kv_len_dev=st["pos"], qkv=qkv,
                                  cos=self.cos, sin=self.sin)
            x, h2 = proj_res_norm(a, L['wo'], x, L['n2'])
            f = norm_proj(x, L['n2'], L['wgu'], h=h2, act="swiglu")
            x, h = proj_res_norm(f, L['wd'], x, self.layers[li + 1]['n1'] if li + 1 < nl else self.norm)
        logits = norm_proj(x, self.norm, self.lm_head, h=h, out_dtype=torch.float32)
        K.batch_advance(K.argmax_rows(logits), st["tok"], st["tok32"], st["out"], st["step"], st["pos"])

    @torch.no_grad()
    def decode_graph_batch(self, inputs_embeds, max_new_tokens, stop_ids=(), use_graph=True, key_padding_mask=None):
        """generate(do_sample=False) for B sequences, per-token loop on the device.  Prompts of equal length, or -- with
        key_padding_mask (bool [B, T]) -- prompts padded to a common T on either side (HF batch generation pads on the
        left; merged serving requests of different lengths pad on the right): see `forward`.  Prefill
        eagerly, then one captured hipGraph replay per step for the whole batch (token ids, position and output slots stay
        in device memory).  The 13.2 GB weight stream of a step is shared by the B sequences -- the throughput mode of
        SURVEY.md 8d config 5.  Sequences that hit a stop id keep running; their later tokens are cut from the result.
        Returns a list of B id lists (same ids as greedy_batch, the host-loop form)."""
        B = inputs_embeds.size(0)
        self.reset(B)
        logits = self.forward(inputs_embeds, all_logits=False, key_padding_mask=key_padding_mask)
        T = self.pos
        assert T + max_new_tokens <= self.max_positions
        dev = inputs_embeds.device
        ragged = self.ragged is not None
        pool = getattr(self, "_bstate", None)
        if pool is None:
            pool = self._bstate = {}
        bs = pool.get((B, ragged))
        if bs is None or bs["out"].size(1) < max_new_tokens:
            cap = max(64, 1 << (max_new_tokens - 1).bit_length())      # output slots: the captured graph knows this stride
            with torch.inference_mode(False):
                bs = pool[(B, ragged)] = dict(B=B, ragged=ragged, tok=torch.zeros(B, dtype=torch.int64, device=dev),
                                              tok32=torch.zeros(B, dtype=torch.int32, device=dev),
                                              out=torch.zeros((B, cap), dtype=torch.int64, device=dev),
                                              pos=torch.zeros(B + 1 if ragged else 1, dtype=torch.int32, device=dev),
                                              step=torch.zeros(1, dtype=torch.int32, device=dev), graph=None)
        st = bs
        if ragged:
            st["pos"].copy_(self._rag_pos - 1)                          # [n_b - 1 ..., T - 1]: the advance below moves them on
        else:
            st["pos"].fill_(T - 1)
        st["step"].zero_()
        K.batch_advance(K.argmax_rows(logits.view(B, -1)), st["tok"], st["tok32"], st["out"], st["step"], st["pos"])
        done = 1
        if max_new_tokens > 1:
            self._decode_step_batch(st)                                  # token 2 (also warms up)
            done = 2
        if use_graph and st["graph"] is None and max_new_tokens > 2:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.inference_mode(False), torch.no_grad():
                with torch.cuda.graph(g):                                # (recorded, not executed: the state does not move)
                    self._decode_step_batch(st)
            st["graph"] = g
        while done < max_new_tokens:
            if st["graph"] is not None:
                st["graph"].replay()
            else:
                self._decode_step_batch(st)
            done += 1
        self.pos = T + max_new_tokens - 1
        res = []
        for row in st["out"][:, :max_new_tokens].tolist():
            hit = [i for i, t in enumerate(row) if t in stop_ids]
            res.append(row[:hit[0] + 1] if hit else row)
        return res

    @torch.no_grad()
    def greedy_batch(self, inputs_embeds, max_new_tokens, stop_ids=(), key_padding_mask=None):
        """generate(do_sample=False) for B sequences with prompts of EQUAL length, or padded to one length under
        key_padding_mask (see `forward`; the reference stacks its samples,
        spi_llava.py:196, so a batch always has one T): one decoder pass per step for the whole batch, i.e. the
        13.5 GB weight stream of a decode step is shared by B sequences.  Sequences that hit a
        stop id keep running (their later tokens are cut from the result).  Returns a list of B id lists."""
        B = inputs_embeds.size(0)
        self.reset(B)
        logits = self.forward(inputs_embeds, all_logits=False, key_padding_mask=key_padding_mask)
        assert self.pos + max_new_tokens <= self.max_positions
        out = torch.empty((B, max_new_tokens), dtype=torch.int64, device=inputs_embeds.device)
        for s in range(max_new_tokens):
            nxt = K.argmax_rows(logits.view(B, -1))
            out[:, s] = nxt
            if s + 1 == max_new_tokens:
                break
            emb = K.gather_rows(self.embed, nxt.to(torch.int32))
            logits = self.forward(emb.view(B, 1, -1), all_logits=False)
        res = []
        for row in out.tolist():
            hit = [i for i, t in enumerate(row) if t in stop_ids]
            res.append(row[:hit[0] + 1] if hit else row)
        return res

    @torch.no_grad()
    def greedy(self, inputs_embeds, max_new_tokens, stop_ids=()):
        """generate(do_sample=False) for batch 1: prefill, then one token per step from the KV cache
        (llava/model/llava.py:263-283 feeds only the last token after step 0)."""
        assert inputs_embeds.size(0) == 1
        self.reset(1)
        logits = self.forward(inputs_embeds, all_logits=False)
        out = []
        for _ in range(max_new_tokens):
            nxt = K.argmax_rows(logits.view(1, -1))
            tok = int(nxt.item())
            out.append(tok)
            if tok in stop_ids:
                break
            logits = self.forward(self.embed[nxt].view(1, 1, -1), all_logits=False)
        return out
