"""oracle/transformer_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Plain-torch CPU restatement of the two third-party transformers the reference calls on its hot
path.  Their arithmetic is NOT under /root/reference: it lives in HuggingFace `transformers`,
pinned by the reference at git cae78c46 (pyproject.toml:19).  Call sites:
  CLIPVisionModel(images, output_hidden_states=True)   gpt4roi/models/spi_llava.py:66-67
       (built llava/model/llava.py:48,61-66); hidden-state selection spi_llava.py:58-82
  LlamaModel.forward(inputs_embeds=...) + lm_head      spi_llava.py:198-205; llava.py:235-238
Published algorithms restated here: CLIP ViT (pre-LN, biased q/k/v/out, QuickGELU MLP, class token
+ learned positions, pre_layrnorm) and LLaMA (RMSNorm, rotary embedding in the rotate_half
convention, SwiGLU, no biases).

Pinning: the reference has no test at this boundary ("parity unpinned" in SURVEY.md 8c).  What
we can and do pin: tests/test_oracle_transformers.py checks these functions against the container's
transformers (5.x) CLIPVisionModel / LlamaForCausalLM with identical random weights -- the same
module structure as the pinned commit for these two models.

Weights are plain dicts keyed by the HF parameter names.  `emulate=True` rounds to bf16 where the
MI355X pipeline stores bf16 (after every GEMM epilogue / norm / attention output).
"""
import math

import torch
import torch.nn.functional as F


def _r(x, emulate):
    """emulate: False = exact fp32; True = round to bfloat16 (the training dtype); a torch dtype (torch.float16: the
    reference's serving dtype, app.py:74-98) = round to that type."""
    if not emulate:
        return x
    return x.to(torch.bfloat16 if emulate is True else emulate).to(torch.float32)


def _lin(x, w, b, emulate):
    y = _r(x, emulate) @ _r(w, emulate).t()
    if b is not None:
        y = y + _r(b, emulate)
    return y


def attention(q, k, v, heads, scale, causal, emulate=False, key_padding_mask=None):
    """q [B,Tq,H*D], k/v [B,Tk,H*D] -> [B,Tq,H*D]; causal: query i sees keys <= i + (Tk - Tq).
    key_padding_mask (bool [B,Tk], True = a real token): the arithmetic of the reference's unpad -> varlen flash attention
    -> pad_input (llava/train/llama_flash_attn_monkey_patch.py:60-85): a masked key is attended by nobody, the order of the
    kept tokens is the causal order, and a masked QUERY row comes back as zeros (pad_input scatters into zeros).  The
    valid rows equal HF LlamaModel's with the same attention_mask (its additive -inf mask; llava.py:263-283 for decode)."""
    B, Tq, HD = q.shape
    Tk, D = k.shape[1], HD // heads
    qh = q.view(B, Tq, heads, D).transpose(1, 2)
    kh = k.view(B, Tk, heads, D).transpose(1, 2)
    vh = v.view(B, Tk, heads, D).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) * scale
    if causal:
        i = torch.arange(Tq, device=q.device)[:, None] + (Tk - Tq)
        s = s.masked_fill(torch.arange(Tk, device=q.device)[None, :] > i, float("-inf"))
    if key_padding_mask is not None:
        km = key_padding_mask.to(device=q.device, dtype=torch.bool)
        s = s.masked_fill(~km[:, None, None, :], float("-inf"))
    m = s.max(-1, keepdim=True).values
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)          # a row with no visible key
    e = torch.exp(s - m)
    o = (_r(e, emulate) @ vh) / e.sum(-1, keepdim=True).clamp_min(1e-30)
    o = o.transpose(1, 2).reshape(B, Tq, HD)
    if key_padding_mask is not None:
        o = o * km[:, Tk - Tq:, None].to(o.dtype)                    # queries are the last Tq positions of the mask
    return o


# ------------------------------------------------------------------------------------------ CLIP ViT
def clip_vit_hidden_states(w, images, heads=16, n_layers=None, eps=1e-5, emulate=False, prefix=None):
    """Returns [hs_0 (after pre_layrnorm), hs_1, ..., hs_n] like HF output_hidden_states.
    Keys are HF's; older releases prefix them with "vision_model." (auto-detected)."""
    if prefix is None:
        prefix = "vision_model." if "vision_model.pre_layrnorm.weight" in w else ""
    pw = w[prefix + "embeddings.patch_embedding.weight"]
    C = pw.shape[0]
    B = images.shape[0]
    patches = F.conv2d(_r(images.float(), emulate), _r(pw, emulate), stride=pw.shape[-1]).flatten(2).transpose(1, 2)
    patches = _r(patches, emulate)
    cls = w[prefix + "embeddings.class_embedding"].expand(B, 1, C)
    x = torch.cat([_r(cls, emulate), patches], 1) + _r(w[prefix + "embeddings.position_embedding.weight"], emulate)
    x = _r(x, emulate)
    # the attribute really is spelled "pre_layrnorm" in HF
    x = _r(F.layer_norm(x, (C,), w[prefix + "pre_layrnorm.weight"], w[prefix + "pre_layrnorm.bias"], eps), emulate)
    hs = [x]
    L = n_layers if n_layers is not None else sum(
        1 for k in w if k.startswith(prefix + "encoder.layers.") and k.endswith("layer_norm1.weight"))
    D = C // heads
    for i in range(L):
        p = f"{prefix}encoder.layers.{i}."
        h = _r(F.layer_norm(x, (C,), w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], eps), emulate)
        q = _r(_lin(h, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"], emulate), emulate)
        k = _r(_lin(h, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"], emulate), emulate)
        v = _r(_lin(h, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"], emulate), emulate)
        a = _r(attention(q, k, v, heads, D ** -0.5, False, emulate), emulate)
        x = _r(_lin(a, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"], emulate) + x, emulate)
        h = _r(F.layer_norm(x, (C,), w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], eps), emulate)
        f = _lin(h, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"], emulate)
        f = _r(f * torch.sigmoid(1.702 * f), emulate)
        x = _r(_lin(f, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"], emulate) + x, emulate)
        hs.append(x)
    return hs


def select_spi_levels(hidden_states, select_layer=-2, num_levels=4):
    """spi_llava.py:58-82: image_features = hs[select][:,1:]; levels = hs[select::-3][::-1][-4:]."""
    image_features = hidden_states[select_layer][:, 1:]
    mlvl = hidden_states[select_layer::-3][::-1][-num_levels:]
    return image_features, [m[:, 1:] for m in mlvl]


# ------------------------------------------------------------------------------------------ LLaMA
def rope_tables(max_pos, head_dim, theta=10000.0):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    ang = torch.arange(max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return ang.cos(), ang.sin()          # [max_pos, head_dim/2]


def apply_rope(x, cos, sin, heads, pos0):
    B, T, HD = x.shape
    D = HD // heads
    x = x.view(B, T, heads, D)
    c = torch.cat([cos, cos], -1)[pos0:pos0 + T][None, :, None, :]
    s = torch.cat([sin, sin], -1)[pos0:pos0 + T][None, :, None, :]
    rot = torch.cat([-x[..., D // 2:], x[..., :D // 2]], -1)
    return (x * c + rot * s).reshape(B, T, HD)


def rmsnorm(x, g, eps, emulate=False):
    y = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return _r(y, emulate) * g


def llama_forward(w, inputs_embeds, heads, eps=1e-6, theta=10000.0, kv_cache=None, pos0=0, emulate=False,
                  n_layers=None, prefix="model.", key_padding_mask=None):
    """inputs_embeds [B,T,C] -> (final hidden [B,T,C], kv_cache).  kv_cache: list of (k, v) per
    layer with the previously cached positions [B, pos0, C].
    key_padding_mask: bool [B, pos0 + T] over cached + new positions (see `attention`); positions stay those of the padded
    layout (the pinned LlamaModel numbers them arange(past, past + T) whatever the mask)."""
    B, T, C = inputs_embeds.shape
    D = C // heads
    cos, sin = (t.to(inputs_embeds.device) for t in rope_tables(pos0 + T, D, theta))   # tests may run this on the GPU
    L = n_layers if n_layers is not None else sum(1 for k in w if k.endswith("input_layernorm.weight"))
    x = _r(inputs_embeds.float(), emulate)
    new_cache = []
    for i in range(L):
        p = f"{prefix}layers.{i}."
        h = _r(rmsnorm(x, w[p + "input_layernorm.weight"], eps, emulate), emulate)
        q = _r(_lin(h, w[p + "self_attn.q_proj.weight"], None, emulate), emulate)
        k = _r(_lin(h, w[p + "self_attn.k_proj.weight"], None, emulate), emulate)
        v = _r(_lin(h, w[p + "self_attn.v_proj.weight"], None, emulate), emulate)
        q = _r(apply_rope(q, cos, sin, heads, pos0), emulate)
        k = _r(apply_rope(k, cos, sin, heads, pos0), emulate)
        if kv_cache is not None:
            k = torch.cat([kv_cache[i][0], k], 1)
            v = torch.cat([kv_cache[i][1], v], 1)
        new_cache.append((k, v))
        a = _r(attention(q, k, v, heads, 1.0 / math.sqrt(D), True, emulate, key_padding_mask), emulate)
        x = _r(_lin(a, w[p + "self_attn.o_proj.weight"], None, emulate) + x, emulate)
        h = _r(rmsnorm(x, w[p + "post_attention_layernorm.weight"], eps, emulate), emulate)
        g = _r(_lin(h, w[p + "mlp.gate_proj.weight"], None, emulate), emulate)
        u = _r(_lin(h, w[p + "mlp.up_proj.weight"], None, emulate), emulate)
        f = _r(_r(F.silu(g), emulate) * u, emulate)
        x = _r(_lin(f, w[p + "mlp.down_proj.weight"], None, emulate) + x, emulate)
    x = _r(rmsnorm(x, w[prefix + "norm.weight"], eps, emulate), emulate)
    return x, new_cache


def lm_logits(w, hidden, emulate=False):
    return _lin(hidden, w["lm_head.weight"], None, emulate)


def greedy_decode(w, inputs_embeds, embed_fn, heads, n_new, eps=1e-6, emulate=False, n_layers=None):
    """Greedy generation from prompt embeddings (generate(do_sample=False), app.py:294-300 with
    sampling off): returns the list of generated token ids for batch element 0."""
    h, cache = llama_forward(w, inputs_embeds, heads, eps, emulate=emulate, n_layers=n_layers)
    ids, pos = [], inputs_embeds.shape[1]
    logits_trace = []
    for _ in range(n_new):
        lg = lm_logits(w, h[:, -1:], emulate)
        logits_trace.append(lg[0, 0])
        nxt = int(lg[0, 0].argmax())
        ids.append(nxt)
        e = embed_fn(torch.tensor([[nxt]]))
        h, cache = llama_forward(w, e, heads, eps, kv_cache=cache, pos0=pos, emulate=emulate, n_layers=n_layers)
        pos += 1
    return ids, logits_trace
