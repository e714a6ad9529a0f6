"""Seeded synthetic weights / inputs of the reference's shapes (no checkpoints or datasets exist in
this environment).  Used by bench.py, __graft_entry__.smoke() and the parity tests; the same
tensors feed the CPU oracle and the HIP pipeline."""
import math
from types import SimpleNamespace

import torch

# ids in the order initialize_vision_tokenizer adds them (spi_llava.py:248-258):
# <im_patch>, <bbox>, <point>, <im_start>, <im_end> appended to the 32000-token LLaMA vocabulary
def token_ids(vocab_base=32000):
    return SimpleNamespace(im_patch_token=vocab_base, bbox_token=vocab_base + 1, point_token=vocab_base + 2,
                           im_start_token=vocab_base + 3, im_end_token=vocab_base + 4, use_im_start_end=True,
                           vocab=vocab_base + 6)


CLIP_L14 = dict(hidden=1024, inter=4096, layers=24, heads=16)
LLAMA_7B = dict(hidden=4096, inter=11008, layers=32, heads=32)


def _randn(shape, std, gen, device, dtype):
    if device == "cpu" or gen is not None and gen.device.type == "cpu":
        return (torch.randn(shape, generator=gen) * std).to(device=device, dtype=dtype)
    return (torch.randn(shape, generator=gen, device=device, dtype=torch.float32) * std).to(dtype)


def vit_state(hidden, inter, layers, image_size, seed=0, device="cpu", dtype=torch.float32):
    g = torch.Generator(device=device).manual_seed(seed)
    n = (image_size // 14) ** 2 + 1
    sd = {"embeddings.class_embedding": _randn((hidden,), 0.5, g, device, dtype),
          "embeddings.patch_embedding.weight": _randn((hidden, 3, 14, 14), 1.0 / math.sqrt(588), g, device, dtype),
          "embeddings.position_embedding.weight": _randn((n, hidden), 0.5, g, device, dtype),
          "pre_layrnorm.weight": 1 + _randn((hidden,), 0.1, g, device, dtype),
          "pre_layrnorm.bias": _randn((hidden,), 0.1, g, device, dtype)}
    for i in range(layers):
        p = f"encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{nm}.weight"] = _randn((hidden, hidden), 1.0 / math.sqrt(hidden), g, device, dtype)
            sd[p + f"self_attn.{nm}.bias"] = _randn((hidden,), 0.05, g, device, dtype)
        for nm in ("layer_norm1", "layer_norm2"):
            sd[p + nm + ".weight"] = 1 + _randn((hidden,), 0.1, g, device, dtype)
            sd[p + nm + ".bias"] = _randn((hidden,), 0.05, g, device, dtype)
        sd[p + "mlp.fc1.weight"] = _randn((inter, hidden), 1.0 / math.sqrt(hidden), g, device, dtype)
        sd[p + "mlp.fc1.bias"] = _randn((inter,), 0.05, g, device, dtype)
        sd[p + "mlp.fc2.weight"] = _randn((hidden, inter), 0.5 / math.sqrt(inter), g, device, dtype)
        sd[p + "mlp.fc2.bias"] = _randn((hidden,), 0.05, g, device, dtype)
    return sd


def llama_state(hidden, inter, layers, vocab, seed=1, device="cpu", dtype=torch.float32):
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {"model.embed_tokens.weight": _randn((vocab, hidden), 1.0, g, device, dtype),
          "model.norm.weight": 1 + _randn((hidden,), 0.1, g, device, dtype),
          "lm_head.weight": _randn((vocab, hidden), 1.0 / math.sqrt(hidden), g, device, dtype)}
    for i in range(layers):
        p = f"model.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj"):
            sd[p + f"self_attn.{nm}.weight"] = _randn((hidden, hidden), 1.0 / math.sqrt(hidden), g, device, dtype)
        sd[p + "self_attn.o_proj.weight"] = _randn((hidden, hidden), 0.5 / math.sqrt(hidden), g, device, dtype)
        sd[p + "mlp.gate_proj.weight"] = _randn((inter, hidden), 1.0 / math.sqrt(hidden), g, device, dtype)
        sd[p + "mlp.up_proj.weight"] = _randn((inter, hidden), 1.0 / math.sqrt(hidden), g, device, dtype)
        sd[p + "mlp.down_proj.weight"] = _randn((hidden, inter), 0.5 / math.sqrt(inter), g, device, dtype)
        sd[p + "input_layernorm.weight"] = 1 + _randn((hidden,), 0.1, g, device, dtype)
        sd[p + "post_attention_layernorm.weight"] = 1 + _randn((hidden,), 0.1, g, device, dtype)
    return sd


def spi_state(module, seed=2, device=None):
    """Weights for a MLVLROIQueryModule-shaped module (ours or the oracle's): every state_dict entry in
    sorted key order from one generator (identical to oracle.spi_oracle.synthetic_state)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    ref = module.state_dict()
    for k in sorted(ref.keys()):
        v = ref[k]
        if k.endswith('gn.weight') or (k.endswith('.weight') and v.dim() == 1):
            t = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith('.bias'):
            t = 0.05 * torch.randn(v.shape, generator=g)
        else:
            t = torch.randn(v.shape, generator=g) * (1.5 / math.sqrt(v[0].numel()))
        sd[k] = t.to(v.dtype)
    return sd


def spi_state_gpu(module, seed=2):
    """Same distribution as spi_state but generated on the module's device (the 205 M-parameter
    flatten_linear makes a CPU round trip slow); NOT bit-identical to spi_state."""
    dev = next(module.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for k, v in sorted(module.state_dict().items()):
            if k.endswith('gn.weight') or (k.endswith('.weight') and v.dim() == 1):
                v.copy_(1.0 + 0.1 * torch.randn(v.shape, generator=g, device=dev))
            elif k.endswith('.bias'):
                v.copy_(0.05 * torch.randn(v.shape, generator=g, device=dev))
            else:
                v.copy_(torch.randn(v.shape, generator=g, device=dev) * (1.5 / math.sqrt(v[0].numel())))


def boxes(n, gen):
    """x1,y1 ~ U(0,.6), w,h ~ U(.05,.35), normalised xyxy (mirrors app.py:120-121)."""
    xy = torch.rand(n, 2, generator=gen) * 0.6
    wh = torch.rand(n, 2, generator=gen) * 0.3 + 0.05
    return torch.cat([xy, xy + wh], 1)


def prompt_ids(ids, P, n_regions, gen, sys_len=40, question_len=20, vocab_base=32000):
    """[bos] + sys + <im_start> + P^2 x <im_patch> + <im_end> + n x ("region", digit, <bbox>, ",") +
    question  (SURVEY.md 8d synthetic input contract)."""
    r = lambda k: torch.randint(3, vocab_base, (k,), generator=gen).tolist()
    seq = [1] + r(sys_len) + [ids.im_start_token] + [ids.im_patch_token] * (P * P) + [ids.im_end_token]
    for _ in range(n_regions):
        seq += r(2) + [ids.bbox_token] + r(1)
    seq += r(question_len)
    return torch.tensor(seq, dtype=torch.int64)
