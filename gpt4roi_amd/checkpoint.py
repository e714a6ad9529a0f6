"""HF-layout checkpoint IO for the region-feature path.

What the reference's callers do with checkpoints, and where it is mirrored here:
  * `SPILlavaMPTForCausalLM.from_pretrained(dir)`   gpt4roi/app.py:68-76, gpt4roi/train/train.py:552-557,
                                                     scripts/apply_delta.py:21   -> `from_pretrained`
  * `model.save_pretrained(dir)`                     scripts/apply_delta.py:42   -> `save_pretrained`
  * `apply_delta` / `make_delta`                     scripts/apply_delta.py:15-43, scripts/make_delta.py:14-53
  * `CLIPVisionModel.from_pretrained(mm_vision_tower)` llava/model/llava.py:48,61 -> `load_vision_tower`
A checkpoint directory is what HF writes: `config.json` + `pytorch_model.bin` | `model.safetensors` (optionally sharded
with an `*.index.json`).  Keys: `model.embed_tokens.weight`, `model.layers.N.*`,
`model.norm.weight`, `lm_head.weight`, `model.mm_projector.{weight,bias}`, `model.spi_module.*`; the vision tower is NOT
in the state dict (llava.py:48 keeps it in a Python list) and is loaded from `config.mm_vision_tower`, which must be a
local directory here (there is no hub access).  Host-side code: the tensors land in the kernel-ready buffers of
LlamaDecoder / ClipVisionTower / MLVLROIQueryModule.
"""
import json
import os
import shutil
from types import SimpleNamespace

import torch

BIN, BIN_INDEX = "pytorch_model.bin", "pytorch_model.bin.index.json"
SAFE, SAFE_INDEX = "model.safetensors", "model.safetensors.index.json"


# ------------------------------------------------------------------------------------------------ state-dict files
def load_hf_state_dict(path):
    """Every tensor of an HF checkpoint directory (single file or sharded, torch pickle or safetensors), on the CPU."""
    def one(fn):
        if fn.endswith(".safetensors"):
            from safetensors.torch import load_file
            return load_file(os.path.join(path, fn))
        return torch.load(os.path.join(path, fn), map_location="cpu", weights_only=True)
    for index in (SAFE_INDEX, BIN_INDEX):
        if os.path.exists(os.path.join(path, index)):
            with open(os.path.join(path, index)) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
            sd = {}
            for fn in files:
                sd.update(one(fn))
            return sd
    for fn in (SAFE, BIN):
        if os.path.exists(os.path.join(path, fn)):
            return one(fn)
    raise FileNotFoundError(f"no {SAFE} / {BIN} (or sharded index) under {path}")


def save_hf_state_dict(sd, path, safe_serialization=True, max_shard_bytes=5 << 30):
    """Writes `sd` the way HF `save_pretrained` does (shards of <= max_shard_bytes + an index when it does not fit one)."""
    os.makedirs(path, exist_ok=True)
    sd = {k: v.detach().to("cpu").contiguous() for k, v in sd.items()}
    shards, cur, cur_b = [], {}, 0
    for k, v in sd.items():
        b = v.numel() * v.element_size()
        if cur and cur_b + b > max_shard_bytes:
            shards.append(cur)
            cur, cur_b = {}, 0
        cur[k] = v
        cur_b += b
    shards.append(cur)
    ext = ".safetensors" if safe_serialization else ".bin"

    def write(d, fn):
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(d, os.path.join(path, fn), metadata={"format": "pt"})
        else:
            torch.save(d, os.path.join(path, fn))
    if len(shards) == 1:
        write(shards[0], SAFE if safe_serialization else BIN)
        return
    base = "model" if safe_serialization else "pytorch_model"
    wmap = {}
    for i, d in enumerate(shards):
        fn = f"{base}-{i + 1:05d}-of-{len(shards):05d}{ext}"
        write(d, fn)
        wmap.update({k: fn for k in d})
    with open(os.path.join(path, SAFE_INDEX if safe_serialization else BIN_INDEX), "w") as f:
        json.dump({"metadata": {"total_size": sum(v.numel() * v.element_size() for v in sd.values())},
                   "weight_map": wmap}, f, indent=2)


def _read_config(path):
    with open(os.path.join(path, "config.json")) as f:
        return json.load(f)


# ------------------------------------------------------------------------------------------------ vision tower
class ClipImageProcessor:
    """The three numbers the callers read off `CLIPImageProcessor` (app.py:125-129, refcoco.py:69-72); the pixel work
    itself is the fused kernel kernels.image_preprocess."""

    def __init__(self, size=224):
        from .kernels import CLIP_MEAN, CLIP_STD
        self.size, self.image_mean, self.image_std = size, list(CLIP_MEAN), list(CLIP_STD)
        self.crop_size = {"height": size, "width": size}

    def preprocess(self, image_u8_hwc, size=None):
        from . import kernels as K
        return K.image_preprocess(image_u8_hwc, size or self.size)


def load_vision_tower(source, device="cuda", select_layer=-2, dtype=torch.bfloat16):
    """`CLIPVisionModel.from_pretrained(config.mm_vision_tower)` (llava.py:48,61): `source` is a local HF CLIP directory
    (config.json with a `vision_config` or flat vision fields) or an HF-keyed state dict."""
    from .vit import ClipVisionTower
    heads, eps, image_size = 16, 1e-5, None
    if isinstance(source, dict):
        sd = source
    elif isinstance(source, str) and os.path.isdir(source):
        sd = load_hf_state_dict(source)
        cfg = _read_config(source)
        vc = cfg.get("vision_config", cfg)
        heads = vc.get("num_attention_heads", heads)
        eps = vc.get("layer_norm_eps", eps)
        image_size = vc.get("image_size")
    else:
        raise FileNotFoundError(
            f"vision tower {source!r}: a hub name cannot be resolved here (no network); pass a local directory or state dict")
    pre = "vision_model." if "vision_model.pre_layrnorm.weight" in sd else ""
    tower = ClipVisionTower(sd, heads=heads, eps=eps, device=device, select_layer=select_layer, dtype=dtype)
    n_pos = sd[pre + "embeddings.position_embedding.weight"].shape[0]
    tower.image_size = image_size or int(round((n_pos - 1) ** 0.5)) * tower.patch
    return tower


# ------------------------------------------------------------------------------------------------ model directory
def model_config(model, extra=None):
    """config.json of a checkpoint this path writes: the LLaMA fields HF needs + the LLaVA / GPT4RoI additions."""
    dec = model.model.llama
    mc = model.model.config
    cfg = dict(architectures=["SPILlavaMPTForCausalLM"], model_type="llava", hidden_size=dec.hidden,
               intermediate_size=dec.inter, num_attention_heads=dec.heads, num_hidden_layers=len(dec.layers),
               vocab_size=dec.vocab, rms_norm_eps=dec.eps, max_position_embeddings=dec.max_positions,
               rope_theta=getattr(dec, "theta", 10000.0),
               torch_dtype="float16" if getattr(dec, "dtype", None) is torch.float16 else "bfloat16", use_mm_proj=True,
               mm_hidden_size=model.model.mm_projector.in_features,
               mm_vision_tower=getattr(mc, "mm_vision_tower", None),
               mm_vision_select_layer=getattr(mc, "mm_vision_select_layer", -2),
               mm_use_im_start_end=bool(getattr(mc, "use_im_start_end", True)))
    for k in ("im_patch_token", "im_start_token", "im_end_token", "bbox_token", "point_token"):
        if hasattr(mc, k):
            cfg[k] = int(getattr(mc, k))
    cfg.update(extra or {})
    return cfg


def save_pretrained(model, path, safe_serialization=True, max_shard_bytes=5 << 30):
    os.makedirs(path, exist_ok=True)
    save_hf_state_dict(model.state_dict(), path, safe_serialization, max_shard_bytes)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(model_config(model), f, indent=2)


def from_pretrained(cls, path, device="cuda", vision_tower=None, tokenizer=None, max_positions=None, torch_dtype=None,
                    low_cpu_mem_usage=None, use_cache=None, **_):
    """Builds the MI355X model from an HF-layout directory.  `torch_dtype` selects the 16-bit storage type of the kernels:
    torch.float16 as app.py:70-75 passes it (the reference serves in fp16), torch.bfloat16 (the default, and what anything
    else maps to) as the training scripts use; accumulation is fp32 either way.  `low_cpu_mem_usage` / `use_cache` are
    accepted for call compatibility."""
    dtype = torch.float16 if torch_dtype in (torch.float16, "float16", "fp16", "half") else torch.bfloat16
    from . import synthetic as syn
    from .llama import LlamaDecoder
    from .spi_llava import SPILlavaLlamaModel
    cfg = _read_config(path)
    sd = load_hf_state_dict(path)
    vocab = sd["model.embed_tokens.weight"].shape[0]
    dec = LlamaDecoder(sd, heads=cfg["num_attention_heads"], eps=cfg.get("rms_norm_eps", 1e-6),
                       theta=cfg.get("rope_theta", 10000.0),
                       max_positions=max_positions or min(cfg.get("max_position_embeddings", 2048), 4096), device=device,
                       num_layers=cfg.get("num_hidden_layers"), dtype=dtype)
    dec.theta = cfg.get("rope_theta", 10000.0)
    src = vision_tower if vision_tower is not None else cfg.get("mm_vision_tower")
    tower = load_vision_tower(src, device=device, select_layer=cfg.get("mm_vision_select_layer", -2), dtype=dtype) \
        if src is not None else None
    # token ids: from the config when it carries them (checkpoints this path wrote), else the positions
    # initialize_vision_tokenizer gives them (spi_llava.py:248-258) at the END of the vocabulary; a tokenizer, when
    # given, is authoritative (app.py:84-104 reads the ids off the tokenizer).
    # with mm_use_im_start_end the tokenizer appends <im_patch>, <bbox>, <point>, <im_start>, <im_end> (five rows
    # at the end of the vocabulary); without it only <im_patch>, <bbox>, <point> (three rows) -- read the flag FIRST
    use_se = bool(cfg.get("mm_use_im_start_end", True))
    ids = syn.token_ids(vocab - 5 if use_se else vocab - 3)
    if not use_se:
        ids.im_start_token = ids.im_end_token = -1          # never present in a prompt of such a checkpoint
    for k in ("im_patch_token", "im_start_token", "im_end_token", "bbox_token", "point_token"):
        if k in cfg:
            setattr(ids, k, int(cfg[k]))
    ids.use_im_start_end = use_se
    ids.vocab = vocab
    ids.mm_vision_tower = cfg.get("mm_vision_tower")
    ids.mm_vision_select_layer = cfg.get("mm_vision_select_layer", -2)
    embed_dims = cfg.get("mm_hidden_size", tower.hidden if tower is not None else 1024)
    inner = SPILlavaLlamaModel(tower, dec, ids, embed_dims=embed_dims)
    spi = {k[len("model.spi_module."):]: v for k, v in sd.items() if k.startswith("model.spi_module.")}
    if spi:
        inner.spi_module.load_state_dict({k: v.float() for k, v in spi.items()}, strict=True)
    proj = {k[len("model.mm_projector."):]: v for k, v in sd.items() if k.startswith("model.mm_projector.")}
    if proj:
        inner.mm_projector.load_state_dict({k: v.float() for k, v in proj.items()}, strict=True)
    if device != "cpu":
        inner.spi_module.to(device)
        inner.mm_projector.to(device)
    model = cls(inner, SimpleNamespace(**cfg))
    if tokenizer is not None:
        bind_tokenizer(model, tokenizer)
    return model


def bind_tokenizer(model, tokenizer):
    """What app.py:84-104 does after from_pretrained: make sure the special tokens exist and read their ids."""
    from .spi_llava import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN
    tokenizer.add_tokens([DEFAULT_IMAGE_PATCH_TOKEN], special_tokens=True)
    tokenizer.add_tokens([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN], special_tokens=True)
    tokenizer.add_tokens(['<bbox>', '<point>'], special_tokens=True)
    c = model.model.config
    c.im_patch_token = tokenizer.convert_tokens_to_ids([DEFAULT_IMAGE_PATCH_TOKEN])[0]
    c.im_start_token, c.im_end_token = tokenizer.convert_tokens_to_ids([DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN])
    c.bbox_token = tokenizer.convert_tokens_to_ids(['<bbox>'])[0]
    c.point_token = tokenizer.convert_tokens_to_ids(['<point>'])[0]
    for m in model.modules():
        m.tokenizer = tokenizer


# ------------------------------------------------------------------------------------------------ weight deltas
def _combine(delta_sd, base_sd, sign):
    """scripts/apply_delta.py:24-39 (sign = +1) / scripts/make_delta.py:29-41 (sign = -1), on state dicts: keys absent
    from the base are kept raw when they belong to the projector or the region module (else NameError, as there);
    equal shapes add element-wise; a grown embedding / lm_head adds into the top-left block."""
    out = {}
    for name, param in delta_sd.items():
        param = param.clone()
        if name not in base_sd:
            if name in ('model.mm_projector.weight', 'model.mm_projector.bias') or "spi_module" in name:
                out[name] = param
                continue
            raise NameError(name)
        b = base_sd[name]
        if param.shape == b.shape:
            param += sign * b.to(param.dtype)
        else:
            assert name in ('model.embed_tokens.weight', 'lm_head.weight'), \
                f'{name} dimension mismatch: {param.shape} vs {b.shape}'
            param[:b.shape[0], :b.shape[1]] += sign * b.to(param.dtype)
        out[name] = param
    return out


def _copy_side_files(src, dst):
    for fn in os.listdir(src):
        if fn.startswith(("tokenizer", "special_tokens", "added_tokens", "generation_config")) or fn == "config.json":
            shutil.copy(os.path.join(src, fn), os.path.join(dst, fn))


def apply_delta(base_model_path, target_model_path, delta_path, safe_serialization=True):
    """target = delta + base  (python -m scripts.apply_delta --base ... --target ... --delta ...)."""
    target = _combine(load_hf_state_dict(delta_path), load_hf_state_dict(base_model_path), +1)
    save_hf_state_dict(target, target_model_path, safe_serialization)
    _copy_side_files(delta_path, target_model_path)
    return target


def make_delta(base_model_path, target_model_path, delta_path, safe_serialization=True):
    """delta = target - base  (scripts/make_delta.py)."""
    delta = _combine(load_hf_state_dict(target_model_path), load_hf_state_dict(base_model_path), -1)
    save_hf_state_dict(delta, delta_path, safe_serialization)
    _copy_side_files(target_model_path, delta_path)
    return delta
