"""CLIP ViT-L/14 patch encoder on the gfx950 kernels.

Carries the arithmetic the reference delegates to HF `CLIPVisionModel(images,
output_hidden_states=True)` at /root/reference/gpt4roi/models/spi_llava.py:66-67 (tower built at
llava/model/llava.py:48,61-66): conv14/14 patch embedding (no bias) + class token + learned
positions, pre_layrnorm, N pre-LN blocks (biased q/k/v/out, 16 heads x 64, QuickGELU MLP).
Weights come from an HF-named state dict (`embeddings.patch_embedding.weight`, ...,
optionally prefixed `vision_model.`).  Only the hidden states the region path consumes are kept,
and the encoder stops after the deepest consumed layer (the reference runs all 24 although
hidden_states[-1] is never read, spi_llava.py:58-82).
"""
import torch

from . import kernels as K


class ClipVisionTower:
    def __init__(self, state_dict, heads=16, eps=1e-5, device="cuda", select_layer=-2, num_levels=4,
                 num_layers=None, dtype=torch.bfloat16):
        """dtype: 16-bit storage type (bf16, or fp16 as the reference serves it: app.py:96 `vision_tower.to(dtype=float16)`)."""
        sd = state_dict
        assert dtype in K.H16
        self.dtype = dtype
        pre = "vision_model." if "vision_model.pre_layrnorm.weight" in sd else ""
        bf = dtype

        def g(name, dtype=bf):
            return sd[pre + name].detach().to(device=device, dtype=dtype).contiguous()

        pw = sd[pre + "embeddings.patch_embedding.weight"]
        self.hidden = pw.shape[0]
        self.patch = pw.shape[-1]
        assert self.patch == 14, "the im2col kernel is specialised to CLIP's 14x14 patches"
        self.heads, self.eps = heads, eps
        self.kpad = 640
        w = torch.zeros(self.hidden, self.kpad, dtype=bf, device=device)
        w[:, :588] = pw.detach().reshape(self.hidden, 588).to(device=device, dtype=bf)
        self.w_patch = w
        self.cls = g("embeddings.class_embedding")
        self.pos = g("embeddings.position_embedding.weight")
        self.pre_ln = (g("pre_layrnorm.weight", torch.float32), g("pre_layrnorm.bias", torch.float32))
        if num_layers is None:
            num_layers = sum(1 for k in sd if k.startswith(pre + "encoder.layers.") and k.endswith("layer_norm1.weight"))
        self.num_layers = num_layers
        # hidden-state indices consumed downstream (spi_llava.py:58-82)
        idx = list(range(num_layers + 1))
        self.image_feature_index = idx[select_layer]
        self.level_indices = idx[select_layer::-3][::-1][-num_levels:]
        self.last_needed = max([self.image_feature_index] + self.level_indices)
        self.layers = []
        for i in range(self.last_needed):
            p = f"encoder.layers.{i}."
            wqkv = torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"),
                              g(p + "self_attn.v_proj.weight")], 0).contiguous()
            bqkv = torch.cat([g(p + "self_attn.q_proj.bias"), g(p + "self_attn.k_proj.bias"),
                              g(p + "self_attn.v_proj.bias")], 0).float().contiguous()
            self.layers.append(dict(
                ln1=(g(p + "layer_norm1.weight", torch.float32), g(p + "layer_norm1.bias", torch.float32)),
                wqkv=wqkv, bqkv=bqkv,
                wo=g(p + "self_attn.out_proj.weight"), bo=g(p + "self_attn.out_proj.bias").float().contiguous(),
                ln2=(g(p + "layer_norm2.weight", torch.float32), g(p + "layer_norm2.bias", torch.float32)),
                w1=g(p + "mlp.fc1.weight"), b1=g(p + "mlp.fc1.bias").float().contiguous(),
                w2=g(p + "mlp.fc2.weight"), b2=g(p + "mlp.fc2.bias").float().contiguous()))

    @torch.no_grad()
    def forward(self, images):
        """images [B,3,S,S] (any float dtype) -> dict {hidden_state_index: [B, n+1, C] bf16}."""
        B, _, S, _ = images.shape
        C, H = self.hidden, self.heads
        D = C // H
        cols = K.im2col_patch14(images.float(), self.kpad, dtype=self.dtype)
        patch = K.gemm(cols, self.w_patch)
        n = patch.size(0) // B
        assert n + 1 <= self.pos.size(0), "image larger than the position table"
        tok = K.vit_assemble(patch, self.cls, self.pos, B)
        T = n + 1
        x = K.layernorm(tok.view(B * T, C), self.pre_ln[0], self.pre_ln[1], self.eps)
        keep = {}
        wanted = set(self.level_indices + [self.image_feature_index])
        if 0 in wanted:
            keep[0] = x.view(B, T, C)
        scale = D ** -0.5
        h = None                    # layer_norm1 output when the previous block's fc2 reduce already produced it
        for i, L in enumerate(self.layers):
            if h is None:
                h = K.layernorm(x, L['ln1'][0], L['ln1'][1], self.eps)
            qkv = K.gemm(h, L['wqkv'], bias=L['bqkv']).view(B, T, 3 * C)
            a = K.flash_attn(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], H, scale, False)
            x = K.gemm(a.view(B * T, C), L['wo'], bias=L['bo'], residual=x)
            h = K.layernorm(x, L['ln2'][0], L['ln2'][1], self.eps)
            f = K.gemm(h, L['w1'], bias=L['b1'], act='quick_gelu')
            plan = K.small_m_split_plan(B * T, C, f.size(1)) if i + 1 < len(self.layers) else None
            if plan is not None:
                # batch-1 tower: fc2 runs as K slices; their reduce (+ bias, + residual) and the NEXT block's layer_norm1 are
                # one pass over the row
                part, ns = K.gemm_partials(f, L['w2'], plan[1], plan[0])
                nxt = self.layers[i + 1]['ln1']
                x, h = K.layernorm_splitk(part, ns, L['b2'], x, nxt[0], nxt[1], self.eps)
            else:
                x = K.gemm(f, L['w2'], bias=L['b2'], residual=x)
                h = None
            if i + 1 in wanted:
                keep[i + 1] = x.view(B, T, C)
        return keep

    def select(self, keep):
        """(image_features [B,n,C], [4 x [B,n,C]]) as spi_llava.py:58-82 (CLS dropped; strided views)."""
        return keep[self.image_feature_index][:, 1:], [keep[i][:, 1:] for i in self.level_indices]
