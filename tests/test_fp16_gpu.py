"""GPU: the IEEE-half (fp16) instantiation of the inference kernels -- the reference's SERVING dtype.

gpt4roi/app.py:74-98 loads the model with torch_dtype=float16, :271 makes the boxes .half() and :296 the image .half():
BASELINE configs[1] (the benchmarked inference path) runs fp16 in the reference.  libgpt4roi_hip.so carries every
inference kernel a second time, compiled from the same sources with -DG4R_F16 (csrc/g4r_common.h; entry points in
include/g4r_f16_names.h); kernels.py picks the instantiation by the tensors' dtype.  This file checks each kernel family of
that instantiation against plain fp32 torch arithmetic on the same fp16 inputs, and the assembled stages against the
oracles run with fp16 rounding points (`emulate=torch.float16`).  The full-depth greedy-id test lives in
test_fullwidth_gpu.py.

Tolerances: an fp16 output is one rounding (2^-11 relative) of an fp32 accumulation, so |err| <= 2^-10 |ref| + an
accumulation-order term (<= 2e-3 of the tensor's max for the K <= 11008 contractions here)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import spi_oracle as S  # noqa: E402
from oracle import transformer_oracle as T  # noqa: E402

if torch.cuda.is_available():
    from gpt4roi_amd import kernels as K
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.layers import MLVLROIQueryModule
    from gpt4roi_amd.llama import LlamaDecoder
    from gpt4roi_amd.spi_llava import SPILlavaLlamaModel
    from gpt4roi_amd.vit import ClipVisionTower

DEV = "cuda"
H = torch.float16


def rnd(*shape, std=1.0, seed=0, dtype=H):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=DEV) * std).to(dtype)


def close(got, want, what, rel=2 ** -10, floor=2e-3):
    got, want = got.float(), want.float()
    tol = rel * want.abs() + floor * want.abs().max()
    bad = ((got - want).abs() > tol)
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} outside tolerance, max err {(got - want).abs().max().item():.3e}"


@pytest.mark.parametrize("M,N,K_,act,extra", [
    (767, 12288, 4096, None, None), (767, 4096, 4096, None, "residual"), (767, 22016, 4096, "swiglu", None),
    (767, 4096, 11008, None, "residual"), (767, 32006, 4096, None, "f32out"), (577, 3072, 1024, None, "bias"),
    (577, 4096, 1024, "quick_gelu", "bias"), (577, 1024, 4096, None, "bias+residual"), (1, 4096, 4096, None, None),
    (8, 12288, 4096, None, None), (1534, 12288, 4096, None, None), (200, 328, 192, "relu", "bias")])
def test_f16_gemm_production_shapes(M, N, K_, act, extra):
    a, w = rnd(M, K_, seed=1), rnd(N, K_, std=1.0 / math.sqrt(K_), seed=2)
    bias = rnd(N, std=0.1, seed=3, dtype=torch.float32) if extra and "bias" in extra else None
    n_out = N // 2 if act == "swiglu" else N
    res = rnd(M, N, seed=4) if extra and "residual" in extra else None
    out_dtype = torch.float32 if extra == "f32out" else None
    got = K.gemm(a, w, bias=bias, residual=res, act=act, out_dtype=out_dtype)
    assert got.dtype == (torch.float32 if extra == "f32out" else H) and got.shape == (M, n_out)
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    if act == "relu":
        y = torch.relu(y)
    elif act == "quick_gelu":
        y = y * torch.sigmoid(1.702 * y)
    elif act == "swiglu":
        g_, u_ = y[:, 0::2], y[:, 1::2]
        y = torch.nn.functional.silu(g_).to(H).float() * u_       # the epilogue rounds silu(g) to the storage type (as HF does)
    if res is not None:
        y = y + res.float()
    close(got, y, f"f16 gemm {M}x{N}x{K_} {act} {extra}")


def test_f16_and_bf16_are_distinct_instantiations_and_do_not_mix():
    a, w = rnd(256, 512, seed=5), rnd(384, 512, std=0.05, seed=6)
    y16 = K.gemm(a, w)
    yb = K.gemm(a.to(torch.bfloat16), w.to(torch.bfloat16))
    ref = a.float() @ w.float().t()
    e16 = (y16.float() - ref).abs().max().item()
    eb = (yb.float() - a.to(torch.bfloat16).float() @ w.to(torch.bfloat16).float().t()).abs().max().item()
    assert y16.dtype == H and yb.dtype == torch.bfloat16 and e16 < eb / 3, (e16, eb)     # 3 more mantissa bits
    with pytest.raises(TypeError):
        K.gemm(a, w.to(torch.bfloat16))
    with pytest.raises(TypeError):
        K.rmsnorm(a.float(), torch.ones(512, device=DEV))


@pytest.mark.parametrize("B,Tq,Tk,heads,D,causal", [(1, 767, 767, 32, 128, True), (2, 577, 577, 16, 64, False),
                                                    (1, 5, 300, 8, 128, True), (2, 699, 699, 4, 128, True)])
def test_f16_flash_attention(B, Tq, Tk, heads, D, causal):
    q, k, v = (rnd(B, t, heads * D, seed=s) for t, s in ((Tq, 1), (Tk, 2), (Tk, 3)))
    got = K.flash_attn(q, k, v, heads, D ** -0.5, causal)
    want = T.attention(q.float(), k.float(), v.float(), heads, D ** -0.5, causal, emulate=False)
    # P is rounded to fp16 before the PV product (11 bits): ~2^-11 relative per term, averaged over the keys
    close(got, want, f"f16 attention {B}x{Tq}x{Tk} h{heads} d{D}", rel=2 ** -9, floor=2e-3)


def test_f16_norms_and_glue():
    x = rnd(300, 1024, seed=7)
    g, b = rnd(1024, std=0.3, seed=8, dtype=torch.float32) + 1, rnd(1024, std=0.1, seed=9, dtype=torch.float32)
    close(K.layernorm(x, g, b, 1e-5), torch.nn.functional.layer_norm(x.float(), (1024,), g, b, 1e-5), "f16 layernorm")
    xf = x.float()
    close(K.rmsnorm(x, g, 1e-6), xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * g, "f16 rmsnorm")
    gu = rnd(64, 2048, seed=10)
    want = torch.nn.functional.silu(gu[:, :1024].float()) * gu[:, 1024:].float()
    close(K.swiglu(gu), want, "f16 swiglu", rel=2 ** -9)
    close(K.add_rows(x, x), 2 * xf, "f16 add_rows")
    c = K.cast_bf16(xf, dtype=H)
    assert c.dtype == H and torch.equal(c, x)


def test_f16_conv3x3_and_groupnorm():
    x = rnd(2, 24, 24, 128, seed=11)
    wt = rnd(256, 128, 3, 3, std=0.03, seed=12, dtype=torch.float32)
    got = K.conv3x3(x, K.prep_conv3x3_weight(wt, H))
    want = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wt.to(H).float(), padding=1).permute(0, 2, 3, 1)
    close(got, want, "f16 conv3x3")
    gam, bet = torch.ones(256, device=DEV), torch.zeros(256, device=DEV)
    ss = K.groupnorm_affine(got, gam, bet, 16, 1e-5)
    y = got.float() * ss[:, 0][:, None, None, :] + ss[:, 1][:, None, None, :]
    ref = torch.nn.functional.group_norm(got.float().permute(0, 3, 1, 2), 16, gam, bet, 1e-5).permute(0, 2, 3, 1)
    close(y, ref, "f16 deferred groupnorm", rel=1e-3, floor=1e-3)


def test_f16_roi_align_mlvl_is_the_fp32_kernel_up_to_one_rounding():
    B, C, P = 2, 256, 8
    g = torch.Generator().manual_seed(9)
    feats32 = [torch.randn(B, s, s, C, generator=g).to(DEV) for s in (8 * P, 4 * P, 2 * P, P)]
    rois = torch.cat([torch.randint(0, B, (24, 1), generator=g).float(), syn.boxes(24, g) * 14 * P], 1).to(DEV)
    scales = [8 / 14, 4 / 14, 2 / 14, 1 / 14]
    f16 = [f.to(H) for f in feats32]
    want = K.roi_align_mlvl([f.float() for f in f16], rois, 14, scales)
    got = K.roi_align_mlvl(f16, rois, 14, scales)
    assert got.dtype == H
    close(got, want, "f16 mlvl roi_align vs the fp32 instantiation", rel=2 ** -10, floor=1e-4)


def test_f16_region_module_vs_oracle_with_fp16_rounding_points():
    C, P = 512, 8
    m = MLVLROIQueryModule(embed_dims=C, out_dims=512, num_levels=4).set_compute_dtype(H)
    o = S.MLVLROIQueryOracle(embed_dims=C, P=P)
    o.roi_align.updims = torch.nn.Linear(1024, 512)
    sd = S.synthetic_state(o, 1)
    o.load_state_dict(sd)
    m.load_state_dict(sd)
    m.to(DEV)
    feats, boxes = S.synthetic_inputs(2, 2, P, C, [3, 5])
    with torch.no_grad():
        got = torch.cat(m([f.to(DEV).to(H) for f in feats], [b.to(DEV) for b in boxes])).float().cpu()
        ref16 = torch.cat(o([f.to(H).float() for f in feats], boxes, emulate=torch.float16))
        ref32 = torch.cat(o([f.to(H).float() for f in feats], boxes, emulate=False))
        mb = MLVLROIQueryModule(embed_dims=C, out_dims=512, num_levels=4)
        mb.load_state_dict(sd)
        mb.to(DEV)
        gotb = torch.cat(mb([f.to(DEV).to(H).to(torch.bfloat16) for f in feats], [b.to(DEV) for b in boxes])).float().cpu()
    e16 = ((got - ref16).abs().max() / ref16.abs().max()).item()
    e32 = ((got - ref32).abs().max() / ref32.abs().max()).item()
    eb = ((gotb - ref32).abs().max() / ref32.abs().max()).item()
    print(f"region module fp16: {e16:.2e} vs the fp16-emulating oracle, {e32:.2e} vs exact fp32 (bf16 instantiation: {eb:.2e})")
    assert e16 < 4e-3 and e32 < 4e-3 and e32 < eb / 2


def test_f16_small_pipeline_prefill_decode_and_greedy_ids_vs_oracle():
    """ViT -> region module -> projector -> splice -> 3-layer decoder, all fp16: logits against the fp16-emulating oracles,
    then KV-cache decode (fused RMSNorm GEMV, split-key attention) -- greedy ids identical to the oracle's."""
    C, P, Hd, heads, inter, L = 512, 8, 512, 4, 1408, 3
    ids = syn.token_ids(990)
    vsd = syn.vit_state(C, 4 * C, 12, 14 * P, seed=3)
    lsd = syn.llama_state(Hd, inter, L, ids.vocab, seed=4)
    vsd = {k: v.to(H).float() for k, v in vsd.items()}
    lsd = {k: v.to(H).float() for k, v in lsd.items()}
    tower = ClipVisionTower(vsd, heads=8, device=DEV, dtype=H)
    dec = LlamaDecoder(lsd, heads=heads, max_positions=256, device=DEV, dtype=H)
    model = SPILlavaLlamaModel(tower, dec, ids, embed_dims=C)
    orc = S.MLVLROIQueryOracle(embed_dims=C, P=P)
    orc.roi_align.updims = torch.nn.Linear(1024, Hd)
    sd = S.synthetic_state(orc, 5)
    orc.load_state_dict(sd)
    model.spi_module.load_state_dict(sd)
    g = torch.Generator().manual_seed(6)
    img = torch.randn(1, 3, 14 * P, 14 * P, generator=g).to(H).float()
    boxes = [syn.boxes(3, g)]
    prompt = syn.prompt_ids(ids, P, 3, g, sys_len=4, question_len=3, vocab_base=990)[None]
    with torch.no_grad():
        pw, pb = model.mm_projector.weight.detach().clone(), model.mm_projector.bias.detach().clone()
        emb_hip = model.embed_inputs(prompt.to(DEV), img.to(DEV), [b.to(DEV) for b in boxes])
        model.check_status()
        hs = T.clip_vit_hidden_states(vsd, img, heads=8, emulate=torch.float16)
        img_feat, lv = T.select_spi_levels(hs, -2, 4)
        spi = orc(lv, boxes, emulate=torch.float16)
        r = lambda t: t.to(H).float()
        proj = r(r(img_feat) @ r(pw).t() + r(pb))
        spliced = S.splice(prompt, lsd["model.embed_tokens.weight"][prompt], proj, spi, ids.im_start_token, ids.im_end_token,
                           ids.bbox_token)
        e_emb = ((emb_hip.float().cpu() - spliced).abs().max() / spliced.abs().max()).item()
        want_ids, _ = T.greedy_decode(lsd, spliced, lambda t: lsd["model.embed_tokens.weight"][t], heads=heads, n_new=8,
                                      emulate=torch.float16)
        got_ids = dec.greedy(spliced.to(DEV).to(H), 8)
        h, _ = T.llama_forward(lsd, spliced, heads, emulate=torch.float16)
        want = T.lm_logits(lsd, h, torch.float16)
        dec.reset(1)
        got = dec.forward(spliced.to(DEV).to(H)).float().cpu()
    e_log = ((got - want).abs().max() / want.abs().max()).item()
    print(f"fp16 small pipeline: inputs_embeds {e_emb:.2e}, logits {e_log:.2e}; ids {got_ids} vs {want_ids}")
    assert e_emb < 4e-3 and e_log < 4e-3
    assert got_ids == want_ids


def test_f16_batched_and_graph_decode_equal_the_single_sequence_loop():
    """The device-resident decode loops in fp16: the hipGraph loop (greedy and sampled) and the batched loop give the ids of the
    plain host loop -- the GEMV with the fused RMSNorm, the split-key attention with RoPE + cache append, the small-M GEMM
    tiles and the row gather all run their -DG4R_F16 instantiation."""
    Hd, heads, inter, L = 512, 4, 1408, 3
    ids = syn.token_ids(990)
    lsd = {k: v.to(H).float() for k, v in syn.llama_state(Hd, inter, L, ids.vocab, seed=14).items()}
    dec = LlamaDecoder(lsd, heads=heads, max_positions=256, device=DEV, dtype=H, max_batch=3)
    g = torch.Generator().manual_seed(15)
    emb = (torch.randn(3, 40, Hd, generator=g) * 0.7).to(DEV).to(H)
    with torch.no_grad():
        single = [dec.greedy(emb[b:b + 1], 9) for b in range(3)]
        assert dec.greedy_graph(emb[0:1], 9) == single[0]
        assert dec.decode_graph_batch(emb, 9) == single
        want, _ = T.greedy_decode(lsd, emb[1:2].float().cpu(), lambda t: lsd["model.embed_tokens.weight"][t], heads=heads, n_new=9,
                                  emulate=torch.float16)
    assert single[1] == want


def test_roi_align_mlvl_edge_boxes_vs_the_reference_cpu_op():
    """the PRODUCTION multi-level NHWC kernel on edge boxes against a fixture made by the reference's own
    compiled CPU op (tests/golden/make_golden.py::edges_mlvl -> oracle/_ref): boxes off every border, fully outside,
    zero-area, whole-image, sub-pixel thin, on the last row / column.  fp32 instantiation <= 1e-5 (north_star: 1e-4);
    the bf16 and fp16 instantiations (LDS-staged rows for narrow RoIs, direct gathers for wide ones) read the same values
    exactly (the maps are multiples of 1/16) and may differ by their output rounding only."""
    import os
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "roi_align_edges_mlvl.npz"))
    rois = torch.from_numpy(d["rois"]).to(DEV)
    scales = [1.0 / s for s in d["strides"]]
    maps = [torch.from_numpy(d[f"x{l}"]).to(DEV).permute(0, 2, 3, 1).contiguous() for l in range(4)]
    want = torch.stack([torch.from_numpy(d[f"out{l}"]).permute(0, 2, 3, 1) for l in range(4)]).to(DEV)     # [L, N, 14, 14, C]
    got = K.roi_align_mlvl(maps, rois, 14, scales, sampling_ratio=2, aligned=True)
    e32 = (got - want).abs().max().item()
    print(f"mlvl edge boxes, fp32 instantiation vs the reference CPU op: max |err| {e32:.2e}")
    assert e32 <= 1e-5
    for dt, ulp in ((torch.bfloat16, 2 ** -8), (torch.float16, 2 ** -11)):
        m16 = [m.to(dt) for m in maps]
        assert all(torch.equal(a.float(), b) for a, b in zip(m16, maps))
        g16 = K.roi_align_mlvl(m16, rois, 14, scales, sampling_ratio=2, aligned=True).float()
        err = (g16 - want).abs()
        tol = ulp * want.abs() + 1e-5 + 2e-6 * want.abs().max()
        assert (err <= tol).all(), f"{dt}: max err {err.max().item():.3e}"
