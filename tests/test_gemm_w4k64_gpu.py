"""GPU: the round-5 production GEMM tile (tile_cfg 34: 256 x 256, one wave per SIMD, K tiles of 64, epilogue mode chosen at
launch) against (a) the ring ping-pong tile of rounds 2-4 -- bit-identical on dense problems: same MFMA, same K order, same
rounding points -- and (b) plain fp32 torch arithmetic.  Both storage types (bf16 / fp16 instantiation).  The arithmetic is the
one the reference delegates to cuBLAS / cuDNN through nn.Linear / nn.Conv2d (gpt4roi/models/layers.py:129-144, 257-270;
llava/model/llava.py:52) and, for the fused epilogue, HF's apply_rotary_pos_emb + KV-cache append."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpt4roi_amd import kernels as K
    from gpt4roi_amd._lib import lib

DEV = "cuda"
DTYPES = [torch.bfloat16, torch.float16]


def rnd(*shape, scale=0.5, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-9)).item()


TOL = {torch.bfloat16: 6e-3, torch.float16: 1e-3}


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,Kd", [(300, 520, 128), (1000, 300, 64), (256, 256, 192), (513, 4096, 1024), (2049, 6216, 320),
                                     (767, 1024, 4096), (3068, 4096, 4096)])
def test_dense_bit_identical_to_the_ring_tile_and_close_to_fp32(M, N, Kd, dtype):
    """ragged M and N (clamped rows, the generic epilogue on the ragged column tile), one / two / three / many K tiles"""
    a, w = rnd(M, Kd, seed=1, dtype=dtype), rnd(N, Kd, seed=2, dtype=dtype)
    got = K.gemm(a, w, tile_cfg=34)
    assert torch.equal(got, K.gemm(a, w, tile_cfg=24))
    assert rel(got, a.float() @ w.float().t()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", [None, "relu", "quick_gelu", "silu"])
@pytest.mark.parametrize("residual", [False, True])
def test_epilogue_modes_bias_activation_residual(act, residual, dtype):
    """16-bit park (no residual) and fp32 park (residual), every activation; M crosses the last row tile"""
    a, w = rnd(900, 512, seed=3, dtype=dtype), rnd(768, 512, seed=4, dtype=dtype)
    bias = rnd(768, seed=5, dtype=torch.float32)
    res = rnd(900, 768, seed=6, dtype=dtype) if residual else None
    got = K.gemm(a, w, bias=bias, residual=res, act=act, tile_cfg=34)
    assert torch.equal(got, K.gemm(a, w, bias=bias, residual=res, act=act, tile_cfg=24))
    r = a.float() @ w.float().t() + bias
    r = {None: r, "relu": r.relu(), "quick_gelu": r * torch.sigmoid(1.702 * r), "silu": F.silu(r)}[act]
    if residual:
        r = r + res.float()
    assert rel(got, r) < 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_fp32_output_swiglu_and_k_slices(dtype):
    a, w = rnd(700, 384, seed=7, dtype=dtype), rnd(1024, 384, seed=8, dtype=dtype)
    bias = rnd(1024, seed=9, dtype=torch.float32)
    f32 = K.gemm(a, w, bias=bias, out_dtype=torch.float32, tile_cfg=34)
    assert torch.equal(f32, K.gemm(a, w, bias=bias, out_dtype=torch.float32, tile_cfg=24))
    assert rel(f32, a.float() @ w.float().t() + bias) < 1e-5
    sw = K.gemm(a, w, act="swiglu", tile_cfg=34)
    assert sw.shape == (700, 512) and torch.equal(sw, K.gemm(a, w, act="swiglu", tile_cfg=24))
    y = a.float() @ w.float().t()
    assert rel(sw, F.silu(y[:, 0::2]) * y[:, 1::2]) < 3 * TOL[dtype]
    for splits in (2, 3, 6):                                     # K slices: fp32 partials + the reduce launch
        assert rel(K.gemm(a, w, splits=splits, tile_cfg=34), y) < TOL[dtype]
    a2, w2 = rnd(767, 11008, seed=10, dtype=dtype), rnd(512, 11008, seed=11, scale=0.1, dtype=dtype)
    assert rel(K.gemm(a2, w2, splits=4, tile_cfg=34), a2.float() @ w2.float().t()) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,heads", [(1, 767, 32), (3, 300, 4)])
def test_fused_qkv_rope_epilogue_bit_identical_to_the_ring_tile(B, T, heads, dtype):
    """epilogue mode W4_ROPE: q rotated, k rotated into the cache rows pos0 + t, v into the cache; untouched cache rows stay 0"""
    HD = heads * 128
    h, wqkv = rnd(B * T, 512, seed=12, dtype=dtype), rnd(3 * HD, 512, seed=13, dtype=dtype)
    ang = torch.rand(1024, 64, generator=torch.Generator().manual_seed(14)) * 6.28
    cos, sin = ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV)
    outs = {}
    for t in (24, 34):
        q = torch.zeros(B, T, HD, dtype=dtype, device=DEV)
        kc = torch.zeros(B, 1024, HD, dtype=dtype, device=DEV)
        vc = torch.zeros(B, 1024, HD, dtype=dtype, device=DEV)
        assert K.gemm_qkv_rope(h, wqkv, B, T, heads, 128, q, kc, vc, cos, sin, 5, tile_cfg=t) is not None
        outs[t] = (q, kc, vc)
    for x, y in zip(outs[24], outs[34]):
        assert torch.equal(x, y)
    q, kc, vc = outs[34]
    assert (kc[:, :5] == 0).all() and (kc[:, 5 + T:] == 0).all() and (q != 0).float().mean() > 0.9
    # against the plain statement: projection rounded to 16 bit, rotate_half in fp32
    y = (h.float() @ wqkv.float().t()).to(dtype).float().view(B, T, 3, heads, 128)
    c, s = cos[5:5 + T][None, :, None, :], sin[5:5 + T][None, :, None, :]

    def rot(x):
        x1, x2 = x[..., :64], x[..., 64:]
        return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], -1)
    assert rel(q.view(B, T, heads, 128), rot(y[:, :, 0])) < 2 * TOL[dtype]
    assert rel(kc[:, 5:5 + T].view(B, T, heads, 128), rot(y[:, :, 1])) < 2 * TOL[dtype]
    assert rel(vc[:, 5:5 + T].view(B, T, heads, 128), y[:, :, 2]) < TOL[dtype]       # (fp32 sums in another order: last-bit roundings)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,Ci,Co,G", [(1, 48, 48, 128, 256, 1), (2, 24, 20, 64, 320, 1), (1, 14, 14, 128, 256, 4)])
def test_implicit_gemm_conv_vs_fp32_conv2d(B, H, W, Ci, Co, G, dtype):
    """AMODE 1: taps as per-lane offsets, out-of-image taps as offsets beyond the buffer descriptor (zeros), groups summed in
    one accumulator; K walks 64-channel slices, so the ring tile (32-channel slices) differs in the last bit only"""
    x = rnd(G, B, H, W, Ci, seed=15, dtype=dtype) if G > 1 else rnd(B, H, W, Ci, seed=15, dtype=dtype)
    ws = [rnd(Co, Ci, 3, 3, scale=0.05, seed=16 + g, dtype=torch.float32) for g in range(G)]
    wk = K.prep_conv3x3_weight(ws, dtype=dtype)
    bias = rnd(Co, seed=20, dtype=torch.float32)
    xs = x if G > 1 else x[None]
    ref = sum(F.conv2d(xs[g].float().permute(0, 3, 1, 2), ws[g].to(dtype).float(), padding=1) for g in range(G))
    ref = (ref + bias[None, :, None, None]).relu().permute(0, 2, 3, 1)
    got = K.conv3x3(x, wk, bias=bias, act="relu", groups=G, tile_cfg=34)
    assert rel(got, ref) < TOL[dtype]
    assert rel(got, K.conv3x3(x, wk, bias=bias, act="relu", groups=G, tile_cfg=24)) < TOL[dtype]


def test_conv_over_all_levels_one_wave_per_simd_vs_ring_kernel():
    """AMODE 2 (all pyramid levels in one launch): the production kernel against its ring ping-pong arm (debug mode 60)"""
    mm = K.MlvlMaps(2, [(16, 16), (8, 8), (4, 4), (2, 2)], 128, DEV)
    mm.flat.copy_(rnd(*mm.flat.shape, seed=21))
    wc = rnd(256, 128, 3, 3, scale=0.05, seed=22, dtype=torch.float32)
    wk = K.prep_conv3x3_weight(wc)
    got = K.conv3x3_mlvl(mm, wk).flat.clone()
    lib().g4r_gemm_debug_mode(60)
    try:
        ring = K.conv3x3_mlvl(mm, wk).flat.clone()
    finally:
        lib().g4r_gemm_debug_mode(0)
    ref = torch.cat([F.conv2d(m.float().permute(0, 3, 1, 2), wc.to(torch.bfloat16).float(), padding=1).permute(0, 2, 3, 1).reshape(-1, 256)
                     for m in mm.levels])
    assert rel(got, ref) < TOL[torch.bfloat16] and rel(got, ring) < TOL[torch.bfloat16]


# ------------------------------------------------------------------------------------------ the persistent form (round 6)
# Launches of MORE than one wave of 256 x 256 tiles with N % 256 == 0 run gemm_bf16_w4k64p_kernel: one workgroup per CU walks
# the tiles, the next tile's first K tiles are requested before the epilogue, which stores straight from the accumulators
# (transposed through spare LDS, whole-row buffer stores).  Same K order and rounding points: bit-identical to the ring tile 24.
# Production (tile 34) offers it from K = 2048 on; tile_cfg 36 = the same kernel offered at every K (one, two, three, five K tiles
# per output tile: the prologue / relaxed-wait / last-tile paths), which is what these tests ask for.
PT = 36
PERSIST_SHAPES = [(4100, 4096, 192), (4353, 4096, 64), (8200, 2304, 128), (4352, 4352, 320), (4100, 4096, 2048)]      # 272 / 288 / 297 / 289 tiles; ragged last row tiles


def _tiles(M, N):
    return -(-M // 256) * -(-N // 256)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,Kd", PERSIST_SHAPES)
def test_persistent_form_plain_and_bias_activation(M, N, Kd, dtype):
    """W4_P16 behind the next tile's pieces (no bias) and ahead of them (bias + every activation); one, two, three, five K tiles"""
    assert _tiles(M, N) > 256 and N % 256 == 0
    a, w = rnd(M, Kd, seed=60, dtype=dtype), rnd(N, Kd, seed=61, scale=0.2, dtype=dtype)
    got = K.gemm(a, w, tile_cfg=PT)
    assert torch.equal(got, K.gemm(a, w, tile_cfg=24))
    assert rel(got, a.float() @ w.float().t()) < TOL[dtype]
    bias = rnd(N, seed=62, dtype=torch.float32)
    for act in (None, "relu", "quick_gelu", "silu"):
        assert torch.equal(K.gemm(a, w, bias=bias, act=act, tile_cfg=PT), K.gemm(a, w, bias=bias, act=act, tile_cfg=24)), act
    # a strided output view: rows 16-byte aligned, nothing written outside it
    big = torch.zeros(M, N + 64, dtype=dtype, device=DEV)
    K.gemm(a, w, out=big[:, 32:32 + N], tile_cfg=PT)
    assert torch.equal(big[:, 32:32 + N], got) and float(big[:, :32].abs().max()) == 0 and float(big[:, 32 + N:].abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,Kd", PERSIST_SHAPES[:3])
def test_persistent_form_residual_and_swiglu(M, N, Kd, dtype):
    a, w = rnd(M, Kd, seed=63, dtype=dtype), rnd(N, Kd, seed=64, scale=0.2, dtype=dtype)
    bias, res = rnd(N, seed=65, dtype=torch.float32), rnd(M, N, seed=66, dtype=dtype)
    for b_, act in ((None, None), (bias, "silu"), (bias, "quick_gelu")):
        got = K.gemm(a, w, bias=b_, residual=res, act=act, tile_cfg=PT)
        assert torch.equal(got, K.gemm(a, w, bias=b_, residual=res, act=act, tile_cfg=24)), act
    # in place (x += a @ w^T, as o_proj / down_proj update the residual stream)
    x = res.clone()
    K.gemm(a, w, residual=x, out=x, tile_cfg=PT)
    assert torch.equal(x, K.gemm(a, w, residual=res, tile_cfg=24))
    r = a.float() @ w.float().t() + res.float()
    assert rel(x, r) < 2 * TOL[dtype]
    sw = K.gemm(a, w, act="swiglu", tile_cfg=PT)
    assert sw.shape == (M, N // 2) and torch.equal(sw, K.gemm(a, w, act="swiglu", tile_cfg=24))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,T,heads,pos0", [(3, 1950, 8, 7), (5, 767, 12, 0)])
def test_persistent_form_fused_qkv_rope(B, T, heads, pos0, dtype):
    """W4_ROPE in the accumulator layout (the rotation partner d +- 64 is accumulator block j +- 2 of the same lane): q, rotated k
    and v rows against the ring tile's LDS-staged epilogue, several sequences, a last row tile that is mostly empty"""
    HD = heads * 128
    assert _tiles(B * T, 3 * HD) > 256
    h, wqkv = rnd(B * T, 128, seed=67, dtype=dtype), rnd(3 * HD, 128, seed=68, dtype=dtype)
    ang = torch.rand(2048, 64, generator=torch.Generator().manual_seed(69)) * 6.28
    cos, sin = ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV)
    outs = {}
    for t in (24, PT):
        q = torch.zeros(B, T, HD, dtype=dtype, device=DEV)
        kc = torch.zeros(B, 2048, HD, dtype=dtype, device=DEV)
        vc = torch.zeros(B, 2048, HD, dtype=dtype, device=DEV)
        assert K.gemm_qkv_rope(h, wqkv, B, T, heads, 128, q, kc, vc, cos, sin, pos0, tile_cfg=t) is not None
        outs[t] = (q, kc, vc)
    for x, y in zip(outs[24], outs[PT]):
        assert torch.equal(x, y)
    assert (outs[PT][0] != 0).float().mean() > 0.9


def test_persistent_form_is_what_the_merged_llama_shapes_run():
    """the dispatch of the benchmarked step: M = 16 x 767 rows through q|k|v, o_proj, gate|up (main part) and down_proj all
    qualify (more than 256 tiles, N % 256 == 0); the same problems against fp32 torch on sampled rows"""
    M, dtype = 12272, torch.float16
    a = rnd(M, 4096, seed=70, dtype=dtype)
    rows = torch.tensor([0, 255, 256, 6000, 12031, 12032, M - 1], device=DEV)
    for N, Kd, kw in ((4096, 4096, {}), (21760, 4096, {"act": "swiglu"})):
        assert _tiles(M, N) > 256 and N % 256 == 0
        w = rnd(N, Kd, seed=71, scale=0.02, dtype=dtype)
        got = K.gemm(a, w, tile_cfg=34, **kw)
        y = a[rows].float() @ w.float().t()
        ref = F.silu(y[:, 0::2]).to(dtype).float() * y[:, 1::2] if kw else y
        assert rel(got[rows], ref) < 3 * TOL[dtype]
        assert torch.equal(got, K.gemm(a, w, tile_cfg=24, **kw))


def test_operands_of_2_GiB_and_more_fall_back_to_the_ring_tile():
    """tile 34 addresses its operands through buffer descriptors (32-bit extents).  An activation matrix
    of 2 GiB and more -- the stage-2 data gradient dgu[tokens, 22016] x W^T above ~48.7 k tokens -- must run (on the ring
    ping-pong tile with 64-bit global_load_lds pieces), under the default dispatch and when tile 34 is asked for by number;
    sampled rows against fp32 torch.  (The arithmetic nn.Linear's backward delegates to cuBLAS, llava/model/llava.py:52.)"""
    M, N, Kd = 66048, 512, 16384                                # A: 2.16 GB of bf16
    g = torch.Generator(device=DEV).manual_seed(50)
    a = (torch.randn(M, Kd, generator=g, device=DEV) * 0.5).to(torch.bfloat16)
    w = rnd(N, Kd, seed=51, scale=0.05)
    assert a.numel() * 2 >= 2 ** 31
    rows = torch.tensor([0, 1, 255, 256, 32767, 65535, 65536, 65537, M - 257, M - 1], device=DEV)
    ref = a[rows].float() @ w.float().t()
    for tile in (None, 34):
        got = K.gemm(a, w, tile_cfg=tile)
        assert rel(got[rows], ref) < TOL[torch.bfloat16], tile
    assert K.pick_tile(M, N, Kd) == 34                           # (the default dispatch did ask for tile 34)
    del a


# ------------------------------------------------------------------------------------------ batched decode projections (round 5)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,N,Kd", [(2, 1000, 512), (3, 4096, 4096), (8, 12288, 4096), (8, 4096, 11008), (16, 2050, 11008), (5, 32006, 4096),
                                     (16, 4096, 4096)])
def test_gemv_batch_rows_share_one_weight_stream(B, N, Kd, dtype):
    """g4r_gemv_batch (csrc/gemv_mfma.hip): B = 2..16 activation rows, K staged in one pass or several (8 x 11008 and 16 x 4096 do not
    fit the LDS budget), N not a multiple of 16, strided activations; every epilogue against fp32 torch.  What HF generate()
    computes per token for a batch (LlamaRMSNorm + nn.Linear on [B, 1, K])."""
    xw = rnd(B, Kd + 64, seed=30, dtype=dtype)
    x = xw[:, :Kd]                                             # row stride Kd + 64
    w = rnd(N, Kd, seed=31, scale=0.05, dtype=dtype)
    bias = rnd(N, seed=32, dtype=torch.float32)
    res = rnd(B, N, seed=33, dtype=dtype)
    ref = x.float() @ w.float().t()
    assert rel(K.gemv_batch(x, w), ref) < TOL[dtype]
    assert rel(K.gemv_batch(x, w, bias=bias, residual=res, act="silu"), F.silu(ref + bias) + res.float()) < 2 * TOL[dtype]
    f32 = K.gemv_batch(x, w, bias=bias, out_dtype=torch.float32)
    assert f32.dtype == torch.float32 and rel(f32, ref + bias) < 2e-4
    if K.gemv_batch_wins(B, N, Kd):                             # gemm() routes these shapes here
        assert torch.equal(K.gemm(x, w, bias=bias, residual=res), K.gemv_batch(x, w, bias=bias, residual=res))
    if N % 4 == 0:
        sw = K.gemv_batch(x, w, act="swiglu")
        assert sw.shape == (B, N // 2) and rel(sw, F.silu(ref[:, 0::2]) * ref[:, 1::2]) < 3 * TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Kd", [(2, 4096), (4, 4096), (5, 4096), (7, 4096), (8, 2048), (8, 4096), (16, 4096), (3, 8192), (2, 576)])
def test_gemv_batch_fused_rmsnorm_is_the_separate_launch(B, Kd, dtype):
    """the RMSNorm in front of q|k|v / gate|up / lm_head fused into the staging: bit-identical to rmsnorm() followed by the
    un-normed call (same element -> thread map, summation order and roundings as rmsnorm_bf16_kernel), one pass and two; up to 8 rows
    of K <= 4096 take the one-round-trip form (round 6: all rows summed at once, 2 or 4 per 256 threads), the others the row loop"""
    x = rnd(B, Kd, seed=40, dtype=dtype)
    gamma = 1 + 0.1 * rnd(Kd, seed=41, dtype=torch.float32)
    w = rnd(1536, Kd, seed=42, scale=0.05, dtype=dtype)
    fused = K.gemv_batch(x, w, norm_weight=gamma, eps=1e-6)
    assert torch.equal(fused, K.gemv_batch(K.rmsnorm(x, gamma, 1e-6), w))
    h = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6)
    assert rel(fused, (h.to(dtype).float() * gamma).to(dtype).float() @ w.float().t()) < TOL[dtype]
    sw = K.gemv_batch(x, w, norm_weight=gamma, eps=1e-6, act="swiglu")
    assert torch.equal(sw, K.gemv_batch(K.rmsnorm(x, gamma, 1e-6), w, act="swiglu"))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Kd", [(8, 4096), (16, 4096), (8, 11008), (5, 11008)])
def test_batched_decode_reduce_fused_with_the_next_rmsnorm_is_bit_identical(M, Kd, dtype):
    """LlamaDecoder._decode_step_batch on the tiles path (round 6): o_proj / down_proj as K slices whose reduce launch also adds the
    residual and applies the next RMSNorm (gemm_partials + rmsnorm_splitk) against gemm(residual=) + rmsnorm(): every element of
    the residual stream and of the normalised rows, six draws each (7.9 M elements compared on the GPU box: no mismatch)"""
    plan = K.decode_split_plan(M, 4096, Kd)
    assert plan == (14, 4)
    for seed in range(6):
        a = rnd(M, Kd, seed=100 + seed, dtype=dtype)
        w = rnd(4096, Kd, seed=200 + seed, scale=0.03, dtype=dtype)
        x = rnd(M, 4096, seed=300 + seed, scale=1.0, dtype=dtype)
        gam = 1 + 0.1 * rnd(4096, seed=400 + seed, dtype=torch.float32)
        x1 = K.gemm(a, w, residual=x)
        h1 = K.rmsnorm(x1, gam, 1e-6)
        part, ns = K.gemm_partials(a, w, plan[1], plan[0])
        x2, h2 = K.rmsnorm_splitk(part, ns, x, gam, 1e-6)
        assert torch.equal(x1, x2) and torch.equal(h1, h2)
    assert K.decode_split_plan(3, 4096, 4096) is None              # few rows of K <= 8192: the weight-streaming kernel, no K slices

