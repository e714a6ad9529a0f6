"""GPU: every hand-written gfx950 kernel against a plain PyTorch fp32 statement of the same op
(and RoIAlign against the CPU oracle / committed reference fixtures).  All calls go through the
C ABI (ctypes) -- see gpt4roi_amd/kernels.py."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import roi_align as O  # noqa: E402  (the checker; never the product path)

if torch.cuda.is_available():
    from gpt4roi_amd import kernels as K
    from gpt4roi_amd.roi_align import RoIAlign, roi_align

DEV = "cuda"


def rnd(*shape, scale=1.0, seed=None, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed if seed is not None else (hash(shape) & 0xffff))
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def close(got, ref, atol, rtol, what=""):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = (err > tol)
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{bad.numel()} off; max abs err {err.max().item():.3e} "
                           f"at ref {ref.flatten()[err.argmax()].item():.3e}")


# ------------------------------------------------------------------------------------------ RoIAlign
def test_roi_align_known_answers_all_dtypes(golden_dir):
    # mmcv-1.4.7/tests/test_ops/test_roi_align.py:66-104 (atol 1e-3 there, for float/double/half)
    z = np.load(os.path.join(golden_dir, "roi_align_known.npz"))
    for dtype, atol in ((torch.float32, 1e-6), (torch.float64, 1e-12), (torch.float16, 1e-3)):
        for i in range(3):
            x = torch.tensor(z[f"x{i}"], dtype=dtype, device=DEV, requires_grad=True)
            r = torch.tensor(z[f"rois{i}"], dtype=dtype, device=DEV)
            out = roi_align(x, r, (2, 2), 1.0, 2, 'avg', True)
            out.backward(torch.ones_like(out))
            np.testing.assert_allclose(out.detach().float().cpu().numpy(), z[f"out{i}"], atol=atol)
            np.testing.assert_allclose(x.grad.float().cpu().numpy(), z[f"grad{i}"], atol=max(atol, 1e-6))


def test_roi_align_gradcheck_fp64(golden_dir):
    # test_roi_align.py:41-64
    z = np.load(os.path.join(golden_dir, "roi_align_known.npz"))
    for i in range(3):
        x = torch.tensor(z[f"x{i}"], dtype=torch.float64, device=DEV, requires_grad=True)
        r = torch.tensor(z[f"rois{i}"], dtype=torch.float64, device=DEV)
        assert torch.autograd.gradcheck(RoIAlign((2, 2), 1.0, 2), (x, r), eps=1e-5, atol=1e-5)


def test_roi_align_reference_fixtures(golden_dir):
    """fixtures = outputs of the reference's own CPU code (tests/golden/make_golden.py)."""
    z = np.load(os.path.join(golden_dir, "roi_align_seeded.npz"))
    worst = 0.0
    for name in [str(n) for n in z["names"]]:
        ph, pw, sr, avg, aligned = [int(v) for v in z[f"{name}.cfg"]]
        mode = "avg" if avg else "max"
        scale = float(z[f"{name}.scale"])
        x = torch.tensor(z[f"{name}.x"], device=DEV, requires_grad=True)
        r = torch.tensor(z[f"{name}.rois"], device=DEV)
        out = roi_align(x, r, (ph, pw), scale, sr, mode, bool(aligned))
        out.backward(torch.tensor(z[f"{name}.gout"], device=DEV))
        e1 = np.abs(out.detach().cpu().numpy() - z[f"{name}.out"]).max()
        e2 = np.abs(x.grad.cpu().numpy() - z[f"{name}.gin"]).max()
        worst = max(worst, e1, e2)
        assert e1 <= 1e-5, (name, "forward", e1)   # north_star tolerance is 1e-4 fp32
        assert e2 <= 1e-4, (name, "backward", e2)  # atomics reorder the sums
    print("roi_align fixtures worst abs err", worst)


def _gpt4roi_case(P=16, B=2, N=32, C=1024, seed=0):
    g = torch.Generator().manual_seed(seed)
    sizes = [8 * P, 4 * P, 2 * P, P]
    strides = [14 / 8, 14 / 4, 14 / 2, 14]
    feats = [torch.randn(B, C, s, s, generator=g) for s in sizes]
    xy = torch.rand(N, 2, generator=g) * 0.6
    wh = torch.rand(N, 2, generator=g) * 0.3 + 0.05
    idx = torch.randint(0, B, (N, 1), generator=g).float()
    rois = torch.cat([idx, xy * 14 * P, (xy + wh) * 14 * P], 1)
    return feats, rois, strides


def test_roi_align_gpt4roi_regime_vs_oracle():
    """14x14 bins, sr 2, the four pyramid levels at 224^2 (layers.py:206-214) vs the C oracle."""
    from oracle import roi_align as O
    feats, rois, strides = _gpt4roi_case(P=16, B=2, N=32, C=256)
    for f, s in zip(feats, strides):
        want = O.forward(f.numpy(), rois.numpy(), 14, 1.0 / s, 2)[0]
        got = roi_align(f.to(DEV), rois.to(DEV), 14, 1.0 / s, 2, 'avg', True).cpu().numpy()
        assert np.abs(got - want).max() <= 1e-5
        gout = torch.randn(got.shape, generator=torch.Generator().manual_seed(1))
        x = f.to(DEV).requires_grad_(True)
        roi_align(x, rois.to(DEV), 14, 1.0 / s, 2, 'avg', True).backward(gout.to(DEV))
        wantg = O.backward(gout.numpy(), rois.numpy(), f.shape, 14, 1.0 / s, 2)
        assert np.abs(x.grad.cpu().numpy() - wantg).max() <= 1e-4 * max(1.0, np.abs(wantg).max())


def test_roi_align_mlvl_nhwc_matches_dropin_and_gn():
    feats, rois, strides = _gpt4roi_case(P=8, B=2, N=9, C=64, seed=3)
    rois_d = rois.to(DEV)
    nchw = [f.to(DEV) for f in feats]
    nhwc = [f.permute(0, 2, 3, 1).contiguous() for f in nchw]
    out = K.roi_align_mlvl(nhwc, rois_d, 14, [1.0 / s for s in strides])
    for l, s in enumerate(strides):
        ref = roi_align(nchw[l], rois_d, 14, 1.0 / s, 2, 'avg', True).permute(0, 2, 3, 1)
        assert torch.equal(out[l], ref), f"level {l}: max diff {(out[l]-ref).abs().max().item()}"
    # bf16 storage + deferred GroupNorm/ReLU affine
    nb = [f.to(torch.bfloat16) for f in nhwc]
    aff = [torch.randn(2, 2, 64, device=DEV) for _ in nb]
    outb = K.roi_align_mlvl(nb, rois_d, 14, [1.0 / s for s in strides], affines=aff)
    for l, s in enumerate(strides):
        a, sh = aff[l][:, 0], aff[l][:, 1]
        y = torch.relu(nb[l].float() * a[:, None, None, :] + sh[:, None, None, :]).permute(0, 3, 1, 2).contiguous()
        ref = roi_align(y, rois_d, 14, 1.0 / s, 2, 'avg', True).permute(0, 2, 3, 1)
        close(outb[l], ref, 1e-2, 1e-2, f"mlvl gn level {l}")
    empty = K.roi_align_mlvl(nhwc, rois_d[:0], 14, [1.0 / s for s in strides])
    assert empty.shape == (4, 0, 14, 14, 64)


def test_roi_align_mlvl_staged_path_edge_boxes():
    """bf16 kernel (LDS-staged rows for narrow RoIs, direct gather for wide ones) against the fp32
    kernel (reference summation order) on boxes that hang off the map, are degenerate, tiny or full-frame."""
    B, C, P = 2, 256, 8
    sizes = [8 * P, 4 * P, 2 * P, P]
    g = torch.Generator().manual_seed(9)
    feats32 = [torch.randn(B, s, s, C, generator=g).to(DEV) for s in sizes]
    img = 14.0 * P
    rois = torch.tensor([[0, -20., -20., 30., 40.], [1, 0., 0., img, img], [0, 50., 50., 50., 50.],
                         [1, 100., 90., 140., 111.9], [0, img - 3, img - 3, img + 30, img + 30],
                         [1, 200., 200., 240., 250.], [0, 10.3, 20.7, 17.1, 25.2], [1, 5., 60., 110., 64.]], device=DEV)
    scales = [8 / 14, 4 / 14, 2 / 14, 1 / 14]
    feats16 = [f.to(torch.bfloat16) for f in feats32]
    want = K.roi_align_mlvl([f.float() for f in feats16], rois, 14, scales)        # fp32 kernel, exact order
    got = K.roi_align_mlvl(feats16, rois, 14, scales)
    close(got, want, 2e-2, 1e-2, "staged bf16 vs fp32 kernel")
    aff = [torch.randn(B, 2, C, device=DEV) for _ in sizes]
    want_a = K.roi_align_mlvl([torch.relu(f.float() * a[:, 0][:, None, None, :] + a[:, 1][:, None, None, :])
                               for f, a in zip(feats16, aff)], rois, 14, scales)
    close(K.roi_align_mlvl(feats16, rois, 14, scales, affines=aff), want_a, 3e-2, 1e-2, "staged + deferred GN")


# ------------------------------------------------------------------------------------------ GEMM / conv
@pytest.mark.parametrize("tile", [24, 28])
@pytest.mark.parametrize("M,N,K", [(3100, 1300, 256), (6136, 4096, 128), (3265, 520, 192)])
def test_gemm_grouped_tile_order_with_ragged_groups(tile, M, N, K):
    """Dense launches with >= 12 row tiles walk the tiles in groups of 8 row tiles x all column tiles (round 4: the 32
    workgroups an XCD runs at a time then share 8 A tiles + 4 W panels instead of re-reading all of A).  Ragged cases: a last
    group of fewer than 8 row tiles, a last row tile and a last column tile that are partly outside the matrix."""
    a, w = rnd(M, K, seed=11), rnd(N, K, seed=12)
    ref = a.float() @ w.float().t()
    got = K_gemm(a, w, tile_cfg=tile)
    close(got, ref, 0.06 * math.sqrt(K / 64), 1e-2, f"grouped gemm {M}x{N}x{K} tile {tile}")


# every tile the shipped library holds (the superseded forms / A-B arms live in the tools build only: -DG4R_TOOLS_BUILD)
SHIPPED_TILES = [0, 4, 7, 13, 14, 24, 28, 34]


@pytest.mark.parametrize("tile", SHIPPED_TILES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 328, 192), (37, 1024, 1024), (800, 512, 2048),
                                   (50, 30, 64), (300, 256, 128)])
def test_gemm_plain(tile, M, N, K):
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2)
    ref = a.float() @ w.float().t()
    got = K_gemm(a, w, tile_cfg=tile)
    close(got, ref, 0.06 * math.sqrt(K / 64), 1e-2, f"gemm {M}x{N}x{K} tile {tile}")


def K_gemm(*a, **k):
    return K.gemm(*a, **k)


def test_gemm_layout_is_not_transposed():
    # asymmetric operands: A = one-hot rows, W = ramp -> C[m, n] must equal W[n, m-th k]
    M, N, Kd = 96, 160, 128
    a = torch.zeros(M, Kd, dtype=torch.bfloat16, device=DEV)
    a[torch.arange(M), torch.arange(M) % Kd] = 1
    w = (torch.arange(N * Kd, device=DEV).reshape(N, Kd) % 251).to(torch.bfloat16)
    got = K.gemm(a, w, tile_cfg=0, out_dtype=torch.float32)
    ref = w.float()[:, torch.arange(M, device=DEV) % Kd].t()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("act", [None, "relu", "quick_gelu", "silu"])
def test_gemm_epilogue(act):
    M, N, Kd = 300, 256, 512
    a, w, bias, res = rnd(M, Kd, seed=3), rnd(N, Kd, scale=0.1, seed=4), rnd(N, seed=5, dtype=torch.float32), rnd(M, N, seed=6)
    x = a.float() @ w.float().t() + bias
    x = {None: x, "relu": torch.relu(x), "quick_gelu": x * torch.sigmoid(1.702 * x), "silu": F.silu(x)}[act]
    ref = x + res.float()
    close(K.gemm(a, w, bias=bias, residual=res, act=act), ref, 0.05, 1e-2, f"epilogue {act}")
    close(K.gemm(a, w, bias=bias, residual=res, act=act, out_dtype=torch.float32), ref, 0.02, 2e-3, "f32 out")
    close(K.gemm(a, w, bias=bias, residual=res, act=act, splits=4), ref, 0.05, 1e-2, "split-K")
    # the 256x256 kernels finish through LDS (row-major, whole-line stores): same arithmetic, bit-identical results to the
    # direct epilogue of the 128-wide tiles on the same fp32 sums -- checked on ragged shapes too (N not a multiple of 8,
    # strided output / residual, fp32 output, split-K partials)
    for tile in (24, 28, 34):
        close(K.gemm(a, w, bias=bias, residual=res, act=act, tile_cfg=tile), ref, 0.05, 1e-2, f"epilogue {act} tile {tile}")
        close(K.gemm(a, w, bias=bias, residual=res, act=act, out_dtype=torch.float32, tile_cfg=tile), ref, 0.02, 2e-3,
              f"f32 out tile {tile}")
        close(K.gemm(a, w, bias=bias, residual=res, act=act, splits=3, tile_cfg=tile), ref, 0.05, 1e-2, f"split-K tile {tile}")
        Nr = 203                                                    # ragged: runs of 8 are cut, rows are odd-strided
        big = torch.zeros(M, 260, dtype=torch.bfloat16, device=DEV)
        rr = rnd(M, 217, seed=40)[:, :Nr]
        got = K.gemm(a, w[:Nr], bias=bias[:Nr], residual=rr, act=act, out=big[:, 5:5 + Nr], tile_cfg=tile)
        xr = a.float() @ w[:Nr].float().t() + bias[:Nr]
        xr = {None: xr, "relu": torch.relu(xr), "quick_gelu": xr * torch.sigmoid(1.702 * xr), "silu": F.silu(xr)}[act]
        close(got, xr + rr.float(), 0.05, 1e-2, f"ragged N tile {tile}")
        assert float(big[:, :5].abs().max()) == 0 and float(big[:, 5 + Nr:].abs().max()) == 0     # nothing outside the view


def test_gemm_swiglu_epilogue():
    M, Fd, Kd = 300, 704, 512
    a, g, u = rnd(M, Kd, seed=20), rnd(Fd, Kd, scale=0.1, seed=21), rnd(Fd, Kd, scale=0.1, seed=22)
    gate, up = a.float() @ g.float().t(), a.float() @ u.float().t()
    ref = F.silu(gate).to(torch.bfloat16).float() * up
    for tile in (0, 4, 7, 24, 28, 34):
        got = K.gemm(a, K.interleave_gate_up(g, u), act="swiglu", tile_cfg=tile)
        assert got.shape == (M, Fd)
        close(got, ref, 0.03, 2e-2, f"swiglu epilogue tile {tile}")


def test_gemv_single_token_path():
    """M = 1 takes the weight-streaming GEMV kernel (decode step); same epilogues as the GEMM."""
    for N, Kd in ((4096, 4096), (12288, 4096), (4096, 11008), (1000, 512), (32006, 4096)):
        a, w = rnd(1, Kd, seed=23), rnd(N, Kd, scale=0.05, seed=24)
        ref = a.float() @ w.float().t()
        close(K.gemm(a, w), ref, 0.03, 1e-2, f"gemv {N}x{Kd}")
        close(K.gemm(a, w, out_dtype=torch.float32), ref, 2e-3, 1e-3, f"gemv f32 {N}x{Kd}")
    a, w, res, bias = rnd(1, 1024, seed=25), rnd(512, 1024, scale=0.05, seed=26), rnd(1, 512, seed=27), rnd(512, seed=28, dtype=torch.float32)
    close(K.gemm(a, w, bias=bias, residual=res), a.float() @ w.float().t() + bias + res.float(), 0.03, 1e-2, "gemv epilogue")
    g, u = rnd(352, 1024, scale=0.05, seed=29), rnd(352, 1024, scale=0.05, seed=30)
    ref = F.silu(a.float() @ g.float().t()).to(torch.bfloat16).float() * (a.float() @ u.float().t())
    close(K.gemm(a, K.interleave_gate_up(g, u), act="swiglu"), ref, 0.02, 2e-2, "gemv swiglu")


def test_gemm_strided_a_and_small_k():
    big = rnd(64, 3, 128, seed=7)
    a = big[:, 1, :]                      # row stride 384
    w = rnd(96, 128, seed=8)
    close(K.gemm(a, w), a.float() @ w.float().t(), 0.1, 1e-2, "strided A")
    a4, w4, b4 = rnd(33, 4, seed=9), rnd(256, 4, seed=10), rnd(256, seed=11, dtype=torch.float32)
    close(K.gemm(a4, w4, bias=b4, act="relu"), torch.relu(a4.float() @ w4.float().t() + b4), 0.02, 1e-2, "K=4")


def test_gemm_flatten_linear_shape():
    # [N_roi, 200704] x [1024, 200704]^T weight-streaming split-K (layers.py:270, 327)
    M, N, Kd = 32, 1024, 200704
    a, w = rnd(M, Kd, scale=0.05, seed=12), rnd(N, Kd, scale=0.05, seed=13)
    ref = a.float() @ w.float().t()
    got = K.gemm(a, w, splits=49, tile_cfg=4, out_dtype=torch.float32)
    close(got, ref, 0.05, 1e-2, "flatten_linear")


@pytest.mark.parametrize("tile", [0, 4, 7, 14, 24, 28, 34])
def test_conv3x3(tile):
    B, H, W, Cin, Cout = 2, 13, 9, 64, 96
    x = rnd(B, H, W, Cin, seed=14)
    w = rnd(Cout, Cin, 3, 3, scale=0.1, seed=15)
    bias = rnd(Cout, seed=16, dtype=torch.float32)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), bias, padding=1).permute(0, 2, 3, 1)
    got = K.conv3x3(x, K.prep_conv3x3_weight(w), bias=bias, tile_cfg=tile)
    close(got, ref, 0.05, 1e-2, f"conv3x3 tile {tile}")
    close(K.conv3x3(x, K.prep_conv3x3_weight(w), act="relu", tile_cfg=tile),
          torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, padding=1)).permute(0, 2, 3, 1),
          0.05, 1e-2, "conv3x3 relu no bias")


def test_conv3x3_grouped_sum_is_pconv():
    # sum_l pconv_l(roi_feats[l]) (layers.py:321-324) on 14x14 RoI maps
    L, N, C, Co = 4, 5, 64, 128
    x = rnd(L, N, 14, 14, C, seed=17)
    ws = [rnd(Co, C, 3, 3, scale=0.05, seed=18 + l) for l in range(L)]
    bs = [rnd(Co, seed=30 + l, dtype=torch.float32) for l in range(L)]
    ref = sum(F.conv2d(x[l].float().permute(0, 3, 1, 2), ws[l].float(), bs[l], padding=1) for l in range(L))
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    got = K.conv3x3(x, K.prep_conv3x3_weight(ws), bias=sum(bs), act="relu", groups=L)
    close(got, ref, 0.05, 1e-2, "pconv")


def test_conv3x3_full_width():
    B, H, W, C = 1, 24, 24, 1024
    x = rnd(B, H, W, C, seed=40)
    w = rnd(C, C, 3, 3, scale=0.01, seed=41)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), None, padding=1).permute(0, 2, 3, 1)
    close(K.conv3x3(x, K.prep_conv3x3_weight(w)), ref, 0.05, 1e-2, "conv 1024ch")


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, heads, scale, causal):
    B, Tq, HD = q.shape
    Tk, D = k.size(1), HD // heads
    qh = q.float().view(B, Tq, heads, D).transpose(1, 2)
    kh = k.float().view(B, Tk, heads, D).transpose(1, 2)
    vh = v.float().view(B, Tk, heads, D).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2) * scale
    if causal:
        i = torch.arange(Tq, device=q.device)[:, None] + (Tk - Tq)
        j = torch.arange(Tk, device=q.device)[None, :]
        s = s.masked_fill(j > i, float("-inf"))
    return (torch.softmax(s, -1) @ vh).transpose(1, 2).reshape(B, Tq, HD)


@pytest.mark.parametrize("B,H,D,Tq,Tk,causal", [(2, 16, 64, 577, 577, False), (1, 4, 64, 257, 257, False),
                                                 (1, 32, 128, 300, 300, True), (1, 8, 128, 1, 200, True),
                                                 (1, 8, 128, 5, 133, True), (2, 2, 128, 64, 64, True)])
def test_flash_attention(B, H, D, Tq, Tk, causal):
    q, k, v = rnd(B, Tq, H * D, seed=50), rnd(B, Tk, H * D, seed=51), rnd(B, Tk, H * D, seed=52)
    scale = 1.0 / math.sqrt(D)
    got = K.flash_attn(q, k, v, H, scale, causal)
    close(got, _attn_ref(q, k, v, H, scale, causal), 2e-2, 2e-2, f"attn {B,H,D,Tq,Tk,causal}")


@pytest.mark.parametrize("B,H,D,Tq,Tk,causal", [(1, 32, 128, 767, 767, True), (1, 16, 64, 577, 577, False),
                                                 (1, 8, 128, 100, 300, True), (2, 3, 128, 129, 129, True),
                                                 (1, 4, 64, 32, 1000, False), (3, 16, 64, 577, 577, False)])
def test_flash_attention_second_form_every_variant_output_and_lse(B, H, D, Tq, Tk, causal):
    """csrc/attention_v2.hip (key groups inside the workgroup, K/V by LDS-DMA, V through the transpose read) at the
    shapes of the path -- LLaMA prefill T = 767, ViT S = 577, a cached prefill with an offset diagonal -- for every
    instantiated (waves per group, groups) form, against the fp32 statement: output and the log2-domain LSE the
    backward consumes.  Head 0 has peaked rows (the rescale path)."""
    from gpt4roi_amd import _lib
    lib = _lib.lib()
    q, k, v = rnd(B, Tq, H * D, seed=60), rnd(B, Tk, H * D, seed=61), rnd(B, Tk, H * D, seed=62)
    q[:, :, :D] *= 6.0
    scale = 1.0 / math.sqrt(D)
    want = _attn_ref(q, k, v, H, scale, causal)
    qh = q.float().view(B, Tq, H, D).transpose(1, 2)
    kh = k.float().view(B, Tk, H, D).transpose(1, 2)
    sc = qh @ kh.transpose(-1, -2) * scale
    if causal:
        i = torch.arange(Tq, device=q.device)[:, None] + (Tk - Tq)
        sc = sc.masked_fill(torch.arange(Tk, device=q.device)[None, :] > i, float("-inf"))
    want_lse = torch.logsumexp(sc, -1) * 1.4426950408889634
    try:
        for var in ([42, 142, 41] if D == 128 else [24, 124, 42, 142, 41]):
            lib.g4r_attn_debug_variant(var)
            lse = torch.full((B, H, Tq), float("nan"), dtype=torch.float32, device=DEV)
            got = K.flash_attn(q, k, v, H, scale, causal, lse=lse)
            close(got, want, 2e-2, 2e-2, f"attention_v2 variant {var} {B,H,D,Tq,Tk,causal}")
            assert float((lse - want_lse).abs().max()) < 2e-2, (var, float((lse - want_lse).abs().max()))
    finally:
        lib.g4r_attn_debug_variant(0)


def test_flash_attention_strided_qkv_and_peaked_rows():
    # fused qkv buffer [B, T, 3*H*D] (ViT) and one dominant key per row (rescale path)
    B, T, H, D = 1, 130, 4, 64
    qkv = rnd(B, T, 3 * H * D, seed=53)
    qkv[:, :, :H * D] *= 8.0
    q, k, v = qkv[:, :, :H * D], qkv[:, :, H * D:2 * H * D], qkv[:, :, 2 * H * D:]
    got = K.flash_attn(q, k, v, H, 0.125, False)
    close(got, _attn_ref(q, k, v, H, 0.125, False), 3e-2, 3e-2, "attn strided")


def _full_size_rois(N, g):
    xy = torch.rand(N, 2, generator=g) * 0.6
    wh = torch.rand(N, 2, generator=g) * 0.3 + 0.05
    return torch.cat([torch.zeros(N, 1), torch.cat([xy, xy + wh], 1) * 336.0], 1)


def test_roi_align_full_size_vs_the_reference_cpu_op():
    """BASELINE configs[1] sizes -- 4 levels 192/96/48/24 x 1024 channels, 32 RoIs, 14x14 bins, sampling_ratio 2, aligned:
    the shape bench.py times -- against the reference's OWN compiled CPU op (oracle/_ref = mmcv cpu/roi_align.cpp built
    unmodified; falls back to the C restatement, which is pinned bit-exact to it, when _ref is absent): the fp32
    instantiation of the fused multi-level NHWC kernel to <= 1e-5 (north_star: 1e-4), the production bf16 kernel
    (LDS-staged narrow-RoI path included) to bf16 rounding, and the backward of the largest level to <= 1e-4."""
    C, N, sizes = 1024, 32, [192, 96, 48, 24]
    scales = [1 / 1.75, 1 / 3.5, 1 / 7.0, 1 / 14.0]
    g = torch.Generator().manual_seed(310)
    rois = _full_size_rois(N, g)
    x = [torch.randn(1, n, n, C, generator=g).to(torch.bfloat16).float() for n in sizes]   # bf16-representable maps
    fwd = O.ref_forward if O.load_ref() is not None else O.forward
    want = []
    for a, sc in zip(x, scales):
        nchw = a.permute(0, 3, 1, 2).contiguous().numpy()
        out = fwd(nchw, rois.numpy(), 14, np.float32(sc), 2, "avg", True)[0]              # [N, C, 14, 14]
        want.append(torch.from_numpy(out).permute(0, 2, 3, 1))
    want = torch.stack(want)                                                               # [L, N, 14, 14, C]
    got32 = K.roi_align_mlvl([a.to(DEV) for a in x], rois.to(DEV), 14, scales).cpu()
    e32 = float((got32 - want).abs().max())
    gotbf = K.roi_align_mlvl([a.to(DEV).to(torch.bfloat16) for a in x], rois.to(DEV), 14, scales).float().cpu()
    ebf = (gotbf - want).abs()
    tol = 2 ** -8 * want.abs() + 1e-5                        # one rounding to bf16 of the fp32 result (layers.py:311-313)
    print(f"full-size mlvl RoIAlign vs {'oracle/_ref (reference CPU op)' if fwd is O.ref_forward else 'C oracle'}: "
          f"fp32 max |err| {e32:.3e}; bf16 max |err| {float(ebf.max()):.3e}, worst err/tol {float((ebf / tol).max()):.3f}")
    assert got32.shape == want.shape == (4, N, 14, 14, C)
    assert e32 <= 1e-5
    assert bool((ebf <= tol).all())
    # backward at the finest level (the largest map) against the reference's backward
    bwd = O.ref_backward if O.load_ref() is not None else O.backward
    w = torch.randn(4, N, 14, 14, C, generator=g).to(torch.bfloat16)
    gw = bwd(w[0].float().permute(0, 3, 1, 2).contiguous().numpy(), rois.numpy(), (1, C, 192, 192), 14, np.float32(scales[0]), 2)
    gw = torch.from_numpy(np.asarray(gw)).permute(0, 2, 3, 1)
    wd = w.to(DEV)
    for atomic in (False, True):
        grads = [torch.zeros(1, n, n, C, device=DEV) if atomic else torch.full((1, n, n, C), float("nan"), device=DEV)
                 for n in sizes]
        K.roi_align_mlvl_bwd(wd, wd.stride(0), wd.stride(3), grads, rois.to(DEV), 14, scales, 2, True, atomic=atomic)
        eb = float((grads[0].cpu() - gw).abs().max())
        print(f"  backward (atomic={atomic}) level 0 vs the reference backward: max |err| {eb:.3e} (max |grad| {float(gw.abs().max()):.2f})")
        assert eb <= 1e-4 * max(1.0, float(gw.abs().max()))


def test_roi_align_full_size_properties():
    """Same sizes, size-independent properties on top of the reference comparison above: a constant map pools to the
    constant, the op is linear, and the backward kernel is its exact transpose (<f(x), w> = <x, f^T(w)>)."""
    C, N, sizes = 1024, 32, [192, 96, 48, 24]
    scales = [1 / 1.75, 1 / 3.5, 1 / 7.0, 1 / 14.0]
    g = torch.Generator().manual_seed(300)
    rois = _full_size_rois(N, g).to(DEV)
    x = [torch.randn(1, n, n, C, generator=g).to(DEV) for n in sizes]
    y = [torch.randn(1, n, n, C, generator=g).to(DEV) for n in sizes]
    fx = K.roi_align_mlvl(x, rois, 14, scales)
    fy = K.roi_align_mlvl(y, rois, 14, scales)
    fz = K.roi_align_mlvl([2.0 * a + 3.0 * b for a, b in zip(x, y)], rois, 14, scales)
    assert fx.shape == (4, N, 14, 14, C)
    assert float((fz - (2.0 * fx + 3.0 * fy)).abs().max()) < 2e-4
    ones = K.roi_align_mlvl([torch.full_like(a, 1.5) for a in x], rois, 14, scales)
    assert float((ones - 1.5).abs().max()) < 1e-5                     # every sample of these boxes is inside the map
    w = torch.randn(fx.shape, generator=g).to(torch.bfloat16).to(DEV)    # [L, N, 14, 14, C]
    lhs = float((fx.double() * w.double()).sum())
    for atomic in (False, True):       # the atomic-free gather (default; writes every texel: NaN-poisoned start) and the scatter
        grads = [torch.zeros_like(a) if atomic else torch.full_like(a, float("nan")) for a in x]
        K.roi_align_mlvl_bwd(w, w.stride(0), w.stride(3), grads, rois, 14, scales, 2, True, atomic=atomic)
        rhs = float(sum((a.double() * gr.double()).sum() for a, gr in zip(x, grads)))
        assert abs(lhs - rhs) < 2e-6 * float(fx.double().norm() * w.double().norm()), (atomic, lhs, rhs)
    # and the bf16 production kernel agrees with the fp32 instantiation to bf16 rounding
    fb = K.roi_align_mlvl([a.to(torch.bfloat16) for a in x], rois, 14, scales)
    fr = K.roi_align_mlvl([a.to(torch.bfloat16).float() for a in x], rois, 14, scales)
    close(fb, fr, 2e-2, 1e-2, "bf16 vs fp32 mlvl roi_align at full size")


def test_gemm_wave_split_path():
    """Column split into whole waves of 256x256 tiles + a tail GEMM (kernels.wave_split), with every epilogue the
    split has to slice: bias, residual, swiglu's interleaved columns, fp32 output."""
    M, K_ = 700, 2048
    N = 256 * 130                                            # 3 x 130 = 390 tiles: one wave of 255 + a tail
    assert K.wave_split(M, N, K_) == 85 * 256
    a, w = rnd(M, K_, scale=0.5, seed=200), rnd(N, K_, scale=0.05, seed=201)
    bias = rnd(N, seed=202, dtype=torch.float32)
    ref = a.float() @ w.float().t() + bias
    close(K.gemm(a, w, bias=bias, out_dtype=torch.float32), ref, 2e-2, 2e-2, "wave split fp32")
    res = rnd(M, N, seed=203)
    close(K.gemm(a, w, bias=bias, residual=res), ref + res.float(), 6e-2, 2e-2, "wave split residual")
    g = a.float() @ w.float()[0::2].t()
    u = a.float() @ w.float()[1::2].t()
    close(K.gemm(a, w, act="swiglu"), F.silu(g) * u, 6e-2, 3e-2, "wave split swiglu")
    # a narrow tail (the LLaMA gate|up case: 256 of 22016 columns) must not pick split-K under the swiglu epilogue
    n2 = 256 * 86
    assert K.wave_split(M, n2, K_) == 85 * 256
    close(K.gemm(a, w[:n2], act="swiglu"), (F.silu(g) * u)[:, :n2 // 2], 6e-2, 3e-2, "wave split swiglu, narrow tail")


def test_image_preprocess_matches_normalise_then_interpolate():
    # app.py:125-136: processor (rescale 1/255, normalise) then F.interpolate(bilinear, align_corners=False)
    g = torch.Generator().manual_seed(77)
    for (H, W, S) in [(480, 640, 336), (100, 37, 224), (224, 224, 224), (30, 50, 112)]:
        img = torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8)
        mean, std = torch.tensor(K.CLIP_MEAN), torch.tensor(K.CLIP_STD)
        ref = ((img.permute(2, 0, 1).float() / 255.0) - mean[:, None, None]) / std[:, None, None]
        ref = F.interpolate(ref[None], size=(S, S), mode="bilinear", align_corners=False)[0].to(DEV)
        got = K.image_preprocess(img.to(DEV), S)
        close(got, ref, 2e-5, 2e-5, f"image preprocess {H}x{W}->{S}")
        got_bgr = K.image_preprocess(img.flip(2).contiguous().to(DEV), S, bgr=True)
        close(got_bgr, ref, 2e-5, 2e-5, "image preprocess bgr")


# ------------------------------------------------------------------------------------------ norms
def test_layernorm_rmsnorm():
    x = rnd(77, 1024, scale=3.0, seed=60)
    g, b = rnd(1024, seed=61, dtype=torch.float32), rnd(1024, seed=62, dtype=torch.float32)
    close(K.layernorm(x, g, b, 1e-5), F.layer_norm(x.float(), (1024,), g, b, 1e-5), 2e-2, 1e-2, "layernorm")
    xf = x.float()
    ref = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(torch.bfloat16).float() * g
    close(K.rmsnorm(x, g, 1e-6), ref, 2e-2, 1e-2, "rmsnorm")
    x2 = rnd(5, 256, seed=63)
    close(K.layernorm(x2, g[:256].contiguous(), b[:256].contiguous()),
          F.layer_norm(x2.float(), (256,), g[:256], b[:256]), 2e-2, 1e-2, "layernorm 256")


def test_groupnorm_affine():
    B, H, W, C, G = 2, 12, 10, 1024, 64
    x = rnd(B, H, W, C, scale=2.0, seed=64) + 0.5
    g, b = rnd(C, seed=65, dtype=torch.float32), rnd(C, seed=66, dtype=torch.float32)
    ss = K.groupnorm_affine(x, g, b, G, 1e-5)
    got = x.float() * ss[:, 0][:, None, None, :] + ss[:, 1][:, None, None, :]
    ref = F.group_norm(x.float().permute(0, 3, 1, 2), G, g, b, 1e-5).permute(0, 2, 3, 1)
    close(got, ref, 1e-3, 1e-3, "groupnorm affine")


# ------------------------------------------------------------------------------------------ SPI glue
def test_upsample_coord():
    B, Pn, C, H = 2, 6, 64, 24
    hs = rnd(B, Pn * Pn + 1, C, seed=70)
    tok = hs[:, 1:]                                         # drop CLS: strided view
    out = K.upsample_coord(tok, Pn, Pn, H, H, 128)
    feat = tok.float().reshape(B, Pn, Pn, C).permute(0, 3, 1, 2)
    up = F.interpolate(feat, size=(H, H), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    close(out[..., :C], up, 1e-2, 1e-2, "upsample")
    xr = torch.linspace(-1, 1, H, device=DEV)
    close(out[..., C], xr[None, None, :].expand(B, H, H), 4e-3, 0, "x coord")
    close(out[..., C + 1], xr[None, :, None].expand(B, H, H), 4e-3, 0, "y coord")
    assert (out[..., C + 2:] == 0).all()


def _shuffle_ref(own, top, down, affs):
    def fin(x, a):
        x = x.float()
        if a is not None:
            x = torch.relu(x * a[:, 0][:, None, None, :] + a[:, 1][:, None, None, :])
        return x.permute(0, 3, 1, 2)
    o, t, d = fin(own, affs[0]), fin(top, affs[1]), fin(down, affs[2])
    C = o.size(1)
    R, S = C // 2, C // 4
    size = o.shape[-2:]
    ft = F.interpolate(t[:, R:][:, S:], size=size, mode="bilinear", align_corners=True)
    fd = F.interpolate(d[:, R:][:, :S], size=size, mode="bilinear", align_corners=True)
    return torch.cat([o[:, :R], ft, fd], 1).permute(0, 2, 3, 1)


def test_fuse_shuffle():
    B, C = 2, 64
    own, top, down = rnd(B, 16, 16, C, seed=71), rnd(B, 8, 8, C, seed=72), rnd(B, 32, 32, C, seed=73)
    close(K.fuse_shuffle(own, top, down), _shuffle_ref(own, top, down, (None, None, None)), 1e-2, 1e-2, "shuffle")
    affs = tuple(torch.randn(B, 2, C, device=DEV) for _ in range(3))
    close(K.fuse_shuffle(own, top, down, *affs), _shuffle_ref(own, top, down, affs), 2e-2, 1e-2, "shuffle+gn")
    # ends of the pyramid: the level is its own neighbour (layers.py:108-112)
    close(K.fuse_shuffle(own, own, down, affs[0], affs[0], affs[2]),
          _shuffle_ref(own, own, down, (affs[0], affs[0], affs[2])), 2e-2, 1e-2, "shuffle top end")


def test_vit_front_end():
    B, S = 2, 56
    img = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(74)).to(DEV)
    w = rnd(128, 3, 14, 14, scale=0.05, seed=75)
    cols = K.im2col_patch14(img, 640)
    wp = torch.zeros(128, 640, dtype=torch.bfloat16, device=DEV)
    wp[:, :588] = w.reshape(128, 588)
    patch = K.gemm(cols, wp)
    ref = F.conv2d(img.to(torch.bfloat16).float(), w.float(), stride=14).flatten(2).transpose(1, 2)
    close(patch.view(B, 16, 128), ref, 3e-2, 1e-2, "patch embed")
    cls, pos = rnd(128, seed=76), rnd(17, 128, seed=77)
    tok = K.vit_assemble(patch, cls, pos, B)
    reft = torch.cat([cls.float().expand(B, 1, 128), patch.float().view(B, 16, 128)], 1) + pos.float()
    close(tok, reft, 2e-2, 1e-2, "vit assemble")


def test_rope_swiglu_argmax():
    T, Hh, D, maxT, pos0 = 9, 4, 128, 64, 5
    qkv = rnd(T, 3 * Hh * D, seed=80)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.arange(maxT).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().to(DEV), ang.sin().to(DEV)
    qo = torch.zeros(T, Hh * D, dtype=torch.bfloat16, device=DEV)
    kc = torch.zeros(maxT, Hh * D, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    K.rope_qkv(qkv, cos, sin, qo, kc, vc, Hh, D, pos0)

    def rot(x, p0):
        x = x.float().view(-1, Hh, D)
        c = torch.cat([cos, cos], -1)[p0:p0 + x.size(0)][:, None, :]
        s = torch.cat([sin, sin], -1)[p0:p0 + x.size(0)][:, None, :]
        xr = torch.cat([-x[..., D // 2:], x[..., :D // 2]], -1)
        return (x * c + xr * s).reshape(-1, Hh * D)
    close(qo, rot(qkv[:, :Hh * D], pos0), 2e-2, 1e-2, "rope q")
    close(kc[pos0:pos0 + T], rot(qkv[:, Hh * D:2 * Hh * D], pos0), 2e-2, 1e-2, "rope k")
    assert torch.equal(vc[pos0:pos0 + T], qkv[:, 2 * Hh * D:]) and (kc[:pos0] == 0).all()
    gu = rnd(7, 2 * 352, seed=81)
    close(K.swiglu(gu), F.silu(gu[:, :352].float()) * gu[:, 352:].float(), 2e-2, 1e-2, "swiglu")
    lg = torch.randn(3, 32006, device=DEV)
    lg[1, 77] = lg[1, 31000] = 50.0
    assert torch.equal(K.argmax_rows(lg), lg.argmax(-1)) and K.argmax_rows(lg)[1].item() == 77


def test_splice_embed():
    B, T, C, V, NP = 2, 40, 64, 100, 9
    PATCH, BBOX, IMS, IME = 90, 91, 92, 93
    ids = torch.randint(3, 80, (B, T), generator=torch.Generator().manual_seed(82))
    ids[:, 4] = IMS
    ids[:, 5:5 + NP] = PATCH
    ids[:, 5 + NP] = IME
    ids[0, [20, 25, 30]] = BBOX
    ids[1, [22]] = BBOX
    ids = ids.to(DEV)
    embed, img, spi = rnd(V, C, seed=83), rnd(B, NP, C, seed=84), rnd(4, C, seed=85)
    off = torch.tensor([0, 3, 4], dtype=torch.int32, device=DEV)
    out, st = K.splice_embed(ids, embed, img, spi, off, NP, PATCH, BBOX, IMS, IME)
    ref = embed[ids]
    ref[:, 5:5 + NP] = img
    ref[0, [20, 25, 30]] = spi[0:3]
    ref[1, 22] = spi[3]
    assert torch.equal(out, ref) and (st == 0).all()
    bad = ids.clone()
    bad[1, 30] = BBOX                       # one <bbox> too many
    bad[0, 5 + NP] = 7                      # missing <im_end>
    _, st = K.splice_embed(bad, embed, img, spi, off, NP, PATCH, BBOX, IMS, IME)
    assert st[1].item() & 2 and st[0].item() & 8


# ------------------------------------------------------------------------------------------ decode step (row a17)
@pytest.mark.parametrize("N,Kd,act,res,f32", [(12288, 4096, None, False, False), (4096, 4096, None, True, False),
                                             (22016, 4096, "swiglu", False, False), (4096, 11008, None, True, False),
                                             (32006, 4096, None, False, True), (1000, 512, None, False, False),
                                             (36, 1088, "swiglu", False, False), (2048, 8192, None, True, False),
                                             (76, 5184, "swiglu", False, False)])
def test_gemv_with_fused_rmsnorm_is_bit_identical_to_the_two_launches(N, Kd, act, res, f32):
    """g4r_gemv_rmsnorm_bf16 (the decode-step projections): the fused RMSNorm reproduces g4r_rmsnorm_bf16 bit for bit, so
    fused == rmsnorm() -> gemm(M = 1); both within bf16 tolerance of an fp32 statement.  LLaMA-7B shapes + ragged ones."""
    x = rnd(1, Kd, seed=1)
    w = rnd(N, Kd, scale=Kd ** -0.5, seed=2)
    gamma = (1 + 0.1 * torch.randn(Kd, generator=torch.Generator().manual_seed(3))).to(DEV)
    r = rnd(1, N, seed=4) if res else None
    od = torch.float32 if f32 else torch.bfloat16
    h = K.rmsnorm(x, gamma, 1e-6) if Kd <= 8192 else x           # (down_proj has no norm in front of it)
    two = K.gemm(h, w, residual=r, act=act, out_dtype=od)
    plain = K.gemv(h, w, residual=r, act=act, out_dtype=od)              # no norm: x staged as it is
    assert plain.shape == two.shape and torch.equal(plain, two)
    one = K.gemv(x, w, norm_weight=gamma, eps=1e-6, residual=r, act=act, out_dtype=od) if Kd <= 8192 else plain
    assert torch.equal(one, two)
    ref = h.float() @ w.float().t()
    if act == "swiglu":
        g, u = ref[:, 0::2], ref[:, 1::2]
        ref = (F.silu(g).to(torch.bfloat16).float() * u)
    if res:
        ref = ref + r.float()
    close(one, ref, 2e-2, 2 ** -6, f"gemv {N}x{Kd}")


@pytest.mark.parametrize("H,D,kv,splits", [(32, 128, 768, 8), (32, 128, 1, 8), (32, 128, 5, 8), (32, 128, 100, 13),
                                           (32, 128, 2047, 8), (32, 128, 333, 1), (16, 64, 577, 8), (4, 64, 40, 3)])
def test_attn_decode_split_keys(H, D, kv, splits):
    """g4r_attn_decode_bf16: one query over `kv` cached rows, keys split over `splits` workgroups that merge in-launch.
    Against fp32 softmax(q K^T) V and against the tiled kernel (Tq = 1); repeated calls re-arm the arrival counters;
    the device-side length (graph replay) gives the same bits as the host-side one."""
    T_max = 2048
    q = rnd(H * D, seed=5)
    kc = rnd(T_max, H * D, seed=6)
    vc = rnd(T_max, H * D, seed=7)
    scale = D ** -0.5
    work = K.DecodeAttnWorkspace(H, D, DEV, splits=splits)
    out = K.attn_decode(q, kc, vc, H, scale, work, kv_len=kv)
    qh, kh, vh = q.float().view(H, 1, D), kc[:kv].float().view(kv, H, D).transpose(0, 1), vc[:kv].float().view(kv, H, D).transpose(0, 1)
    ref = (torch.softmax(qh @ kh.transpose(1, 2) * scale, -1) @ vh).reshape(-1)
    close(out, ref, 2e-3, 2 ** -7, "attn_decode vs fp32")
    tiled = K.flash_attn(q.view(1, 1, -1), kc[None, :kv], vc[None, :kv], H, scale, True).view(-1)
    close(out, tiled, 4e-3, 2 ** -6, "attn_decode vs tiled kernel")
    assert int(work.cnt.abs().sum()) == 0
    pos = torch.tensor([kv - 1], dtype=torch.int32, device=DEV)
    for _ in range(3):
        again = K.attn_decode(q, kc, vc, H, scale, work, kv_len_dev=pos)
        assert torch.equal(again, out)
    # a strided cache view (one batch slot of [L, B, T, C]) and a peaked row
    big = torch.zeros(2, T_max, H * D, dtype=torch.bfloat16, device=DEV)
    big[1].copy_(kc)
    kq = big[1]
    kq[min(3, kv - 1)] = (q.float() * 4).to(torch.bfloat16)
    out2 = K.attn_decode(q, kq, vc, H, scale, work, kv_len=kv)
    kh2 = kq[:kv].float().view(kv, H, D).transpose(0, 1)
    ref2 = (torch.softmax(qh @ kh2.transpose(1, 2) * scale, -1) @ vh).reshape(-1)
    close(out2, ref2, 2e-3, 2 ** -7, "attn_decode peaked")


@pytest.mark.parametrize("H,D,pos", [(32, 128, 767), (32, 128, 0), (32, 128, 17), (16, 64, 300)])
def test_attn_decode_with_fused_rope_and_cache_append(H, D, pos):
    """qkv= form of g4r_attn_decode_bf16: RoPE of q and k at row `pos`, cache append and attention in one launch ==
    g4r_rope_qkv_bf16 followed by the q= form, bit for bit (output and the appended cache rows)."""
    T_max, C = 1024, H * D
    qkv = rnd(1, 3 * C, seed=11)
    kc, vc = rnd(T_max, C, seed=12), rnd(T_max, C, seed=13)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.arange(T_max).float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV)
    scale = D ** -0.5
    k1, v1, k2, v2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    q = torch.empty(1, C, dtype=torch.bfloat16, device=DEV)
    K.rope_qkv(qkv, cos, sin, q, k1, v1, H, D, pos)
    work = K.DecodeAttnWorkspace(H, D, DEV)
    want = K.attn_decode(q.view(-1), k1, v1, H, scale, work, kv_len=pos + 1)
    pos_dev = torch.tensor([pos], dtype=torch.int32, device=DEV)
    got = K.attn_decode(None, k2, v2, H, scale, work, kv_len_dev=pos_dev, qkv=qkv.view(-1), cos=cos, sin=sin)
    assert torch.equal(got, want)
    assert torch.equal(k2, k1) and torch.equal(v2, v1)               # row `pos` appended, nothing else touched
    assert not torch.equal(k2[pos], kc[pos])


@pytest.mark.parametrize("H,D,kv,splits", [(32, 128, 800, 8), (32, 128, 3, 8), (32, 128, 2047, 4), (16, 64, 577, 5)])
def test_attn_partials_merged_by_the_o_proj_gemv(H, D, kv, splits):
    """defer_merge: the attention leaves its per-split partials in the workspace and g4r_gemv_attn_merge_bf16 assembles
    the attention output while staging its input -- bit-identical to the in-launch merge followed by the plain GEMV."""
    C = H * D
    q, kc, vc = rnd(C, seed=21), rnd(2048, C, seed=22), rnd(2048, C, seed=23)
    wo, res = rnd(C, C, scale=C ** -0.5, seed=24), rnd(1, C, seed=25)
    scale = D ** -0.5
    work = K.DecodeAttnWorkspace(H, D, DEV, splits=splits)
    a = K.attn_decode(q, kc, vc, H, scale, work, kv_len=kv)
    want = K.gemv(a, wo, residual=res)
    assert K.attn_decode(q, kc, vc, H, scale, work, kv_len=kv, defer_merge=True) is None
    got = K.gemv_attn_merge(work, H, D, wo, residual=res)
    assert torch.equal(got, want)
    close(got, a.float()[None] @ wo.float().t() + res.float(), 2e-2, 2 ** -6, "o_proj of merged partials")


@pytest.mark.parametrize("N", [32006, 1000, 7, 4096])
def test_greedy_advance_is_argmax_with_lowest_index_ties(N):
    """g4r_greedy_advance (device-side token selection of generate(do_sample=False)): argmax with torch's tie rule,
    output slot / step / position counters advanced on the device."""
    g = torch.Generator().manual_seed(N)
    lg = torch.randn(N, generator=g).to(DEV)
    hi = float(lg.max()) + 1.0
    for plant in ([], [N - 1], [N // 2, N // 2 + 1, N - 1], [0, N - 1]):
        row = lg.clone()
        for i in plant:
            row[i] = hi
        tok = torch.zeros((1, 1), dtype=torch.int64, device=DEV)
        out = torch.full((8,), -1, dtype=torch.int64, device=DEV)
        step = torch.tensor([2], dtype=torch.int32, device=DEV)
        pos = torch.tensor([40], dtype=torch.int32, device=DEV)
        K.greedy_advance(row, tok, out, step, pos)
        want = int(row.argmax())
        assert int(tok) == want and out.tolist() == [-1, -1, want, -1, -1, -1, -1, -1]
        assert int(step) == 3 and int(pos) == 41


@pytest.mark.parametrize("B,C,sizes", [(2, 128, [(20, 20), (10, 10), (5, 5), (3, 2)]), (1, 64, [(17, 9)]),
                                       (1, 1024, [(192, 192), (96, 96), (48, 48), (24, 24)]), (3, 64, [(8, 8), (4, 4)])])
def test_conv3x3_over_all_pyramid_levels_in_one_launch(B, C, sizes):
    """g4r_conv3x3_mlvl_nhwc_bf16: the levels' maps stacked in one buffer, one implicit GEMM; rows of a 256-row tile may
    belong to different levels (and images).  Bit-identical to one g4r_conv3x3_nhwc_bf16 launch per level on the same
    tile, and within bf16 tolerance of F.conv2d."""
    w4 = rnd(C, C, 3, 3, scale=(9 * C) ** -0.5, seed=31)
    w = K.prep_conv3x3_weight(w4)
    x = K.MlvlMaps(B, sizes, C, DEV)
    x.flat.copy_(rnd(x.flat.size(0), C, seed=32))
    y = K.conv3x3_mlvl(x, w)
    assert [tuple(m.shape) for m in y.levels] == [(B, h, ww, C) for h, ww in sizes]
    for xl, yl in zip(x.levels, y.levels):
        one = K.conv3x3(xl, w, tile_cfg=K.BIG_TILE, splits=1)       # the same kernel and K order (64-channel slices of a tap)
        assert torch.equal(yl, one)
        if C <= 128:
            ref = F.conv2d(xl.float().permute(0, 3, 1, 2), w4.float(), padding=1).permute(0, 2, 3, 1)
            close(yl, ref, 2e-2, 2 ** -6, "conv3x3_mlvl")
    # with bias + ReLU
    bias = rnd(C, seed=33, dtype=torch.float32)
    y2 = K.conv3x3_mlvl(x, w, bias=bias, act="relu")
    for xl, yl in zip(x.levels, y2.levels):
        assert torch.equal(yl, K.conv3x3(xl, w, bias=bias, act="relu", tile_cfg=K.BIG_TILE, splits=1))


@pytest.mark.parametrize("B,H,D,kv", [(3, 32, 128, 300), (8, 32, 128, 21), (2, 16, 64, 577)])
def test_attn_decode_batch_of_sequences_in_one_launch(B, H, D, kv):
    """batch > 1: grid z = the sequence; each has its own qkv row, cache slot, output row, partials and counters.
    Bit-identical to one launch per sequence; the appended cache rows too."""
    T_max, C = 1024, H * D
    qkv = rnd(B, 3 * C, seed=41)
    kbig, vbig = rnd(B, 2, T_max, C, seed=42), rnd(B, 2, T_max, C, seed=43)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.arange(T_max).float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous().to(DEV), ang.sin().contiguous().to(DEV)
    scale = D ** -0.5
    k1, v1, k2, v2 = (t.clone()[:, 1] for t in (kbig, vbig, kbig, vbig))          # batch stride 2 * T_max * C
    w1 = K.DecodeAttnWorkspace(H, D, DEV, splits=8)
    want = torch.stack([K.attn_decode(None, k1[b], v1[b], H, scale, w1, kv_len=kv, qkv=qkv[b], cos=cos, sin=sin)
                        for b in range(B)])
    wB = K.DecodeAttnWorkspace(H, D, DEV, splits=8, batch=B)
    pos = torch.tensor([kv - 1], dtype=torch.int32, device=DEV)
    got = K.attn_decode(None, k2, v2, H, scale, wB, kv_len_dev=pos, qkv=qkv, cos=cos, sin=sin)
    assert got.shape == (B, C) and torch.equal(got, want)
    assert torch.equal(k2, k1) and torch.equal(v2, v1) and int(wB.cnt.abs().sum()) == 0


def test_merged_level_shuffle_and_groupnorm_equal_the_per_level_launches():
    """Round 3: the fuse round's per-level launches merged (g4r_fuse_shuffle_mlvl_nhwc_bf16, g4r_groupnorm_affine_mlvl_nhwc_bf16)
    must be BIT-identical to the per-level kernels they replace (gpt4roi/models/layers.py:152-195): a batch of two images,
    four levels, with and without the deferred GroupNorm affines."""
    B, C, sizes = 2, 256, [(24, 24), (12, 12), (6, 6), (3, 3)]
    g = torch.Generator().manual_seed(77)
    maps = [torch.randn(B, h, w, C, generator=g).to(torch.bfloat16).to(DEV) for h, w in sizes]
    affs = [torch.stack([1 + 0.2 * torch.randn(B, C, generator=g), 0.3 * torch.randn(B, C, generator=g)], 1).float().to(DEV).contiguous()
            for _ in sizes]
    lvl_list = [(l, min(l + 1, 3), max(l - 1, 0)) for l in range(4)]
    for use_aff in (False, True):
        a = affs if use_aff else [None] * 4
        out = K.MlvlMaps(B, sizes, C, DEV)
        K.fuse_shuffle_mlvl(maps, a, lvl_list, out)
        for tar, top, dow in lvl_list:
            want = K.fuse_shuffle(maps[tar], maps[top], maps[dow], a[tar], a[top], a[dow])
            d = (out.levels[tar].float() - want.float()).abs()
            assert torch.equal(out.levels[tar], want), (
                f"shuffle: affine={use_aff} level {tar}: {int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.3e}, "
                f"first at {[int(i) for i in (d > 0).nonzero()[0]] if (d > 0).any() else None}")
    z = K.MlvlMaps(B, sizes, C, DEV)
    for lv, m in zip(z.levels, maps):
        lv.copy_(m)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    got = K.groupnorm_affine_mlvl(z, gamma, beta, 16, 1e-5)
    for l in range(4):
        want = K.groupnorm_affine(z.levels[l], gamma, beta, 16, 1e-5)
        d = (got[l] - want).abs()
        assert torch.equal(got[l], want), f"groupnorm level {l}: {int((d > 0).sum())} of {d.numel()} differ, max {float(d.max()):.3e}"


def test_split_k_reduce_folded_into_the_next_rmsnorm_is_bit_identical():
    """g4r_gemm_bf16_nt_partials + g4r_rmsnorm_splitk_bf16 (the LLaMA down_proj's K-slice reduce + residual folded into the
    following RMSNorm) against the two-launch form it replaces: same residual stream, same normalised rows, bit for bit."""
    M, N, K_ = 767, 4096, 11008
    a, w = rnd(M, K_, scale=0.5, seed=400), rnd(N, K_, scale=0.02, seed=401)
    res = rnd(M, N, seed=402)
    gamma = (1 + 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(403))).to(DEV)
    tile, splits = K.long_k_plan(M, N, K_)
    x_want = K.gemm(a, w, residual=res)                       # production dispatch: the same tile x K slices + reduce launch
    h_want = K.rmsnorm(x_want, gamma, 1e-6)
    part, ns = K.gemm_partials(a, w, splits, tile)
    assert ns == splits and part.shape == (splits, M, N)
    x_got, h_got = K.rmsnorm_splitk(part, ns, res, gamma, 1e-6)
    assert torch.equal(x_got, x_want) and torch.equal(h_got, h_want)


@pytest.mark.parametrize("B,T,pos0,tile", [(1, 767, 0, 28), (1, 767, 0, 24), (2, 300, 17, 28), (3, 211, 5, 24)])
def test_qkv_projection_with_rope_and_cache_append_in_the_epilogue(B, T, pos0, tile):
    """g4r_gemm_qkv_rope_bf16 against the two launches it replaces (g4r_gemm_bf16_nt + g4r_rope_qkv_bf16 per sequence): the
    rotated queries and the appended cache rows must be bit-identical, cache rows outside [pos0, pos0 + T) untouched."""
    heads, D, Kd, maxT = 32, 128, 4096, 1024
    HD = heads * D
    h = rnd(B * T, Kd, seed=500)
    w = rnd(3 * HD, Kd, scale=1 / 64, seed=501)
    inv = 1.0 / (10000.0 ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    ang = torch.arange(maxT, dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = ang.cos().to(DEV).contiguous(), ang.sin().to(DEV).contiguous()
    qkv = K.gemm(h, w, tile_cfg=tile).view(B, T, 3 * HD)
    q_want = torch.empty((B, T, HD), dtype=torch.bfloat16, device=DEV)
    kc_want = torch.full((B, maxT, HD), 7.0, dtype=torch.bfloat16, device=DEV)
    vc_want = torch.full((B, maxT, HD), 7.0, dtype=torch.bfloat16, device=DEV)
    for b in range(B):
        K.rope_qkv(qkv[b], cos, sin, q_want[b], kc_want[b], vc_want[b], heads, D, pos0)
    q_got = torch.empty_like(q_want)
    kc_got, vc_got = torch.full_like(kc_want, 7.0), torch.full_like(vc_want, 7.0)
    assert K.gemm_qkv_rope(h, w, B, T, heads, D, q_got, kc_got, vc_got, cos, sin, pos0, tile_cfg=tile) is not None
    assert torch.equal(q_got, q_want) and torch.equal(kc_got, kc_want) and torch.equal(vc_got, vc_want)


def test_clip_fc2_reduce_folded_into_the_next_layernorm_is_bit_identical():
    """g4r_gemm_bf16_nt_partials + g4r_layernorm_splitk_bf16 (CLIP fc2's K-slice reduce + bias + residual folded into the next
    block's layer_norm1) against gemm(bias, residual) + layernorm(): bit for bit."""
    M, N, K_ = 577, 1024, 4096
    a, w = rnd(M, K_, scale=0.5, seed=410), rnd(N, K_, scale=0.02, seed=411)
    res = rnd(M, N, seed=412)
    g = torch.Generator().manual_seed(413)
    bias = (0.1 * torch.randn(N, generator=g)).to(DEV)
    gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).to(DEV), (0.1 * torch.randn(N, generator=g)).to(DEV)
    tile, splits = K.small_m_split_plan(M, N, K_)
    x_want = K.gemm(a, w, bias=bias, residual=res)
    h_want = K.layernorm(x_want, gamma, beta, 1e-5)
    part, ns = K.gemm_partials(a, w, splits, tile)
    x_got, h_got = K.layernorm_splitk(part, ns, bias, res, gamma, beta, 1e-5)
    assert ns == splits and torch.equal(x_got, x_want) and torch.equal(h_got, h_want)
