"""GPU: the backward / optimizer kernels of the training rows (include/g4r_train.h) against torch autograd on
the fp32 statement of the same op, and the LLaMA stack's input gradient against autograd through the CPU
oracle (oracle/transformer_oracle.py).  Everything goes through the C ABI."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import spi_oracle as S  # noqa: E402
from oracle import transformer_oracle as T  # noqa: E402

if torch.cuda.is_available():
    from gpt4roi_amd import kernels as K
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.layers import MLVLROIQueryModule
    from gpt4roi_amd.llama import LlamaDecoder

DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def relerr(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-9)).item()


def cosine(got, want):
    got, want = got.double().cpu().flatten(), want.double().cpu().flatten()      # (fp32 dot products of 10^7..10^8 terms drift by 1e-3)
    return (got @ want / (got.norm() * want.norm()).clamp_min(1e-30)).item()


def leaf(x):
    return x.detach().float().cpu().requires_grad_(True)


# ------------------------------------------------------------------------------------------ attention
def _attn(q, k, v, H, scale, causal):
    B, Tq, HD = q.shape
    Tk, D = k.size(1), HD // H
    qh, kh, vh = (t.view(B, -1, H, D).transpose(1, 2) for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) * scale
    if causal:
        i = torch.arange(Tq)[:, None] + (Tk - Tq)
        s = s.masked_fill(torch.arange(Tk)[None, :] > i, float("-inf"))
    return (s.softmax(-1) @ vh).transpose(1, 2).reshape(B, Tq, HD)


@pytest.mark.parametrize("B,H,D,Tq,Tk,causal", [
    (1, 2, 128, 64, 64, True), (2, 3, 128, 200, 200, True), (1, 2, 128, 767, 767, True),
    (1, 2, 64, 130, 130, False), (1, 4, 64, 97, 97, True), (1, 2, 128, 70, 150, True), (1, 1, 128, 33, 33, False)])
def test_flash_attention_backward(B, H, D, Tq, Tk, causal):
    q, k, v = rnd(B, Tq, H * D, seed=1), rnd(B, Tk, H * D, seed=2), rnd(B, Tk, H * D, seed=3)
    do = rnd(B, Tq, H * D, seed=4)
    scale = 1.0 / math.sqrt(D)
    lse = torch.empty((B, H, Tq), dtype=torch.float32, device=DEV)
    o = K.flash_attn(q, k, v, H, scale, causal, lse=lse)
    qr, kr, vr = leaf(q), leaf(k), leaf(v)
    ref = _attn(qr, kr, vr, H, scale, causal)
    ref.backward(do.float().cpu())
    # lse is in the log2 domain of the scaled scores
    s = (qr.view(B, Tq, H, D).transpose(1, 2) @ kr.view(B, Tk, H, D).transpose(1, 2).transpose(-1, -2)) * scale
    if causal:
        i = torch.arange(Tq)[:, None] + (Tk - Tq)
        s = s.masked_fill(torch.arange(Tk)[None, :] > i, float("-inf"))
    assert relerr(lse, torch.logsumexp(s.detach(), -1) * 1.4426950408889634) < 5e-3
    dq, dk, dv = K.flash_attn_bwd(q, k, v, o, do, lse, H, scale, causal)
    assert relerr(dq, qr.grad) < 3e-2, ("dq", relerr(dq, qr.grad))
    assert relerr(dk, kr.grad) < 3e-2, ("dk", relerr(dk, kr.grad))
    assert relerr(dv, vr.grad) < 3e-2, ("dv", relerr(dv, vr.grad))


def test_flash_attention_backward_strided_cache():
    # K/V read from a [maxT, C] cache slice, q/dO contiguous: the layout LlamaDecoder.backward uses
    B, H, D, T, maxT = 2, 2, 128, 90, 128
    C = H * D
    kc, vc = rnd(B, maxT, C, seed=5), rnd(B, maxT, C, seed=6)
    q, do = rnd(B, T, C, seed=7), rnd(B, T, C, seed=8)
    k, v = kc[:, :T], vc[:, :T]
    lse = torch.empty((B, H, T), dtype=torch.float32, device=DEV)
    o = K.flash_attn(q, k, v, H, 0.09, True, lse=lse)
    dq, dk, dv = K.flash_attn_bwd(q, k, v, o, do, lse, H, 0.09, True)
    qr, kr, vr = leaf(q), leaf(k), leaf(v)
    _attn(qr, kr, vr, H, 0.09, True).backward(do.float().cpu())
    assert max(relerr(dq, qr.grad), relerr(dk, kr.grad), relerr(dv, vr.grad)) < 3e-2


# ------------------------------------------------------------------------------------------ row kernels
def test_rmsnorm_backward():
    x, dy, dres = rnd(37, 1024, scale=2.0, seed=10), rnd(37, 1024, seed=11), rnd(37, 1024, seed=12)
    g = 1 + rnd(1024, scale=0.2, seed=13, dtype=torch.float32)
    xr, gr = leaf(x), leaf(g)
    y = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6) * gr
    y.backward(dy.float().cpu())
    dgamma = torch.zeros(1024, dtype=torch.float32, device=DEV)
    dx = K.rmsnorm_bwd(x, g, dy, dres=dres, dgamma=dgamma, eps=1e-6)
    assert relerr(dx, xr.grad + dres.float().cpu()) < 1e-2
    assert relerr(dgamma, gr.grad) < 1e-3
    dx2 = K.rmsnorm_bwd(x, g, dy, eps=1e-6)
    assert relerr(dx2, xr.grad) < 1e-2


@pytest.mark.parametrize("relu_in", [False, True])
def test_layernorm_backward(relu_in):
    x, dy = rnd(29, 256, scale=1.5, seed=14), rnd(29, 256, seed=15)
    g, b = 1 + rnd(256, scale=0.2, seed=16, dtype=torch.float32), rnd(256, seed=17, dtype=torch.float32)
    xr, gr, br = leaf(x), leaf(g), leaf(b)
    y = F.layer_norm(F.relu(xr) if relu_in else xr, (256,), gr, br, 1e-5)
    y.backward(dy.float().cpu())
    dgamma, dbeta = (torch.zeros(256, dtype=torch.float32, device=DEV) for _ in range(2))
    dx = K.layernorm_bwd(x, g, dy, dgamma, dbeta, 1e-5, relu_in)
    assert relerr(dx, xr.grad) < 1e-2
    assert relerr(dgamma, gr.grad) < 1e-3 and relerr(dbeta, br.grad) < 1e-3


def test_swiglu_interleaved_forward_backward():
    T_, F_ = 45, 352
    gu, dy = rnd(T_, 2 * F_, scale=1.5, seed=18), rnd(T_, F_, seed=19)
    r = leaf(gu)
    y = F.silu(r[:, 0::2]) * r[:, 1::2]
    y.backward(dy.float().cpu())
    assert relerr(K.swiglu_il(gu), y.detach()) < 1e-2
    assert relerr(K.swiglu_il_bwd(gu, dy), r.grad) < 1e-2


def test_rope_backward_is_the_transpose_of_forward():
    T_, H, D, pos0 = 23, 3, 128, 5
    C = H * D
    cos, sin = (t.to(DEV).contiguous() for t in T.rope_tables(64, D))
    qkv = rnd(T_, 3 * C, seed=20)
    dq, dk, dv = rnd(T_, C, seed=21), rnd(T_, C, seed=22), rnd(T_, C, seed=23)
    r = leaf(qkv)
    q = T.apply_rope(r[None, :, :C], cos.cpu(), sin.cpu(), H, pos0)[0]
    k = T.apply_rope(r[None, :, C:2 * C], cos.cpu(), sin.cpu(), H, pos0)[0]
    ((q * dq.float().cpu()).sum() + (k * dk.float().cpu()).sum() + (r[:, 2 * C:] * dv.float().cpu()).sum()).backward()
    got = K.rope_qkv_bwd(dq, dk, dv, cos, sin, H, D, pos0)
    assert relerr(got, r.grad) < 1e-2


def test_cross_entropy_loss_and_gradient():
    R, N, n_pad = 19, 1003, 1024
    logits = rnd(R, N, scale=3.0, seed=24, dtype=torch.float32)
    labels = torch.randint(0, N, (R,), generator=torch.Generator().manual_seed(25))
    labels[3] = -100
    labels[11] = -100
    r = leaf(logits)
    ref = F.cross_entropy(r, labels, ignore_index=-100)
    ref.backward()
    n_valid = int((labels >= 0).sum())
    loss_sum = torch.zeros(1, dtype=torch.float32, device=DEV)
    gs = torch.full((1,), 1.0 / n_valid, dtype=torch.float32, device=DEV)
    dl = torch.full((R, n_pad), 7.0, dtype=torch.bfloat16, device=DEV)
    K.cross_entropy(logits, labels.to(DEV), loss_sum, gs, dl, n_pad)
    assert abs(loss_sum.item() / n_valid - ref.item()) < 1e-4 * abs(ref.item())
    assert relerr(dl[:, :N], r.grad) < 1e-2
    assert float(dl[:, N:].abs().max()) == 0.0 and float(dl[3].abs().max()) == 0.0


def test_transpose_colsum_relu_gather():
    x = rnd(70, 200, seed=26)
    t = K.transpose(x, 128)
    assert torch.equal(t[:, :70], x.t()) and float(t[:, 70:].abs().max()) == 0.0
    xs = rnd(50, 96, seed=27)[:, :64]                      # row-strided input
    assert torch.equal(K.transpose(xs), xs.t().contiguous())
    big = rnd(700, 300, seed=28)
    assert relerr(K.colsum(big), big.float().sum(0)) < 1e-3
    y, dy = rnd(16, 64, seed=29), rnd(16, 64, seed=30)
    assert torch.equal(K.relu_bwd(y, dy), torch.where(y > 0, dy, torch.zeros_like(dy)))
    idx = torch.tensor([5, -1, 0, 69, 5], dtype=torch.int32, device=DEV)
    g = K.gather_rows(x[:, :64], idx)
    want = x[:, :64][idx.clamp(min=0).long()].clone()
    want[1] = 0
    assert torch.equal(g, want)


def test_linear_gradients_through_nt_gemm():
    M, N, Kd = 150, 192, 256
    x, w, dy = rnd(M, Kd, seed=31), rnd(N, Kd, scale=0.1, seed=32), rnd(M, N, seed=33)
    xr, wr = leaf(x), leaf(w)
    (xr @ wr.t()).backward(dy.float().cpu())
    dx = K.linear_dgrad(dy, K.transpose(w))
    dw = K.linear_wgrad(dy, x)
    assert relerr(dx, xr.grad) < 1e-2 and relerr(dw, wr.grad) < 1e-2


@pytest.mark.parametrize("Kd,M,N,slices", [(150, 192, 256, None), (1000, 520, 264, 1), (5000, 256, 1088, None),
                                           (333, 24, 8, 3), (4096, 1024, 512, 7)])
def test_gemm_tn_reads_row_major_operands(Kd, M, N, slices):
    """C = a^T b over the rows of both operands (g4r_gemm_tn_bf16): edge tiles in M and N, a K that is no multiple of the K
    tile, row-strided operands, the direct / sliced / accumulating forms."""
    big_a, big_b = rnd(Kd, M + 16, seed=35), rnd(Kd, N + 8, seed=36)
    a, b = big_a[:, 8:8 + M], big_b[:, :N]                          # row strides > widths
    want = a.float().cpu().t() @ b.float().cpu()
    got = K.gemm_tn(a, b, slices=slices)
    assert relerr(got, want) < 2e-3, relerr(got, want)
    acc = K.gemm_tn(a, b, out=got.clone(), accumulate=True, slices=slices)
    assert relerr(acc, 2 * want) < 2e-3
    assert torch.equal(K.gemm_tn(a, b, slices=slices), got)


def test_adamw_matches_torch():
    n = 5000
    p0 = rnd(n, seed=34, dtype=torch.float32)
    ref_p = torch.nn.Parameter(p0.cpu().clone())
    opt = torch.optim.AdamW([ref_p], lr=2e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pb = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step in range(1, 4):
        g = rnd(n, seed=40 + step, dtype=torch.float32)
        ref_p.grad = g.cpu() * 0.5
        opt.step()
        K.adamw(p, g, m, v, step, 2e-3, (0.9, 0.95), 1e-8, 0.1, grad_scale=0.5, param_bf16=pb)
    assert relerr(p, ref_p.detach()) < 1e-5
    assert torch.equal(pb, p.to(torch.bfloat16))
    gb = rnd(n, seed=50)
    K.adamw(p, gb, m, v, 4, 1e-3)                                    # bf16 gradient path
    assert torch.isfinite(p).all()


# ------------------------------------------------------------------------------------------ LLaMA stack
@pytest.mark.parametrize("B,T_", [(1, 75), (2, 40)])
def test_llama_input_gradient_and_loss(B, T_):
    hidden, inter, layers, vocab, heads = 256, 384, 2, 1003, 2
    sd = syn.llama_state(hidden, inter, layers, vocab, seed=3)
    dec = LlamaDecoder(sd, heads=heads, max_positions=128, device=DEV, max_batch=B)
    dec.prepare_training()
    emb = rnd(B, T_, hidden, seed=60)
    labels = torch.randint(0, vocab, (B, T_), generator=torch.Generator().manual_seed(61))
    labels[:, :T_ // 3] = -100
    logits, ctx = dec.forward_train(emb)
    loss, dlogits = dec.loss_and_dlogits(logits, labels.to(DEV))
    dx = dec.backward(ctx, dlogits)
    # oracle: same weights (bf16-rounded, as the decoder stores them), fp32 autograd
    w = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    er = leaf(emb)
    hid, _ = T.llama_forward(w, er, heads)
    ref_logits = T.lm_logits(w, hid)
    ref_loss = F.cross_entropy(ref_logits[:, :-1].reshape(-1, vocab), labels[:, 1:].reshape(-1), ignore_index=-100)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
    e = relerr(dx.view(B, T_, hidden), er.grad)
    assert e < 6e-2, e


def test_llama_weight_gradients():
    hidden, inter, layers, vocab, heads, B, T_ = 256, 384, 1, 515, 2, 1, 70
    sd = syn.llama_state(hidden, inter, layers, vocab, seed=4)
    dec = LlamaDecoder(sd, heads=heads, max_positions=128, device=DEV)
    dec.prepare_training(train_weights=True)
    emb = rnd(B, T_, hidden, seed=62)
    labels = torch.randint(0, vocab, (B, T_), generator=torch.Generator().manual_seed(63))
    logits, ctx = dec.forward_train(emb)
    loss, dlogits = dec.loss_and_dlogits(logits, labels.to(DEV))
    dec.backward(ctx, dlogits)
    w = {k: v.to(torch.bfloat16).float().requires_grad_(True) for k, v in sd.items()}
    hid, _ = T.llama_forward(w, emb.float().cpu(), heads)
    F.cross_entropy(T.lm_logits(w, hid)[:, :-1].reshape(-1, vocab), labels[:, 1:].reshape(-1)).backward()
    p = "model.layers.0."
    g = dec.grads
    ref_qkv = torch.cat([w[p + f"self_attn.{n}_proj.weight"].grad for n in "qkv"], 0)
    ref_gu = torch.stack([w[p + "mlp.gate_proj.weight"].grad, w[p + "mlp.up_proj.weight"].grad], 1).reshape(-1, hidden)
    for name, got, want in (("lm_head", g["lm_head"], w["lm_head.weight"].grad),
                            ("wd", g["0.wd"], w[p + "mlp.down_proj.weight"].grad),
                            ("wgu", g["0.wgu"], ref_gu), ("wo", g["0.wo"], w[p + "self_attn.o_proj.weight"].grad),
                            ("wqkv", g["0.wqkv"], ref_qkv),
                            ("n1", g["0.n1"], w[p + "input_layernorm.weight"].grad),
                            ("n2", g["0.n2"], w[p + "post_attention_layernorm.weight"].grad),
                            ("norm", g["norm"], w["model.norm.weight"].grad)):
        e = relerr(got, want)
        assert e < 6e-2, (name, e)


# ------------------------------------------------------------------------------------------ region module
def test_groupnorm_relu_backward():
    B, H, W, C, G = 2, 9, 7, 512, 64
    z = rnd(B, H, W, C, scale=1.5, seed=70) + 0.3
    dy = rnd(B, H, W, C, seed=71, dtype=torch.float32)
    g, b = 1 + rnd(C, scale=0.2, seed=72, dtype=torch.float32), rnd(C, scale=0.3, seed=73, dtype=torch.float32)
    zr, gr, br = leaf(z), leaf(g), leaf(b)
    y = F.relu(F.group_norm(zr.permute(0, 3, 1, 2), G, gr, br, 1e-5)).permute(0, 2, 3, 1)
    y.backward(dy.cpu())
    aff = K.groupnorm_affine(z, g, b, G, 1e-5)
    stats = K.groupnorm_stats(z, G, 1e-5)
    dgamma, dbeta = (torch.zeros(C, dtype=torch.float32, device=DEV) for _ in range(2))
    dz = K.gn_relu_bwd(z, dy, aff, g, stats, dgamma, dbeta, G)
    assert relerr(dz, zr.grad) < 1.5e-2
    assert relerr(dgamma, gr.grad) < 2e-3 and relerr(dbeta, br.grad) < 2e-3


def test_fuse_shuffle_backward_is_the_transpose():
    B, C = 2, 64
    sizes = {"own": 12, "top": 6, "down": 24}
    src = {k: rnd(B, v, v, C, seed=74 + i) for i, (k, v) in enumerate(sizes.items())}
    dinp = rnd(B, 12, 12, C, seed=78)
    leaves = {k: leaf(v) for k, v in src.items()}

    def interp(t, n):
        return F.interpolate(t.permute(0, 3, 1, 2), size=(n, n), mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    R, Sh = C // 2, C // 4
    out = torch.cat([leaves["own"][..., :R], interp(leaves["top"][..., R + Sh:], 12),
                     interp(leaves["down"][..., R:R + Sh], 12)], -1)
    assert relerr(K.fuse_shuffle(src["own"], src["top"], src["down"], None, None, None), out.detach()) < 1e-2
    out.backward(dinp.float().cpu())
    d = {k: torch.zeros(v.shape, dtype=torch.float32, device=DEV) for k, v in src.items()}
    K.fuse_shuffle_bwd(dinp, d["own"], d["top"], d["down"])
    for k in d:
        assert relerr(d[k], leaves[k].grad) < 1e-4, k


def test_fuse_shuffle_backward_gather_equals_scatter():
    """The per-source-level gather (used by MLVLFuseModule.backward) against the atomic scatter transpose above,
    over a whole 4-level round with the reference's neighbour table (l, min(l+1, 3), max(l-1, 0))."""
    B, C, sizes = 2, 64, [24, 12, 6, 3]
    dinps = [rnd(B, n, n, C, seed=100 + i) for i, n in enumerate(sizes)]
    want = [torch.zeros((B, n, n, C), dtype=torch.float32, device=DEV) for n in sizes]
    for tar in range(4):
        K.fuse_shuffle_bwd(dinps[tar], want[tar], want[min(tar + 1, 3)], want[max(tar - 1, 0)])
    for l in range(4):
        got = K.fuse_shuffle_bwd_gather(l, dinps)
        assert relerr(got, want[l]) < 1e-5, l


@pytest.mark.parametrize("B,H,W,cin,cout", [(1, 14, 14, 64, 128), (2, 20, 12, 128, 64),
                                            (2, 20, 12, 256, 256), (1, 24, 24, 512, 256), (3, 7, 33, 256, 768)])
def test_conv3x3_weight_and_input_gradients(B, H, W, cin, cout):
    """Channel counts that are multiples of 256 take the NHWC (TN) kernel of csrc/gemm_tn.hip, the others the
    channel-major NT path."""
    x, dy = rnd(B, H, W, cin, seed=80), rnd(B, H, W, cout, seed=81)
    w = rnd(cout, cin, 3, 3, scale=0.1, seed=82)
    xr, wr = leaf(x), leaf(w)
    F.conv2d(xr.permute(0, 3, 1, 2), wr, padding=1).backward(dy.float().cpu().permute(0, 3, 1, 2))
    plan = K.ConvWgradPlan(B, H, W, cin, cout, DEV)
    dw = plan.wgrad(x, dy)
    assert relerr(dw, wr.grad) < 1e-2
    assert plan.nhwc == (cin % 256 == 0 and cout % 256 == 0)
    dw2 = plan.wgrad(x, dy)                                     # plan buffers are reusable
    assert torch.equal(dw, dw2)
    dw3 = plan.wgrad(x, dy, accumulate_into=dw2.clone())        # += form (the levels of a pyramid share the weight)
    assert relerr(dw3, 2 * wr.grad) < 1e-2
    dx = K.conv3x3(dy, K.conv3x3_dgrad_weight(w))
    assert relerr(dx, xr.grad) < 1e-2


@pytest.mark.parametrize("G,co,ci", [(1, 64, 48), (4, 32, 40), (1, 1024, 1024)])
def test_conv_weight_layouts_in_one_pass(G, co, ci):
    """g4r_conv3x3_weight_layout: fp32 torch conv weights -> the forward rows [Co, G*9*Ci] and the data-gradient rows
    [G*Ci, 9*Co] (rotated, channel-transposed filter) == the torch permute / flip / cat / cast expressions, bit for bit."""
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(co, ci, 3, 3, generator=g).to(DEV) for _ in range(G)]
    for dt in (torch.bfloat16, torch.float16):
        want = torch.cat([w.permute(0, 2, 3, 1).reshape(co, 1, 9 * ci) for w in ws], 1).reshape(co, -1).to(dt)
        got = K.prep_conv3x3_weight(ws, dt)
        assert got.dtype == dt and torch.equal(got, want)
    wt = torch.cat([w.flip(2, 3).permute(1, 0, 2, 3) for w in ws], 0)
    want = wt.permute(0, 2, 3, 1).reshape(G * ci, 9 * co).to(torch.bfloat16)
    assert torch.equal(K.conv3x3_dgrad_weight(ws), want)
    assert torch.equal(K.conv3x3_dgrad_weight([w.to(torch.bfloat16) for w in ws]), want)      # other dtypes: the torch path


def test_conv3x3_weight_gradient_of_a_pyramid_in_one_launch():
    """ConvWgradNHWC over several map geometries that share the weight (the levels of a fuse round,
    gpt4roi/models/layers.py:218-236) == the sum of the per-level autograd gradients; slices of different levels end up in
    one partial buffer and one reduce."""
    B, cin, cout = 2, 256, 512
    sizes = [(24, 24), (12, 12), (6, 6), (3, 3)]
    w = rnd(cout, cin, 3, 3, scale=0.1, seed=90)
    wr = leaf(w)
    xs = [rnd(B, h, ww, cin, seed=91 + i) for i, (h, ww) in enumerate(sizes)]
    dys = [rnd(B, h, ww, cout, seed=95 + i) for i, (h, ww) in enumerate(sizes)]
    for x, dy in zip(xs, dys):
        F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wr, padding=1).backward(dy.float().cpu().permute(0, 3, 1, 2))
    plan = K.ConvWgradNHWC(B, sizes, cin, cout, DEV)
    assert plan.slices >= len(sizes)
    dw = plan.wgrad(xs, dys)
    assert relerr(dw, wr.grad) < 1e-2
    assert torch.equal(plan.wgrad(xs, dys), dw)                 # fixed summation order: bit-reproducible
    singles = None
    for (h, ww), x, dy in zip(sizes, xs, dys):
        singles = K.ConvWgradPlan(B, h, ww, cin, cout, DEV).wgrad(x, dy, accumulate_into=singles)
    assert relerr(singles, dw) < 1e-5                           # the same products, another grouping of the fp32 sums


def test_roi_align_mlvl_backward():
    B, C, N, L = 2, 64, 5, 2
    sizes, scales = [80, 12], [1.0, 0.15]          # level 0: wide RoIs (direct path); level 1: narrow (LDS-staged path)
    rois = torch.tensor([[0, 3.3, 4.1, 30.2, 25.7], [1, 10.0, 2.0, 75.0, 70.0], [0, -3.0, -2.0, 9.0, 12.0],
                         [1, 20.5, 20.5, 21.0, 23.0], [0, 0.0, 0.0, 47.9, 47.9]], dtype=torch.float32)
    dout = rnd(N, 7, 7, L * C, seed=83)
    # gather kernel (default): every texel is WRITTEN once -> start from NaN-poisoned maps; bit-reproducible
    grads = [torch.full((B, s, s, C), float("nan"), dtype=torch.float32, device=DEV) for s in sizes]
    K.roi_align_mlvl_bwd(dout, C, L * C, grads, rois.to(DEV), 7, scales, 2, True)
    again = [torch.full((B, s, s, C), float("nan"), dtype=torch.float32, device=DEV) for s in sizes]
    K.roi_align_mlvl_bwd(dout, C, L * C, again, rois.to(DEV), 7, scales, 2, True)
    # the reference-style atomic scatter (kept for A/B) into zeroed maps
    atom = [torch.zeros((B, s, s, C), dtype=torch.float32, device=DEV) for s in sizes]
    K.roi_align_mlvl_bwd(dout, C, L * C, atom, rois.to(DEV), 7, scales, 2, True, atomic=True)
    # RoIs grouped by image + per-image offsets (what the region module passes)
    order = torch.tensor([0, 2, 4, 1, 3])
    offs = torch.tensor([0, 3, 5], dtype=torch.int32, device=DEV)
    grouped = [torch.full((B, s, s, C), float("nan"), dtype=torch.float32, device=DEV) for s in sizes]
    K.roi_align_mlvl_bwd(dout[order.to(DEV)].contiguous(), C, L * C, grouped, rois[order].to(DEV), 7, scales, 2, True,
                         roi_offsets=offs)
    from oracle import roi_align as RO
    for l in range(L):
        g = dout[..., l * C:(l + 1) * C].float().cpu().permute(0, 3, 1, 2).contiguous().numpy()
        want = RO.backward(g, rois.numpy(), (B, C, sizes[l], sizes[l]), 7, scales[l], 2, "avg", True)
        assert torch.isfinite(grads[l]).all() and torch.equal(grads[l], again[l])
        assert relerr(grads[l].permute(0, 3, 1, 2), torch.from_numpy(want)) < 1e-5, l
        assert relerr(atom[l].permute(0, 3, 1, 2), torch.from_numpy(want)) < 1e-5, l
        assert relerr(grouped[l], grads[l]) < 1e-6, l


@pytest.mark.parametrize("rounds", [1, 2])
def test_fuse_module_backward_isolated(rounds):
    """MLVLFuseModule.backward against autograd through the oracle from IDENTICAL level inputs (so the ReLU masks of
    the two pipelines agree and a random upstream gradient is a fair probe)."""
    from gpt4roi_amd.layers import MLVLFuseModule
    C, P, B = 512, 4, 2
    m = MLVLFuseModule(input_dims=C, embed_dims=C, num_levels=4, num_fuse=rounds)
    o = S.MLVLFuseOracle(C, C, 4, num_fuse=rounds)
    sd = S.synthetic_state(o, 11)
    o.load_state_dict(sd)
    m.load_state_dict(sd)
    m.to(DEV)
    g = torch.Generator().manual_seed(12)
    toks = [torch.randn(B, P * P, C, generator=g).to(torch.bfloat16) for _ in range(4)]
    sizes = [P * 2 ** l for l in range(4)][::-1]
    pyr = [S._r(F.interpolate(t.float().reshape(B, P, P, C).permute(0, 3, 1, 2), size=(n, n), mode="bilinear",
                              align_corners=True), True) for t, n in zip(toks, sizes)]
    ys = o(pyr, emulate=True)
    d_y = [torch.randn(y.shape, generator=g) for y in ys]
    sum((y * d).sum() for y, d in zip(ys, d_y)).backward()
    maps, affs, ctx = m.forward_train([t.to(DEV) for t in toks], P, sizes)
    for l in range(4):
        y = torch.relu(maps[l].float() * affs[l][:, 0][:, None, None, :] + affs[l][:, 1][:, None, None, :])
        assert relerr(y.permute(0, 3, 1, 2), ys[l].detach()) < 1e-2, l
        flips = ((y.permute(0, 3, 1, 2).cpu() > 0) != (ys[l].detach() > 0)).float().mean().item()
        print(f"rounds {rounds} level {l}: ReLU mask mismatches {flips:.2e}")
        # the loose max-norm tolerances below are attributed to ReLU-mask flips between two bf16 pipelines: that rate itself
        # is bounded here (identical inputs: a few 1e-4 observed), so a kernel bug cannot hide behind the explanation
        assert flips < 2e-3, (l, flips)
    grads = m.backward(ctx, [d.permute(0, 2, 3, 1).contiguous().to(DEV) for d in d_y])
    ref = {k: v.grad for k, v in o.named_parameters()}
    assert set(grads) == set(ref)
    # Even from identical inputs ~2e-4 of the ReLU masks differ between the two bf16 pipelines (printed above); with
    # a random upstream gradient every flipped unit moves a bias-like gradient by a full |d_y|, i.e. a few per cent
    # of the largest entry (gn.bias / conv.weight), while gn.weight (flipped units have xhat ~ 0) stays at 3e-3.
    errs = {k: round(relerr(grads[k], ref[k]), 4) for k in sorted(ref)}
    coss = {k: round(cosine(grads[k], ref[k]), 5) for k in sorted(ref)}
    print("fuse gradient errors:", errs, "cosines:", coss)
    assert max(errs.values()) < 0.12 and min(coss.values()) > 0.997, (errs, coss)
    assert errs[f"fuse_convs.{rounds - 1}.gn.weight"] < 6e-3


def test_region_module_parameter_gradients():
    """Every parameter gradient of MLVLROIQueryModule against autograd through the oracle (bf16 rounding points)."""
    C, P, B, out_dims = 512, 8, 2, 512
    m = MLVLROIQueryModule(embed_dims=C, out_dims=out_dims, num_levels=4)
    o = S.MLVLROIQueryOracle(embed_dims=C, P=P)
    o.roi_align.updims = torch.nn.Linear(1024, out_dims)
    sd = S.synthetic_state(o, 5)
    o.load_state_dict(sd)
    m.load_state_dict(sd)
    m.to(DEV)
    feats, boxes = S.synthetic_inputs(6, B, P, C, [4, 2])
    want = torch.cat(o([f.to(torch.bfloat16).float() for f in feats], boxes, emulate=True), 0)
    # upstream gradient of the loss sum(out^2)/2.  (A RANDOM upstream gradient turns every parameter gradient into
    # a random-walk sum, and the ~0.5 % of ReLU masks that differ between two bf16 pipelines then show up as a
    # sqrt(fraction) ~ 10 % error that says nothing about the kernels.)
    d_out = want.detach().to(torch.bfloat16).to(DEV)
    (want * d_out.float().cpu()).sum().backward()
    toks = [f.to(DEV).to(torch.bfloat16) for f in feats]
    out, ctx = m.forward_train(toks, [b.to(DEV) for b in boxes])
    assert relerr(out, want.detach()) < 1.5e-2
    grads = m.backward(ctx, d_out)
    ref = {k: v.grad for k, v in o.named_parameters()}
    assert set(grads) == set(ref), set(grads) ^ set(ref)
    worst = {}
    for k in sorted(ref):
        assert grads[k].shape == ref[k].shape, (k, grads[k].shape, ref[k].shape)
        worst[k] = relerr(grads[k], ref[k])
    coss = {k: cosine(grads[k], ref[k]) for k in ref}
    print("gradient errors:", [(k, round(v, 4), round(coss[k], 5)) for k, v in sorted(worst.items())])
    # the head of the module (no ReLU-mask sensitivity) must be tight; through the 5 GN+ReLU rounds the mask flips
    # between two bf16 pipelines dominate the max-norm error (see test_fuse_module_backward_isolated)
    head = [k for k in ref if k.startswith(("roi_align.updims", "roi_align.pos_embedd", "roi_align.flatten_linear"))]
    assert max(worst[k] for k in head) < 1e-2, {k: worst[k] for k in head}
    assert max(worst.values()) < 0.2 and min(coss.values()) > 0.99, (worst, coss)


def test_region_module_parameter_gradients_at_the_training_shape():
    """the region module's backward at the geometry of configs[2] (train_stage1.sh:8; B = 8 images, P = 24,
    C = 1024, out 4096, 1..15 regions per image as refcoco.py:55 draws them) -- every parameter gradient of the HIP path against
    autograd through oracle/spi_oracle.py (pinned by tests/golden/spi_module_ref_grads_c64.npz to the reference's own
    layers.py:96-335), run with PyTorch-ROCm's fp32 kernels on the device (arithmetic independent of gpt4roi_amd/; the RoIAlign
    node stays the C oracle on the host).  Upstream gradient = d(sum(out^2) / 2): aligned with the forward, so the few 1e-4
    of ReLU masks that differ between two pipelines rounding to bf16 enter as a small relative error instead of a random walk.

    What separates "bf16 noise" from "kernel bug": the oracle is run TWICE, with the bf16 rounding points (the comparison target)
    and in exact fp32.  The distance between those two is what the storage type itself costs a parameter's gradient (it grows
    with depth: five GroupNorm + ReLU rounds lie between the loss and the input convolutions); the HIP gradient must sit
    within 1.25x of that yardstick, or within the absolute bounds 3e-2 (max-norm) / 1e-3 (1 - cosine), whichever is larger --
    the criterion of the greedy-id test applied to gradients.  Measured (profiles/r05_pytest_gpu.log): worst max-norm error
    6.2e-2 on fuse_convs.0.conv.weight, worst cosine 0.9975 on input_conv.3.weight, every parameter of the last two fuse
    rounds and of the head under 3e-2 / 0.9995 (the mini-width test above needs 0.2 / 0.99)."""
    C, P, B, out_dims = 1024, 24, 8, 4096
    m = MLVLROIQueryModule(embed_dims=C, out_dims=out_dims, num_levels=4)
    sd = None
    g = torch.Generator().manual_seed(77)
    n_i = torch.randint(1, 16, (B,), generator=g).tolist()
    feats, boxes = S.synthetic_inputs(6, B, P, C, n_i)
    dboxes = [b.to(DEV) for b in boxes]
    d_out, want, refs = None, None, {}
    for emulate in (True, False):
        o = S.MLVLROIQueryOracle(embed_dims=C, P=P)
        if sd is None:
            sd = S.synthetic_state(o, 5)
        o.load_state_dict(sd)
        o.to(DEV)
        y = torch.cat(o([f.to(torch.bfloat16).float().to(DEV) for f in feats], dboxes, emulate=emulate), 0)
        if d_out is None:
            d_out, want = y.detach().to(torch.bfloat16), y.detach().cpu()
        (y * d_out.float()).sum().backward()
        refs[emulate] = {k: v.grad.detach().cpu() for k, v in o.named_parameters()}
        del o, y
        torch.cuda.empty_cache()
    ref, ref32 = refs[True], refs[False]
    m.load_state_dict(sd)
    m.to(DEV)
    toks = [f.to(DEV).to(torch.bfloat16) for f in feats]
    out, ctx = m.forward_train(toks, dboxes)
    assert out.shape == (sum(n_i), out_dims) and relerr(out, want) < 1.5e-2
    grads = m.backward(ctx, d_out)
    assert set(grads) == set(ref), set(grads) ^ set(ref)
    rows, bad = [], []
    for k in sorted(ref):
        e, c = relerr(grads[k], ref[k]), cosine(grads[k], ref[k])
        ye, yc = relerr(ref[k], ref32[k]), cosine(ref[k], ref32[k])
        rows.append((round(e, 4), round(1 - c, 5), round(ye, 4), round(1 - yc, 5), k))
        if e > max(3e-2, 1.25 * ye) or (1 - c) > max(1e-3, 1.25 * (1 - yc)):
            bad.append(rows[-1])
    print(f"training shape B={B} P={P} C={C}, regions {n_i}: per parameter (HIP max-norm err, 1 - cos | bf16-vs-fp32 yardstick err, "
          f"1 - cos): worst eight by error {sorted(rows)[-8:]}")
    assert not bad, f"gradients further from the bf16 oracle than the storage type's own distance to fp32: {bad}"
    head = [r for r in rows if r[4].startswith(("roi_align.updims", "roi_align.pos_embedd", "roi_align.flatten_linear"))]
    assert max(r[0] for r in head) < 1e-2 and max(r[1] for r in head) < 1e-4, head


def test_conv3x3_weight_gradient_tn_kernel_at_192x192x1024():
    """VERDICT r04 item 5, second half: the TN weight-gradient kernel (csrc/gemm_tn.hip, g4r_conv3x3_wgrad_nhwc_bf16) ALONE at
    the largest map of the pyramid (192 x 192 x 1024 -> 1024, K = 36 864 pixels per tap) against fp32 torch autograd of
    F.conv2d on the device.  Operands are bf16 on both sides, so the only difference is the order of the fp32 sums."""
    H = W = 192
    cin = cout = 1024
    x, dy = rnd(1, H, W, cin, seed=180), rnd(1, H, W, cout, scale=0.05, seed=181)
    w = torch.zeros(cout, cin, 3, 3, device=DEV, requires_grad=True)
    F.conv2d(x.float().permute(0, 3, 1, 2), w, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    plan = K.ConvWgradNHWC(1, [(H, W)], cin, cout, DEV)
    dw = plan.wgrad([x], [dy])
    e, c = relerr(dw, w.grad), cosine(dw, w.grad)
    print(f"conv3x3 weight gradient 192x192x1024: max-norm err {e:.2e}, cosine {c:.7f}")
    assert dw.shape == (cout, cin, 3, 3) and e < 2e-3 and c > 0.99999


# ------------------------------------------------------------------------------------------ whole step
def test_stage1_training_step_end_to_end():
    """forward -> loss -> backward -> clip -> AdamW of train.RegionTrainer on a mini model: loss and every region-
    module gradient against autograd through the CPU oracles, the update against torch.optim.AdamW fed with the same
    gradients, and the loss must go down when the step is repeated on the batch."""
    from gpt4roi_amd.spi_llava import SPILlavaLlamaModel
    from gpt4roi_amd.train import RegionTrainer, cosine_lr
    from gpt4roi_amd.vit import ClipVisionTower
    H, P, image = 512, 8, 112
    ids = syn.token_ids(vocab_base=990)
    vsd = syn.vit_state(H, 4 * H, 12, image, seed=8)
    lsd = syn.llama_state(512, 1408, 2, ids.vocab, seed=9)
    tower = ClipVisionTower(vsd, heads=8, device=DEV)
    dec = LlamaDecoder(lsd, heads=4, max_positions=256, device=DEV)
    model = SPILlavaLlamaModel(tower, dec, ids, embed_dims=H)
    orc = S.MLVLROIQueryOracle(embed_dims=H, P=P)
    orc.roi_align.updims = torch.nn.Linear(1024, 512)
    spi_sd = S.synthetic_state(orc, 10)
    orc.load_state_dict(spi_sd)
    model.spi_module.load_state_dict(spi_sd)
    g = torch.Generator().manual_seed(11)
    pw, pb = torch.randn(512, H, generator=g) / H ** 0.5, torch.randn(512, generator=g) * 0.05
    with torch.no_grad():
        model.mm_projector.weight.copy_(pw)
        model.mm_projector.bias.copy_(pb)
    img = torch.randn(1, 3, image, image, generator=g)
    boxes = [syn.boxes(3, g)]
    prompt = syn.prompt_ids(ids, P, 3, g, sys_len=6, question_len=9, vocab_base=990)[None]
    labels = prompt.clone()
    labels[:, :8 + P * P] = -100                                   # system prompt and image: no loss
    labels[labels >= 990] = -100                                   # special tokens are never targets
    tr = RegionTrainer(model, lr=2e-6, max_grad_norm=1.0)
    dev = lambda t: t.to(DEV)  # noqa: E731
    loss, grads = tr.loss_and_grads(dev(prompt), dev(img), [dev(b) for b in boxes], dev(labels))
    from gpt4roi_amd.spi_llava import SPILlavaMPTForCausalLM
    ev = SPILlavaMPTForCausalLM(model)(input_ids=dev(prompt), images=dev(img), bboxes=[dev(b) for b in boxes],
                                       labels=dev(labels))          # the reference's forward(labels=...) call shape
    assert abs(ev.loss.item() - loss.item()) < 2e-2 * abs(loss.item()), (ev.loss.item(), loss.item())
    # ---- oracle pipeline with autograd ----
    bf = lambda x: x.to(torch.bfloat16).float()  # noqa: E731
    vb = {k: bf(v) for k, v in vsd.items()}
    lb = {k: bf(v) for k, v in lsd.items()}
    with torch.no_grad():
        hs = T.clip_vit_hidden_states(vb, img, heads=8, n_layers=11, emulate=True)
        img_feat, lv = T.select_spi_levels(hs + [hs[-1]], -2, 4)
        proj = bf(bf(img_feat) @ bf(pw).t() + bf(pb))
        emb = bf(lsd["model.embed_tokens.weight"])[prompt]
    spi = orc(lv, boxes, emulate=True)
    spliced = S.splice(prompt, emb, proj, spi, ids.im_start_token, ids.im_end_token, ids.bbox_token)
    h, _ = T.llama_forward(lb, spliced, heads=4, emulate=True)
    logits = T.lm_logits(lb, h, emulate=True)
    ref_loss = F.cross_entropy(logits[:, :-1].reshape(-1, ids.vocab), labels[:, 1:].reshape(-1), ignore_index=-100)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 3e-2 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
    ref = {f"spi_module.{k}": v.grad for k, v in orc.named_parameters()}
    assert set(grads) == set(ref)
    coss = {k: cosine(grads[k], ref[k]) for k in ref}
    errs = {k: relerr(grads[k], ref[k]) for k in ref}
    print("step gradients (cos, max-norm err):", sorted((round(coss[k], 4), round(errs[k], 3), k) for k in ref)[:6])
    assert min(coss.values()) > 0.97 and max(errs.values()) < 0.3, (coss, errs)
    # ---- clip + AdamW against torch on the same gradients ----
    lr = 2e-6                                                       # keeps the first Adam step in the linear regime
    names = list(tr.params)
    before = {k: tr.params[k].detach().cpu().clone() for k in names}
    cpu_params = [torch.nn.Parameter(before[k].clone()) for k in names]
    for p, k in zip(cpu_params, names):
        p.grad = grads[k].float().cpu().clone()
    torch.nn.utils.clip_grad_norm_(cpu_params, 1.0)
    opt = torch.optim.AdamW(cpu_params, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    opt.step()
    tr.apply(grads, lr=lr)
    predicted = 0.0
    for p, k in zip(cpu_params, names):
        assert relerr(tr.params[k].detach(), p.detach()) < 1e-4, k
        assert not torch.equal(tr.params[k].detach().cpu(), before[k]), k
        predicted += float(((p.detach() - before[k]).double() * grads[k].double().cpu()).sum())
    # ---- the update moves the loss by what the gradient predicts (a directional derivative of the whole step) ----
    l1 = tr.step(dev(prompt), dev(img), [dev(b) for b in boxes], dev(labels), lr=cosine_lr(50, 100, 2 * lr))
    l2 = tr.step(dev(prompt), dev(img), [dev(b) for b in boxes], dev(labels), lr=lr)
    print("loss over three steps:", loss.item(), l1.item(), l2.item(), "first-order prediction of step 1:", predicted)
    assert predicted < -0.02
    assert 0.5 * predicted > l1.item() - loss.item() > 1.5 * predicted, (l1.item() - loss.item(), predicted)
    assert l2.item() < l1.item() < loss.item()


def test_stage2_training_step_end_to_end():
    """train.FullTrainer (everything but the ViT trainable): decoder / embedding / projector gradients against autograd
    through the chained oracles, master -> bf16 -> W^T refresh after the update, and the loss follows <grad, dw>."""
    from gpt4roi_amd.spi_llava import SPILlavaLlamaModel
    from gpt4roi_amd.train import FullTrainer
    from gpt4roi_amd.vit import ClipVisionTower
    H, P, image = 512, 8, 112
    ids = syn.token_ids(vocab_base=990)
    vsd = syn.vit_state(H, 4 * H, 12, image, seed=8)
    lsd = syn.llama_state(512, 1408, 2, ids.vocab, seed=9)
    tower = ClipVisionTower(vsd, heads=8, device=DEV)
    dec = LlamaDecoder(lsd, heads=4, max_positions=256, device=DEV)
    hf = dec.export_hf_state_dict()
    for k, v in lsd.items():                                        # the kernel layouts round-trip to the HF names
        assert torch.equal(hf[k].float().cpu(), v.to(torch.bfloat16).float() if v.dim() > 1 else v.float()), k
    model = SPILlavaLlamaModel(tower, dec, ids, embed_dims=H)
    orc = S.MLVLROIQueryOracle(embed_dims=H, P=P)
    orc.roi_align.updims = torch.nn.Linear(1024, 512)
    spi_sd = S.synthetic_state(orc, 10)
    orc.load_state_dict(spi_sd)
    model.spi_module.load_state_dict(spi_sd)
    g = torch.Generator().manual_seed(11)
    pw, pb = torch.randn(512, H, generator=g) / H ** 0.5, torch.randn(512, generator=g) * 0.05
    with torch.no_grad():
        model.mm_projector.weight.copy_(pw)
        model.mm_projector.bias.copy_(pb)
    img = torch.randn(1, 3, image, image, generator=g)
    boxes = [syn.boxes(3, g)]
    prompt = syn.prompt_ids(ids, P, 3, g, sys_len=6, question_len=9, vocab_base=990)[None]
    labels = prompt.clone()
    labels[:, :8 + P * P] = -100
    labels[labels >= 990] = -100
    lr = 2e-6
    tr = FullTrainer(model, lr=lr, max_grad_norm=1.0)
    dev = lambda t: t.to(DEV)  # noqa: E731
    args = (dev(prompt), dev(img), [dev(b) for b in boxes], dev(labels))
    loss, grads = tr.loss_and_grads(*args)
    # ---- oracle: decoder, embedding and projector as autograd leaves ----
    bf = lambda x: x.to(torch.bfloat16).float()  # noqa: E731
    vb = {k: bf(v) for k, v in vsd.items()}
    lb = {k: bf(v).requires_grad_(True) for k, v in lsd.items()}
    pwr, pbr = bf(pw).requires_grad_(True), bf(pb).requires_grad_(True)
    with torch.no_grad():
        hs = T.clip_vit_hidden_states(vb, img, heads=8, n_layers=11, emulate=True)
        img_feat, lv = T.select_spi_levels(hs + [hs[-1]], -2, 4)
    proj = S._r(S._r(img_feat, True) @ pwr.t() + pbr, True)
    emb = lb["model.embed_tokens.weight"][prompt]
    spi = orc(lv, boxes, emulate=True)
    spliced = S.splice(prompt, emb, proj, spi, ids.im_start_token, ids.im_end_token, ids.bbox_token)
    h, _ = T.llama_forward(lb, spliced, heads=4, emulate=True)
    logits = T.lm_logits(lb, h, emulate=True)
    ref_loss = F.cross_entropy(logits[:, :-1].reshape(-1, ids.vocab), labels[:, 1:].reshape(-1), ignore_index=-100)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 3e-2 * abs(ref_loss.item())
    p0 = "model.layers.0."
    ref = {"llama.embed_tokens": lb["model.embed_tokens.weight"].grad, "llama.lm_head": lb["lm_head.weight"].grad,
           "llama.norm": lb["model.norm.weight"].grad,
           "llama.0.wqkv": torch.cat([lb[p0 + f"self_attn.{n}_proj.weight"].grad for n in "qkv"], 0),
           "llama.0.wo": lb[p0 + "self_attn.o_proj.weight"].grad,
           "llama.0.wgu": torch.stack([lb[p0 + "mlp.gate_proj.weight"].grad, lb[p0 + "mlp.up_proj.weight"].grad], 1).reshape(-1, 512),
           "llama.1.wd": lb["model.layers.1.mlp.down_proj.weight"].grad,
           "llama.1.n1": lb["model.layers.1.input_layernorm.weight"].grad,
           "mm_projector.weight": pwr.grad, "mm_projector.bias": pbr.grad}
    coss = {k: round(cosine(grads[k], v), 4) for k, v in ref.items()}
    print("stage-2 gradient cosines:", coss)
    assert min(coss.values()) > 0.97, coss
    assert set(grads) == set(tr.params) | set(tr.dec_master)
    # ---- update: masters, bf16 copies and transposes stay consistent; the loss follows the gradient ----
    before = {k: v.clone() for k, v in tr.dec_master.items()}
    tr.apply(grads, lr=lr)
    predicted = sum(float(((tr.dec_master[k] - before[k]).double() * grads[k].reshape(before[k].shape).double()).sum())
                    for k in before)
    L0 = dec.layers[0]
    assert torch.equal(L0["wo"], tr.dec_master["llama.0.wo"].to(torch.bfloat16))
    assert torch.equal(L0["wo_t"], L0["wo"].t().contiguous())
    assert not torch.equal(tr.dec_master["llama.0.wo"], before["llama.0.wo"])
    l1 = tr.step(*args, lr=lr)
    print("stage-2 loss:", loss.item(), l1.item(), "predicted change from the decoder part alone:", predicted)
    assert predicted < 0 and l1.item() < loss.item()


def _tiny_stage2(seed=8):
    from gpt4roi_amd.spi_llava import SPILlavaLlamaModel
    from gpt4roi_amd.vit import ClipVisionTower
    H, P, image = 512, 8, 112
    ids = syn.token_ids(vocab_base=990)
    tower = ClipVisionTower(syn.vit_state(H, 4 * H, 12, image, seed=seed), heads=8, device=DEV)
    dec = LlamaDecoder(syn.llama_state(512, 1408, 2, ids.vocab, seed=seed + 1), heads=4, max_positions=256, device=DEV)
    model = SPILlavaLlamaModel(tower, dec, ids, embed_dims=H)
    orc = S.MLVLROIQueryOracle(embed_dims=H, P=P)
    orc.roi_align.updims = torch.nn.Linear(1024, 512)
    model.spi_module.load_state_dict(S.synthetic_state(orc, seed + 2))
    g = torch.Generator().manual_seed(seed + 3)
    with torch.no_grad():
        model.mm_projector.weight.copy_(torch.randn(512, H, generator=g) / H ** 0.5)
        model.mm_projector.bias.copy_(torch.randn(512, generator=g) * 0.05)
    img = torch.randn(2, 3, image, image, generator=g)
    boxes = [syn.boxes(3, g), syn.boxes(3, g)]
    prompt = torch.stack([syn.prompt_ids(ids, P, 3, g, sys_len=6, question_len=9, vocab_base=990) for _ in range(2)])
    labels = prompt.clone()
    labels[:, :8 + P * P] = -100
    labels[labels >= 990] = -100
    return model, (prompt.to(DEV), img.to(DEV), [b.to(DEV) for b in boxes], labels.to(DEV))


def test_sharded_stage2_trainer_equals_unsharded_on_one_rank():
    """train.ShardedFullTrainer at world size 1 (the shard is the whole bucket): same losses and the same weights as
    FullTrainer over three steps -- the flat parameter buckets, the rebinding of every live tensor into them, the fused
    clip + AdamW on the bucket slices and the W^T / kernel-copy refresh.  The two-rank exchange itself (reduce-scatter,
    global norm, in-place all-gather) is covered on CPU by tests/test_sharded_gloo.py."""
    from gpt4roi_amd.train import FullTrainer, ShardedFullTrainer
    lr = 5e-5
    ma, args = _tiny_stage2()
    mb, _ = _tiny_stage2()
    ta = FullTrainer(ma, lr=lr, max_grad_norm=1.0)
    tb = ShardedFullTrainer(mb, lr=lr, max_grad_norm=1.0, bucket_bytes=1 << 20)
    assert len(tb.sharded.buckets) >= 4
    own, full = tb.sharded.state_bytes()
    assert own == full                                              # one rank owns everything
    # the decoder reads views of the flat parameter buckets now
    b0, i0 = tb.sharded.where["llama.0.wqkv"]
    assert mb.llama.layers[0]["wqkv"].data_ptr() == b0.pviews[i0].data_ptr()
    assert mb.mm_projector.weight.data_ptr() == tb.sharded.where["mm_projector.weight"][0].pviews[
        tb.sharded.where["mm_projector.weight"][1]].data_ptr()
    la, lb = [], []
    for _ in range(3):
        la.append(ta.step(*args).item())
        lb.append(tb.step(*args).item())
    print("losses unsharded:", la, "sharded:", lb, "grad norms:", ta.last_grad_norm.item(), tb.last_grad_norm.item())
    for a, b in zip(la, lb):
        assert abs(a - b) < 2e-3 * abs(a)                           # (atomics order in a few backward kernels)
    assert la[-1] < la[0]
    assert abs(ta.last_grad_norm.item() - tb.last_grad_norm.item()) < 1e-2 * ta.last_grad_norm.item()
    wa, wb = ma.llama.export_hf_state_dict(), mb.llama.export_hf_state_dict()
    for k in wa:
        d = (wa[k].float() - wb[k].float()).abs().max().item()
        assert d <= 2 ** -7 * wa[k].float().abs().max().item() + 1e-6, (k, d)
    for (k, pa), (_, pb) in zip(ma.spi_module.named_parameters(), mb.spi_module.named_parameters()):
        d = (pa - pb).abs()                                         # Adam turns atomics-order noise on ~0 gradients into
        assert d.max().item() <= 2 * 3 * lr and d.mean().item() <= 0.05 * 3 * lr, k   # +-lr steps: bound by the step size
    # W^T was refreshed from the gathered weights
    L0 = mb.llama.layers[0]
    assert torch.equal(L0["wo_t"], L0["wo"].t().contiguous())
    # sharded state round trip
    sd = tb.state_dict()
    tb.load_state_dict(sd)
    assert tb.steps == 3 and len(sd["buckets"]) == len(tb.sharded.buckets)


def test_fsdp_stage2_trainer_equals_unsharded_on_one_rank():
    """train.FSDPFullTrainer (gpt4roi_amd/fsdp.py: parameters + gradients + optimizer state sharded per decoder layer, the
    strategy of train_stage2.sh:51-52) at world size 1: the same losses and weights as FullTrainer over three steps -- the
    per-layer gather / release around the forward and the backward (the live tensors are views of pool buffers only while
    their layer runs, None otherwise), the per-layer W^T scratch, the transient full-gradient buffers and their hand-off to
    the owned slices, the fused clip + AdamW on the slices.  The two-rank collectives run over gloo in tests/test_fsdp_gloo.py."""
    from gpt4roi_amd.train import FSDPFullTrainer, FullTrainer
    lr = 5e-5
    ma, args = _tiny_stage2()
    mb, _ = _tiny_stage2()
    ta = FullTrainer(ma, lr=lr, max_grad_norm=1.0)
    tb = FSDPFullTrainer(mb, lr=lr, max_grad_norm=1.0)
    n_layers = len(mb.llama.layers)
    assert len(tb.fsdp.units) == 1 + n_layers
    # nothing but the shards persists: the live tensors are released
    assert mb.llama.layers[0]["wqkv"] is None and mb.llama.lm_head is None and mb.mm_projector.weight.numel() == 0
    la, lb = [], []
    for _ in range(3):
        la.append(ta.step(*args).item())
        lb.append(tb.step(*args).item())
        assert mb.llama.layers[0]["wo"] is None and "wo_t" not in mb.llama.layers[0]          # released again after the step
    own, full, transient = tb.fsdp.memory()
    print("losses unsharded:", la, "fsdp:", lb, "grad norms:", ta.last_grad_norm.item(), tb.last_grad_norm.item(),
          "persistent bytes", own, "peak transient pool bytes", transient)
    assert own == full                                              # one rank owns every slice
    # the pool never held more than the root + (prefetch + 1) layers of parameters + one unit of gradients
    per_layer = sum(f.padded * 2 for f in tb.fsdp.units[1]["flats"] if f.dtype == torch.bfloat16) + \
        sum(f.padded * 4 for f in tb.fsdp.units[1]["flats"] if f.dtype == torch.float32)
    root = sum(f.padded * (2 if f.dtype == torch.bfloat16 else 4) for f in tb.fsdp.units[0]["flats"])
    root_g = sum(f.padded * 4 for f in tb.fsdp.units[0]["flats"])
    assert transient <= root + root_g + 2 * per_layer + 2 * per_layer * 2 + 4096, (transient, root, per_layer)
    for a, b in zip(la, lb):
        assert abs(a - b) < 2e-3 * abs(a)
    assert la[-1] < la[0]
    assert abs(ta.last_grad_norm.item() - tb.last_grad_norm.item()) < 1e-2 * ta.last_grad_norm.item()
    full_b = tb.full_state_dict()
    live_a = ma.llama.trainable_tensors()
    for k, va in live_a.items():
        vb = full_b[f"llama.{k}"]
        d = (va.float() - vb.float()).abs().max().item()
        assert d <= 2 ** -7 * va.float().abs().max().item() + 1e-6, (k, d)
    for k, pa in ma.spi_module.named_parameters():
        d = (pa - full_b[f"spi_module.{k}"]).abs()
        assert d.max().item() <= 2 * 3 * lr and d.mean().item() <= 0.05 * 3 * lr, k


def test_fsdp_stage2_trainer_checkpoint_save_reload_continue(tmp_path):
    """FSDPFullTrainer writes the reference-format checkpoint and resumes.  (a) state_dict() -> a FRESH trainer ->
    load_state_dict(): the next two steps are bit-identical to the uninterrupted run (parameter shards, masters, moments,
    step count); (b) export_hf_state_dict() / save_pretrained(): HF-named FULL tensors (q|k|v and gate/up de-fused) equal to
    the unsharded FullTrainer's export after the same steps, and the directory loads back through from_pretrained's reader
    (gpt4roi/train/train.py:86-95 is what the reference's stage 2 leaves behind); (c) summon_full_params(): the model runs a
    forward between two steps and is released again afterwards."""
    import copy

    from gpt4roi_amd import checkpoint as ckpt
    from gpt4roi_amd.train import FSDPFullTrainer, FullTrainer
    lr = 5e-5
    ma, args = _tiny_stage2()
    mb, _ = _tiny_stage2()
    mc, _ = _tiny_stage2()
    ta = FullTrainer(ma, lr=lr, max_grad_norm=1.0)
    tb = FSDPFullTrainer(mb, lr=lr, max_grad_norm=1.0)
    for _ in range(2):
        ta.step(*args)
        tb.step(*args)
    sd = copy.deepcopy(tb.state_dict())
    assert sd["step"] == 2 and len(sd["units"]) == 1 + len(mb.llama.layers)
    # (b) the reference-format export against the unsharded trainer
    hf_b = tb.export_hf_state_dict()
    hf_a = ma.llama.export_hf_state_dict()
    assert set(hf_a) <= set(hf_b) and any(k.startswith("model.spi_module.") for k in hf_b) and "model.mm_projector.weight" in hf_b
    for k, va in hf_a.items():
        assert hf_b[k].shape == va.shape, k
        d = (va.float() - hf_b[k].float()).abs().max().item()
        assert d <= 2 ** -7 * va.float().abs().max().item() + 1e-6, (k, d)
    tb.save_pretrained(str(tmp_path / "ckpt"))
    back = ckpt.load_hf_state_dict(str(tmp_path / "ckpt"))
    for k, v in hf_b.items():
        assert torch.equal(back[k].to(v.device).to(v.dtype), v), k
    # (c) a forward between two steps
    with tb.summon_full_params() as m:
        assert m.llama.layers[0]["wqkv"] is not None and m.llama.lm_head is not None
        with torch.no_grad():
            logits, _ = m.forward_train(args[0], args[1], args[2])
        assert torch.isfinite(logits.float()).all()
    assert mb.llama.layers[0]["wqkv"] is None and mb.llama.lm_head is None
    # (a) resume in a fresh trainer == the uninterrupted run
    lb = [tb.step(*args).item() for _ in range(2)]
    tc = FSDPFullTrainer(mc, lr=lr, max_grad_norm=1.0)
    tc.load_state_dict(sd)
    assert tc.steps == 2
    lc = [tc.step(*args).item() for _ in range(2)]
    print("continued:", lb, "resumed:", lc)
    fb, fc = tb.full_state_dict(), tc.full_state_dict()
    for a, b in zip(lb, lc):
        assert abs(a - b) < 2e-3 * abs(a)                           # (atomics order in a few backward kernels)
    for k in fb:
        d = (fb[k].float() - fc[k].float()).abs().max().item()
        assert d <= 2 ** -7 * fb[k].float().abs().max().item() + 4 * lr, (k, d)


def test_sharded_stage2_trainer_steps_on_a_region_less_batch():
    """The same region-less batch through ShardedFullTrainer: every bucket's reduce-scatter must be fed -- a rank
    that reported no `spi_module.*` gradient would raise in `_wait()` while the other ranks block in the collective.  The step
    runs, the region module's masters do not move, the decoder's do."""
    from gpt4roi_amd.train import ShardedFullTrainer
    model, (_, img, _, _) = _tiny_stage2(seed=21)
    ids = syn.token_ids(vocab_base=990)
    g = torch.Generator().manual_seed(5)
    P = 8
    prompt = torch.stack([syn.prompt_ids(ids, P, 0, g, sys_len=6, question_len=9, vocab_base=990) for _ in range(2)]).to(DEV)
    labels = prompt.clone()
    labels[:, :8 + P * P] = -100
    labels[labels >= 990] = -100
    none = [torch.zeros(0, 4, device=DEV), torch.zeros(0, 4, device=DEV)]
    tr = ShardedFullTrainer(model, lr=1e-4, max_grad_norm=1.0, bucket_bytes=1 << 20)
    spi_before = {k: p.detach().clone() for k, p in model.spi_module.named_parameters()}
    wo_before = model.llama.layers[0]["wo"].clone()
    loss = tr.step(prompt, img, none, labels)
    assert torch.isfinite(loss).all() and tr.steps == 1
    for k, p in model.spi_module.named_parameters():
        assert float((p.detach() - spi_before[k]).abs().max()) == 0.0, k
    assert not torch.equal(model.llama.layers[0]["wo"], wo_before)
    loss2 = tr.step(prompt, img, none, labels)
    assert torch.isfinite(loss2).all() and float(loss2) < float(loss) + 1e-2


def test_region_less_batch_steps_with_zero_region_gradients():
    """A batch without any region (text-only / image-only sample): the reference keeps training through a zero dummy term
    (gpt4roi/models/layers.py:314-317, spi_llava.py:94-108).  RegionTrainer must report a ZERO gradient for every region-
    module parameter -- through `on_grad` too, which is what the bucketed exchange waits for on every rank -- and the step
    must leave the region module untouched while the projector trains."""
    from gpt4roi_amd.spi_llava import SPILlavaLlamaModel
    from gpt4roi_amd.train import RegionTrainer
    from gpt4roi_amd.vit import ClipVisionTower
    H, P, image = 512, 8, 112
    ids = syn.token_ids(vocab_base=990)
    tower = ClipVisionTower(syn.vit_state(H, 4 * H, 12, image, seed=8), heads=8, device=DEV)
    dec = LlamaDecoder(syn.llama_state(512, 1408, 2, ids.vocab, seed=9), heads=4, max_positions=256, device=DEV)
    model = SPILlavaLlamaModel(tower, dec, ids, embed_dims=H)
    model.spi_module.load_state_dict(syn.spi_state(model.spi_module, 5))
    g = torch.Generator().manual_seed(12)
    img = torch.randn(1, 3, image, image, generator=g).to(DEV)
    prompt = syn.prompt_ids(ids, P, 0, g, sys_len=6, question_len=9, vocab_base=990)[None].to(DEV)
    labels = prompt.clone()
    labels[:, :8 + P * P] = -100
    labels[labels >= 990] = -100
    tr = RegionTrainer(model, lr=1e-3, max_grad_norm=1.0, train_projector=True)
    before = {k: p.detach().clone() for k, p in tr.params.items()}
    seen = []
    m = tr.model
    logits, ctx = m.forward_train(prompt, img, [torch.zeros(0, 4, device=DEV)])
    loss, dlogits = m.llama.loss_and_dlogits(logits, labels)
    grads = m.backward(ctx, dlogits, train_projector=True, on_grad=lambda n, t: seen.append(n))
    assert set(grads) == set(tr.params) and set(seen) == set(tr.params)            # every registered tensor is reported
    assert all(float(grads[k].abs().max()) == 0.0 and grads[k].numel() == tr.params[k].numel()
               for k in grads if k.startswith("spi_module."))
    assert float(grads["mm_projector.weight"].abs().max()) > 0
    loss2 = tr.step(prompt, img, [torch.zeros(0, 4, device=DEV)], labels)
    assert torch.isfinite(loss2).all() and abs(float(loss2) - float(loss)) < 1e-3
    for k, p in tr.params.items():
        moved = float((p.detach() - before[k]).abs().max())
        assert (moved == 0.0) if k.startswith("spi_module.") else (moved > 0.0), (k, moved)
