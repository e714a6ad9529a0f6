"""GPU: the backward / optimizer kernels of the training rows (include/g4r_train.h) against torch autograd on
the fp32 statement of the same op, and the LLaMA stack's input gradient against autograd through the CPU
oracle (oracle/transformer_oracle.py).  Everything goes through the C ABI."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import transformer_oracle as T  # noqa: E402

if torch.cuda.is_available():
    from gpt4roi_amd import kernels as K
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.llama import LlamaDecoder

DEV = "cuda"


def rnd(*shape, scale=1.0, seed=0, dtype=torch.bfloat16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def relerr(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-9)).item()


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
