"""GPU: batches that carry a key padding mask (left / right padding, holes) through the decoder.

The reference's training attention unpads the batch with `attention_mask`, runs varlen flash attention over the kept
tokens and pads the result back (llava/train/llama_flash_attn_monkey_patch.py:60-85); its serving path hands the mask to
HF's LlamaModel (llava/model/llava.py:263-283).  The oracle (oracle/transformer_oracle.py, `key_padding_mask`) is pinned
to HF on the kept rows in tests/test_oracle_transformers.py; here the HIP path -- packed rows, per-sequence attention
launches, the ragged decode launch -- is compared with it: prefill logits, last-position selection, batched greedy decode
(host loop and hipGraph), the training loss and gradients.  Everything goes through the C ABI."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import transformer_oracle as T  # noqa: E402

if torch.cuda.is_available():
    from gpt4roi_amd import kernels as K
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.llama import LlamaDecoder, RaggedLayout

DEV = "cuda"
H16 = {"bf16": torch.bfloat16, "fp16": torch.float16}


def relerr(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-9)).item()


def _mask(kind, B, T_):
    m = torch.ones(B, T_, dtype=torch.bool)
    if kind == "left":
        m[0, :T_ // 3] = False
        m[2, :5] = False
    elif kind == "right":
        m[0, T_ - 9:] = False
        m[1, T_ - 1:] = False
    elif kind == "holes":
        m[0, [0, 3, 4, 17, T_ - 1]] = False
        m[1, 7:19] = False
        m[2, :2] = False
        m[2, T_ - 4:] = False
    return m


def _decoder(layers=3, dtype=torch.bfloat16, hidden=256, heads=4, inter=704, vocab=1000, B=3, seed=5):
    sd = syn.llama_state(hidden, inter, layers, vocab, seed=seed)
    dec = LlamaDecoder(sd, heads=heads, max_positions=128, device=DEV, max_batch=B, dtype=dtype)
    sdr = {k: v.to(dtype).float() for k, v in sd.items()}
    return sd, sdr, dec


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_attn_decode_ragged_launch_vs_oracle(dt):
    """g4r_attn_decode_ragged_*: B sequences of different cached lengths, the new token rotated at one shared position
    that is NOT its cache row, appended at the sequence's own length, attention over its own rows only."""
    dtype = H16[dt]
    B, H, D, Tmax = 4, 4, 64, 96
    C = H * D
    g = torch.Generator().manual_seed(3)
    lens = [5, 64, 17, 0]
    rpos = 70
    kc = (torch.randn(B, Tmax, C, generator=g)).to(dtype).to(DEV)
    vc = (torch.randn(B, Tmax, C, generator=g)).to(dtype).to(DEV)
    qkv = torch.randn(B, 3 * C, generator=g).to(dtype).to(DEV)
    cos, sin = (t.to(DEV).contiguous() for t in T.rope_tables(Tmax, D))
    k0, v0 = kc.clone(), vc.clone()
    work = K.DecodeAttnWorkspace(H, D, DEV, splits=4, batch=B)
    pos = torch.tensor(lens + [rpos], dtype=torch.int32, device=DEV)
    out = K.attn_decode(None, kc, vc, H, 1.0 / math.sqrt(D), work, qkv=qkv, cos=cos, sin=sin, kv_lens_dev=pos[:B],
                        rope_pos_dev=pos[B:])
    torch.cuda.synchronize()
    cs, sn = T.rope_tables(Tmax, D)
    for b in range(B):
        q, k, v = (qkv[b].float().cpu()[i * C:(i + 1) * C].view(1, 1, C) for i in range(3))
        qr = T._r(T.apply_rope(q, cs, sn, H, rpos), dtype)
        kr = T._r(T.apply_rope(k, cs, sn, H, rpos), dtype)
        n = lens[b]
        assert torch.equal(kc[b, n].float().cpu(), kr.view(-1)) and torch.equal(vc[b, n].float().cpu(), v.view(-1))
        keep = torch.ones(Tmax, dtype=torch.bool)
        keep[n] = False
        assert torch.equal(kc[b][keep], k0[b][keep]) and torch.equal(vc[b][keep], v0[b][keep])   # nothing else written
        kk = torch.cat([k0[b, :n].float().cpu(), kr.view(1, C)], 0)[None]
        vv = torch.cat([v0[b, :n].float().cpu(), v.view(1, C)], 0)[None]
        want = T.attention(qr, kk, vv, H, 1.0 / math.sqrt(D), True, emulate=dtype)
        e = relerr(out[b], want.view(-1))
        assert e < (2e-2 if dt == "bf16" else 3e-3), (b, e)
    assert int(work.cnt.abs().sum()) == 0


@pytest.mark.parametrize("kind", ["left", "right", "holes"])
@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_prefill_with_a_padding_mask_vs_oracle(kind, dt):
    """Logits of the kept rows == the oracle's with the same mask (which is HF's with attention_mask); the last-position
    form returns each sequence's last KEPT row; the cache holds each sequence's kept K/V rows squeezed to the front."""
    dtype = H16[dt]
    B, T_ = 3, 45
    sd, sdr, dec = _decoder(dtype=dtype)
    mask = _mask(kind, B, T_)
    g = torch.Generator().manual_seed(11)
    emb = sdr["model.embed_tokens.weight"][torch.randint(0, 1000, (B, T_), generator=g)]
    dec.reset(B)
    logits = dec.forward(emb.to(DEV).to(dtype), key_padding_mask=mask.to(DEV))
    assert dec.ragged is not None and dec.ragged.lens == mask.sum(1).tolist()
    h, cache = T.llama_forward(sdr, emb, heads=4, emulate=dtype, key_padding_mask=mask)
    want = T.lm_logits(sdr, h, emulate=dtype)
    e = relerr(logits.cpu()[mask], want[mask])
    print(f"masked prefill [{kind}, {dt}] logits vs the emulating oracle: {e:.3e}")
    assert e < (2.5e-2 if dt == "bf16" else 4e-3), e
    # an unmasked batch of the same shape is a different result on the rows that lost keys (the mask is not ignored)
    dec.reset(B)
    dense = dec.forward(emb.to(DEV).to(dtype))
    assert dec.ragged is None
    if kind != "right":
        assert relerr(dense.cpu()[mask], want[mask]) > 5 * e
    # compacted cache: sequence b's rows [0, n_b) are its kept keys in order (the oracle keeps them at the padded rows)
    dec.reset(B)
    last = dec.forward(emb.to(DEV).to(dtype), all_logits=False, key_padding_mask=mask.to(DEV))
    for b in range(B):
        n = int(mask[b].sum())
        for li in (0, len(dec.layers) - 1):
            ek = relerr(dec.kc[li, b, :n], cache[li][0][b][mask[b]])
            ev = relerr(dec.vc[li, b, :n], cache[li][1][b][mask[b]])
            assert ek < 3e-2 and ev < 3e-2, (b, li, ek, ev)
        tl = int(torch.nonzero(mask[b]).max())
        assert torch.equal(last[b, 0], logits[b, tl]), (b, tl)


@pytest.mark.parametrize("kind", ["left", "right"])
def test_ragged_batch_decode_continues_every_sequence_at_its_own_length(kind):
    """greedy_batch / decode_graph_batch under a mask: (i) host loop == hipGraph replay; (ii) teacher-forced against the
    emulating oracle run with the mask grown by ones (HF's decode): every emitted id is the oracle's argmax up to the
    storage-rounding noise; (iii) a right-padded batch emits what each prompt emits alone (same positions, same keys)."""
    B, T_, n_new = 3, 33, 7
    sd, sdr, dec = _decoder(layers=2)
    mask = _mask(kind, B, T_)
    g = torch.Generator().manual_seed(21)
    emb_w = sdr["model.embed_tokens.weight"]
    emb = emb_w[torch.randint(0, 1000, (B, T_), generator=g)]
    e_dev = emb.to(DEV).to(torch.bfloat16)
    got = dec.greedy_batch(e_dev, n_new, key_padding_mask=mask.to(DEV))
    assert dec.pos == T_ + n_new - 1 and dec._rag_pos.tolist() == [int(n) + n_new - 1 for n in mask.sum(1)] + [T_ + n_new - 1]
    graph = dec.decode_graph_batch(e_dev, n_new, key_padding_mask=mask.to(DEV))
    assert graph == got
    assert dec.decode_graph_batch(e_dev, n_new, key_padding_mask=mask.to(DEV)) == got       # captured graph reused
    other = _mask("left" if kind == "right" else "right", B, T_)
    assert dec.decode_graph_batch(e_dev, 4, key_padding_mask=other.to(DEV)) == dec.greedy_batch(
        e_dev, 4, key_padding_mask=other.to(DEV))                                          # same graph, other lengths
    # (ii) oracle, teacher-forced with the ids the HIP path chose
    h, cache = T.llama_forward(sdr, emb, heads=4, emulate=True, key_padding_mask=mask)
    lastrow = torch.tensor([int(torch.nonzero(mask[b]).max()) for b in range(B)])
    hl = h[torch.arange(B), lastrow][:, None]
    m, pos, same = mask, T_, 0
    for s in range(n_new):
        lg = T.lm_logits(sdr, hl, emulate=True)[:, 0]
        ids = torch.tensor([got[b][s] for b in range(B)])
        top = lg.max(-1).values
        chosen = lg[torch.arange(B), ids]
        assert float((top - chosen).max()) <= 2e-2 * float(lg.abs().max()), (s, (top - chosen).tolist())
        same += int((lg.argmax(-1) == ids).sum())
        m = torch.cat([m, torch.ones(B, 1, dtype=torch.bool)], 1)
        hl, cache = T.llama_forward(sdr, emb_w[ids][:, None], heads=4, emulate=True, kv_cache=cache, pos0=pos,
                                    key_padding_mask=m)
        pos += 1
    assert same >= int(0.85 * B * n_new), same
    # (iii) right padding: each prompt alone, at the same positions
    if kind == "right":
        agree = 0
        for b in range(B):
            n = int(mask[b].sum())
            alone = dec.greedy(e_dev[b:b + 1, :n], n_new)
            agree += sum(int(x == y) for x, y in zip(alone, got[b]))
        assert agree >= int(0.85 * B * n_new), agree


def test_generate_batched_with_left_padding():
    """SPILlavaMPTForCausalLM.generate for B > 1 (HF-style left-padded batch): same new ids as the decoder-level call,
    sequences returned as [B, T + n] with pad after a stop id."""
    from types import SimpleNamespace

    from gpt4roi_amd.spi_llava import SPILlavaLlamaModel, SPILlavaMPTForCausalLM
    B, T_, n_new = 2, 20, 5
    sd, sdr, dec = _decoder(layers=2, B=B)
    ids = SimpleNamespace(im_patch_token=996, bbox_token=997, im_start_token=998, im_end_token=999, point_token=995)
    model = SPILlavaLlamaModel(None, dec, ids, embed_dims=512)
    lm = SPILlavaMPTForCausalLM(model, SimpleNamespace(eos_token_id=None, pad_token_id=0))
    g = torch.Generator().manual_seed(31)
    input_ids = torch.randint(1, 900, (B, T_), generator=g).to(DEV)
    mask = torch.ones(B, T_, dtype=torch.long, device=DEV)
    mask[1, :6] = 0
    input_ids[1, :6] = 0
    out = lm.generate(input_ids=input_ids, attention_mask=mask, max_new_tokens=n_new, do_sample=False)
    emb = K.gather_rows(dec.embed, input_ids.reshape(-1).to(torch.int32)).view(B, T_, -1)
    want = dec.greedy_batch(emb, n_new, key_padding_mask=mask.bool())
    assert out.shape == (B, T_ + n_new) and out[:, T_:].tolist() == want
    stop = want[0][1]
    cut = lm.generate(input_ids=input_ids, attention_mask=mask, max_new_tokens=n_new, do_sample=False, stop_ids=(stop,),
                      pad_token_id=7)
    k = want[0].index(stop)
    assert cut[0, T_:T_ + k + 1].tolist() == want[0][:k + 1] and all(int(t) == 7 for t in cut[0, T_ + k + 1:])


@pytest.mark.parametrize("kind,ckpt", [("left", False), ("holes", False), ("holes", True), ("right", False)])
def test_training_step_with_a_padding_mask_vs_oracle_autograd(kind, ckpt):
    """forward_train / backward with key_padding_mask == autograd through the fp32 oracle with the same mask (the
    reference's unpad -> varlen flash attention -> pad_input): loss, d(inputs_embeds) and every weight gradient.  Labels
    are -100 on the masked positions, as the collator writes them (data_modules.py:22-56)."""
    hidden, inter, layers, vocab, heads, B, T_ = 256, 384, 2, 515, 2, 3, 48
    sd = syn.llama_state(hidden, inter, layers, vocab, seed=4)
    dec = LlamaDecoder(sd, heads=heads, max_positions=128, device=DEV, max_batch=B)
    dec.prepare_training(train_weights=True)
    mask = _mask(kind, B, T_)
    g = torch.Generator().manual_seed(62)
    emb = (torch.randn(B, T_, hidden, generator=g)).to(torch.bfloat16)
    labels = torch.randint(0, vocab, (B, T_), generator=g)
    labels[~mask] = -100
    logits, ctx = dec.forward_train(emb.to(DEV), checkpoint=ckpt, key_padding_mask=mask.to(DEV))
    assert ctx["rag"] is not None
    loss, dlogits = dec.loss_and_dlogits(logits, labels.to(DEV))
    dx = dec.backward(ctx, dlogits)
    w = {k: v.to(torch.bfloat16).float().requires_grad_(True) for k, v in sd.items()}
    er = emb.float().requires_grad_(True)
    hid, _ = T.llama_forward(w, er, heads, key_padding_mask=mask)
    # the shifted loss reads the logits of position t for label t + 1: a masked position followed by a kept one would carry
    # loss on a zeroed attention row in both implementations -- the same arithmetic, kept in the comparison
    ref_loss = F.cross_entropy(T.lm_logits(w, hid)[:, :-1].reshape(-1, vocab), labels[:, 1:].reshape(-1), ignore_index=-100)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
    e = relerr(dx.view(B, T_, hidden), er.grad)
    assert e < 6e-2, e
    gds = dec.grads
    for li in range(layers):
        p = f"model.layers.{li}."
        ref_qkv = torch.cat([w[p + f"self_attn.{n}_proj.weight"].grad for n in "qkv"], 0)
        ref_gu = torch.stack([w[p + "mlp.gate_proj.weight"].grad, w[p + "mlp.up_proj.weight"].grad], 1).reshape(-1, hidden)
        for name, got, want in (("wd", gds[f"{li}.wd"], w[p + "mlp.down_proj.weight"].grad), ("wgu", gds[f"{li}.wgu"], ref_gu),
                                ("wo", gds[f"{li}.wo"], w[p + "self_attn.o_proj.weight"].grad),
                                ("wqkv", gds[f"{li}.wqkv"], ref_qkv),
                                ("n1", gds[f"{li}.n1"], w[p + "input_layernorm.weight"].grad),
                                ("n2", gds[f"{li}.n2"], w[p + "post_attention_layernorm.weight"].grad)):
            ee = relerr(got, want)
            assert ee < 7e-2, (li, name, ee)
    assert relerr(gds["lm_head"], w["lm_head.weight"].grad) < 6e-2
    if kind == "right":
        # the collator's right padding may also ride the dense kernels: same loss, same gradients
        logits2, ctx2 = dec.forward_train(emb.to(DEV), checkpoint=ckpt)
        loss2, dl2 = dec.loss_and_dlogits(logits2, labels.to(DEV))
        dx2 = dec.backward(ctx2, dl2)
        assert abs(loss2.item() - loss.item()) < 1e-3 * abs(loss.item())
        m3 = mask.reshape(-1)
        assert relerr(dx2[m3.to(DEV)], dx[m3.to(DEV)]) < 2e-2
        assert relerr(dec.grads["0.wqkv"], gds["0.wqkv"]) < 2e-2


def test_ragged_layout_index_sets():
    m = torch.tensor([[0, 0, 1, 1, 1], [1, 1, 1, 1, 1], [1, 0, 1, 0, 0], [0, 0, 0, 0, 0]], device=DEV)
    r = RaggedLayout.of(m, 16)
    assert r.lens == [3, 5, 2, 0] and r.cu == [0, 3, 8, 10, 10]
    assert r.Tc == 5 and r.idx.tolist() == [2, 3, 4, -1, -1, 5, 6, 7, 8, 9, 10, 12, -1, -1, -1] + [-1] * 5
    assert r.idx_cache.tolist() == [2, 3, 4, -1, -1, 16, 17, 18, 19, 20, 32, 34, -1, -1, -1] + [-1] * 5
    assert r.inv.tolist() == [-1, -1, 0, 1, 2, 5, 6, 7, 8, 9, 10, -1, 11] + [-1] * 7
    assert r.last.tolist() == [4, 9, 12, -1]
    assert RaggedLayout.of(torch.ones(2, 3, device=DEV), 16) is None and RaggedLayout.of(None, 16) is None
