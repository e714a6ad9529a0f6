"""CPU: pins the restated CLIP ViT / LLaMA (oracle/transformer_oracle.py) against the container's
HuggingFace transformers with identical random weights (tiny configs, fp32)."""
import pytest
import torch

from oracle import transformer_oracle as T

transformers = pytest.importorskip("transformers")


def test_clip_vit_hidden_states_match_hf():
    from transformers import CLIPVisionConfig, CLIPVisionModel
    torch.manual_seed(0)
    cfg = CLIPVisionConfig(hidden_size=64, intermediate_size=256, num_hidden_layers=5, num_attention_heads=4,
                           image_size=56, patch_size=14, hidden_act="quick_gelu")
    m = CLIPVisionModel(cfg).eval()
    img = torch.randn(2, 3, 56, 56)
    with torch.no_grad():
        want = m(img, output_hidden_states=True).hidden_states
    w = {k: v.float() for k, v in m.state_dict().items()}
    got = T.clip_vit_hidden_states(w, img, heads=4)
    assert len(got) == len(want) == 6
    for a, b in zip(got, want):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4)


def test_level_selection_indices():
    # spi_llava.py:58-82 with mm_vision_select_layer = -2 on 25 hidden states -> [14, 17, 20, 23]
    hs = [torch.full((1, 3, 2), float(i)) for i in range(25)]
    img, lv = T.select_spi_levels(hs, -2, 4)
    assert img[0, 0, 0].item() == 23 and [int(l[0, 0, 0]) for l in lv] == [14, 17, 20, 23]
    assert img.shape[1] == 2  # CLS dropped


def test_llama_matches_hf_prefill_and_cached_decode():
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(1)
    cfg = LlamaConfig(vocab_size=97, hidden_size=64, intermediate_size=176, num_hidden_layers=3,
                      num_attention_heads=4, num_key_value_heads=4, rms_norm_eps=1e-6, max_position_embeddings=64,
                      attention_bias=False, tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    w = {k: v.float() for k, v in m.state_dict().items()}
    ids = torch.randint(0, 97, (2, 11))
    emb = m.get_input_embeddings()(ids)
    with torch.no_grad():
        want = m(inputs_embeds=emb).logits
    h, cache = T.llama_forward(w, emb, heads=4)
    torch.testing.assert_close(T.lm_logits(w, h), want, rtol=1e-4, atol=1e-4)
    # one cached decode step == full forward on the extended sequence
    nxt = torch.randint(0, 97, (2, 1))
    e2 = m.get_input_embeddings()(nxt)
    h2, _ = T.llama_forward(w, e2, heads=4, kv_cache=cache, pos0=11)
    with torch.no_grad():
        want2 = m(inputs_embeds=torch.cat([emb, e2], 1)).logits[:, -1:]
    torch.testing.assert_close(T.lm_logits(w, h2), want2, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("kind", ["left", "right", "holes"])
def test_llama_with_padding_mask_matches_hf_on_the_kept_rows(kind):
    """key_padding_mask of the oracle == HF LlamaModel with attention_mask and the pinned commit's positions
    (arange over the padded layout, whatever the mask): prefill and one cached decode step, compared on the kept rows;
    the masked query rows of the oracle are zeros after attention as pad_input leaves them
    (llama_flash_attn_monkey_patch.py:60-85)."""
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(5)
    cfg = LlamaConfig(vocab_size=97, hidden_size=64, intermediate_size=176, num_hidden_layers=3,
                      num_attention_heads=4, num_key_value_heads=4, rms_norm_eps=1e-6, max_position_embeddings=64,
                      attention_bias=False, tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    w = {k: v.float() for k, v in m.state_dict().items()}
    B, Tn = 3, 12
    mask = torch.ones(B, Tn, dtype=torch.bool)
    if kind == "left":
        mask[0, :5] = False
        mask[2, :2] = False
    elif kind == "right":
        mask[0, 8:] = False
        mask[1, 11:] = False
    else:
        mask[0, [0, 3, 4, 9]] = False
        mask[1, [5]] = False
    emb = m.get_input_embeddings()(torch.randint(0, 97, (B, Tn)))
    pos = torch.arange(Tn)[None].expand(B, Tn)
    with torch.no_grad():
        want = m(inputs_embeds=emb, attention_mask=mask.long(), position_ids=pos).logits
    h, cache = T.llama_forward(w, emb, heads=4, key_padding_mask=mask)
    got = T.lm_logits(w, h)
    torch.testing.assert_close(got[mask], want[mask], rtol=1e-4, atol=1e-4)
    e2 = m.get_input_embeddings()(torch.randint(0, 97, (B, 1)))
    mask2 = torch.cat([mask, torch.ones(B, 1, dtype=torch.bool)], 1)
    h2, _ = T.llama_forward(w, e2, heads=4, kv_cache=cache, pos0=Tn, key_padding_mask=mask2)
    with torch.no_grad():
        want2 = m(inputs_embeds=torch.cat([emb, e2], 1), attention_mask=mask2.long(),
                  position_ids=torch.arange(Tn + 1)[None].expand(B, Tn + 1)).logits[:, -1:]
    torch.testing.assert_close(T.lm_logits(w, h2), want2, rtol=1e-4, atol=1e-4)


def test_greedy_decode_matches_hf_generate():
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(2)
    cfg = LlamaConfig(vocab_size=97, hidden_size=64, intermediate_size=176, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=4, rms_norm_eps=1e-6, max_position_embeddings=64,
                      tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).eval()
    w = {k: v.float() for k, v in m.state_dict().items()}
    ids = torch.randint(3, 97, (1, 7))
    emb = m.get_input_embeddings()(ids)
    with torch.no_grad():
        want = m.generate(inputs_embeds=emb, attention_mask=torch.ones(1, 7, dtype=torch.long), do_sample=False,
                          max_new_tokens=6, min_new_tokens=6, pad_token_id=0)
    got, _ = T.greedy_decode(w, emb, m.get_input_embeddings(), heads=4, n_new=6)
    assert got == want[0].tolist()[-6:]


def test_llama_loss_and_gradients_match_hf_autograd():
    """Training rows: loss (HF's shifted-label CE, labels = -100 ignored) and the gradients w.r.t. inputs_embeds and
    weights from autograd through the restated decoder equal those of HF LlamaForCausalLM(inputs_embeds=, labels=)
    -- the reference's training forward (llava/model/llava.py:203-261 delegates exactly there)."""
    import torch.nn.functional as F
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(3)
    cfg = LlamaConfig(vocab_size=97, hidden_size=64, intermediate_size=176, num_hidden_layers=2,
                      num_attention_heads=4, num_key_value_heads=4, rms_norm_eps=1e-6, max_position_embeddings=64,
                      attention_bias=False, tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg).train()
    emb = torch.randn(2, 11, 64, requires_grad=True)
    labels = torch.randint(0, 97, (2, 11))
    labels[:, :4] = -100
    out = m(inputs_embeds=emb, labels=labels)
    out.loss.backward()
    w = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    emb2 = emb.detach().clone().requires_grad_(True)
    h, _ = T.llama_forward(w, emb2, heads=4)
    logits = T.lm_logits(w, h)
    loss = F.cross_entropy(logits[:, :-1].reshape(-1, 97), labels[:, 1:].reshape(-1), ignore_index=-100)
    loss.backward()
    assert abs(float(loss) - float(out.loss)) < 1e-5
    assert torch.allclose(emb2.grad, emb.grad, atol=1e-6)
    for k in ("model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight",
              "model.layers.0.input_layernorm.weight", "model.norm.weight", "lm_head.weight"):
        assert torch.allclose(w[k].grad, dict(m.named_parameters())[k].grad, atol=1e-6), k
