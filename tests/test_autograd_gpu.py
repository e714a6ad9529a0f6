"""GPU: the drop-in seams under autograd -- what the reference's training caller does
(gpt4roi/train/train.py:698-712: HF Trainer.training_step = forward -> loss.backward() -> optimizer.step()) must run
unchanged against the MI355X classes and give the step gpt4roi_amd/train.py::RegionTrainer gives."""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.layers import MLVLROIQueryModule
    from gpt4roi_amd.llama import LlamaDecoder
    from gpt4roi_amd.spi_llava import SPILlavaLlamaModel, SPILlavaMPTForCausalLM
    from gpt4roi_amd.train import RegionTrainer
    from gpt4roi_amd.vit import ClipVisionTower

DEV = "cuda"


def relerr(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).abs().max() / want.abs().max().clamp_min(1e-12)).item()


def _mini(seed=0, layers=2, train_projector=True):
    H, P, image = 512, 8, 112
    ids = syn.token_ids(vocab_base=990)
    tower = ClipVisionTower(syn.vit_state(H, 4 * H, 12, image, seed=8), heads=8, device=DEV)
    dec = LlamaDecoder(syn.llama_state(512, 1408, layers, ids.vocab, seed=9), heads=4, max_positions=256, device=DEV,
                       max_batch=2)
    model = SPILlavaLlamaModel(tower, dec, ids, embed_dims=H)
    model.spi_module.load_state_dict(syn.spi_state(model.spi_module, 3 + seed))
    model.spi_module.to(DEV)
    model.mm_projector.to(DEV)
    for p in model.mm_projector.parameters():
        p.requires_grad_(train_projector)
    g = torch.Generator().manual_seed(21 + seed)
    img = torch.randn(2, 3, image, image, generator=g).to(DEV)
    boxes = [syn.boxes(3, g).to(DEV), syn.boxes(1, g).to(DEV)]
    p0 = syn.prompt_ids(ids, P, 3, g, sys_len=5, question_len=6, vocab_base=990)
    p1 = syn.prompt_ids(ids, P, 1, g, sys_len=5, question_len=6 + 8, vocab_base=990)
    prompt = torch.stack([p0, p1]).to(DEV)
    labels = prompt.clone()
    labels[:, :7 + P * P] = -100
    labels[labels >= 990] = -100
    return model, ids, prompt, img, boxes, labels


def test_training_step_through_autograd_equals_region_trainer():
    """HF Trainer.training_step, spelled out: model.train(); loss = model(**batch).loss; loss.backward();
    clip_grad_norm_; torch.optim.AdamW.step() -- against RegionTrainer.step on an identical second model."""
    a, ids, prompt, img, boxes, labels = _mini()
    b, *_ = _mini()
    b.mm_projector.load_state_dict(a.mm_projector.state_dict())     # nn.Linear's default init is not seeded
    lm = SPILlavaMPTForCausalLM(a)
    lm.train()
    attn = torch.ones_like(prompt)
    out = lm(input_ids=prompt, attention_mask=attn, labels=labels, images=img, img_metas=[None, None], bboxes=boxes)
    assert out.loss.requires_grad and out.logits.shape == (2, prompt.size(1), ids.vocab)
    out.loss.backward()
    named = {**{f"spi_module.{k}": p for k, p in a.spi_module.named_parameters()},
             **{f"mm_projector.{k}": p for k, p in a.mm_projector.named_parameters()}}
    assert all(p.grad is not None for p in named.values())
    tr = RegionTrainer(b, lr=2e-5, max_grad_norm=1.0, train_projector=True)
    loss_b, grads_b = tr.loss_and_grads(prompt, img, boxes, labels)
    assert abs(out.loss.item() - loss_b.item()) <= 1e-5 * abs(loss_b.item())
    for k, p in named.items():
        assert relerr(p.grad, grads_b[k]) <= 1e-2, k            # the same kernels ran: equal up to the order of the
                                                                # RoIAlign-backward atomics (bf16 re-rounding downstream)
    # optimizer step: torch AdamW on the nn.Parameters vs the fused kernel of the trainer
    torch.nn.utils.clip_grad_norm_(list(named.values()), 1.0)
    opt = torch.optim.AdamW(list(named.values()), lr=2e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    opt.step()
    opt.zero_grad(set_to_none=True)
    tr.apply(grads_b, lr=2e-5)
    for k, p in named.items():
        assert relerr(p.detach(), tr.params[k].detach()) < 1e-3, k       # first Adam step ~ lr * sign(g): noise-level
                                                                         # gradient entries may flip between the two runs
    # second step: the bf16 kernel copies must follow the optimizer's in-place update (parameter version stamps)
    l2 = lm(input_ids=prompt, labels=labels, images=img, bboxes=boxes).loss
    l2b, _ = tr.loss_and_grads(prompt, img, boxes, labels)
    assert abs(l2.item() - l2b.item()) < 2e-3 * abs(l2b.item()), (l2.item(), l2b.item())
    assert l2.item() != out.loss.item()


def test_logits_seam_accepts_an_arbitrary_downstream_loss():
    """Not only the built-in loss: any torch loss on the returned logits backpropagates into the parameters."""
    a, ids, prompt, img, boxes, labels = _mini(seed=1)
    lm = SPILlavaMPTForCausalLM(a)
    logits = lm(input_ids=prompt, images=img, bboxes=boxes).logits
    assert logits.requires_grad
    shift = logits[:, :-1].reshape(-1, ids.vocab)
    loss = torch.nn.functional.cross_entropy(shift, labels[:, 1:].reshape(-1), ignore_index=-100)
    loss.backward()
    g_torch = {k: p.grad.clone() for k, p in a.spi_module.named_parameters()}
    a.zero_grad(set_to_none=True)
    lm(input_ids=prompt, images=img, bboxes=boxes, labels=labels).loss.backward()
    for k, p in a.spi_module.named_parameters():
        # torch's CE gives fp32 dlogits, the fused kernel bf16 ones: agreement to bf16 rounding of the loss gradient
        assert relerr(p.grad, g_torch[k]) < 3e-2, k


def test_gradient_checkpointing_changes_memory_not_results():
    a, ids, prompt, img, boxes, labels = _mini(seed=2, layers=3)
    lm = SPILlavaMPTForCausalLM(a)
    lm(input_ids=prompt, images=img, bboxes=boxes, labels=labels).loss.backward()
    ref = {k: p.grad.clone() for k, p in a.spi_module.named_parameters()}
    a.zero_grad(set_to_none=True)
    lm.gradient_checkpointing_enable()
    _, ctx = a.forward_train(prompt, img, boxes)
    assert all(set(rec) == {"x"} for rec in ctx["lctx"]["saved"])           # only the layer inputs are kept
    out = lm(input_ids=prompt, images=img, bboxes=boxes, labels=labels)
    out.loss.backward()
    for k, p in a.spi_module.named_parameters():
        assert relerr(p.grad, ref[k]) <= 1e-2, k                 # recomputation is bit-identical; atomics order is not


def test_region_module_seam_B2_under_autograd():
    """MLVLROIQueryModule.forward(mlvl_feats, bboxes) (layers.py:218-236) with grad enabled: out.backward() fills the
    module's .grad with what the explicit forward_train / backward pair returns; no_grad gives the inference result."""
    C, P = 512, 8
    m = MLVLROIQueryModule(embed_dims=C, out_dims=512, num_levels=4)
    m.load_state_dict(syn.spi_state(m, 7))
    m.to(DEV)
    g = torch.Generator().manual_seed(5)
    feats = [torch.randn(2, P * P, C, generator=g).to(DEV).to(torch.bfloat16) for _ in range(4)]
    boxes = [syn.boxes(2, g).to(DEV), syn.boxes(3, g).to(DEV)]
    outs = m(feats, boxes)
    assert [o.shape for o in outs] == [(2, 512), (3, 512)] and outs[0].requires_grad
    d = torch.randn(5, 512, generator=g).to(DEV)
    (torch.cat(outs).float() * d).sum().backward()
    with torch.no_grad():
        out2, ctx = m.forward_train(feats, boxes)
        want = m.backward(ctx, d.to(torch.bfloat16))
        inf = torch.cat(m(feats, boxes))
    assert torch.equal(torch.cat(outs).detach(), out2)
    assert relerr(inf, out2) < 1e-2                              # inference path: same maths, fused kernels
    for k, p in m.named_parameters():
        assert relerr(p.grad, want[k]) <= 1e-2, k
    # an optimizer step invalidates the prepared bf16 buffers
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(1.05)
        assert relerr(torch.cat(m(feats, boxes)), inf) > 1e-3


def test_batch_without_regions_still_trains_the_projector():
    a, ids, prompt, img, boxes, labels = _mini(seed=3)
    P = 8
    g = torch.Generator().manual_seed(4)
    p0 = syn.prompt_ids(ids, P, 0, g, sys_len=5, question_len=9, vocab_base=990)[None].to(DEV)
    lab = p0.clone()
    lab[:, :7 + P * P] = -100
    lm = SPILlavaMPTForCausalLM(a)
    out = lm(input_ids=p0, images=img[:1], bboxes=[torch.zeros(0, 4, device=DEV)], labels=lab)
    out.loss.backward()
    assert a.mm_projector.weight.grad is not None and a.mm_projector.weight.grad.abs().sum() > 0
    # the reference keeps the region module in the graph through a zero dummy term (layers.py:314-317): zero gradients
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in a.spi_module.parameters())


def test_malformed_training_batch_raises_like_the_reference():
    a, ids, prompt, img, boxes, labels = _mini(seed=4)
    lm = SPILlavaMPTForCausalLM(a)
    bad = [boxes[0][:2], torch.cat([boxes[1], boxes[0][2:]])]       # totals match (2 + 2), per-sample counts do not
    with pytest.raises(ValueError):
        lm(input_ids=prompt, images=img, bboxes=bad, labels=labels)
    # a left-padded batch is NOT malformed: the mask reaches the decoder (unpad -> varlen attention -> pad back,
    # llama_flash_attn_monkey_patch.py:60-85; tests/test_varlen_gpu.py checks the arithmetic) and the loss differentiates
    left_padded = torch.ones_like(prompt)
    left_padded[0, :3] = 0
    lab = labels.clone()
    lab[0, :3] = -100
    plain = lm(input_ids=prompt, images=img, bboxes=boxes, labels=lab).loss
    out = lm(input_ids=prompt, attention_mask=left_padded, images=img, bboxes=boxes, labels=lab)
    assert torch.isfinite(out.loss) and a.llama.pos == prompt.size(1)
    assert abs(float(out.loss.detach()) - float(plain.detach())) > 0                   # three keys fewer for every later row of sample 0
    out.loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in a.spi_module.parameters() if p.requires_grad)


def test_stage2_training_through_autograd_equals_full_trainer():
    """stage 2 (train_stage2.sh: everything but the vision tower trains) through the REFERENCE's caller
    (gpt4roi/train/train.py:698-712, HF Trainer.training_step): `enable_decoder_training()` exposes the LLaMA weights as fp32
    nn.Parameters, then it is model.train(); loss = model(**batch).loss; loss.backward(); clip_grad_norm_;
    torch.optim.AdamW(model.parameters()).step() -- step for step against train.FullTrainer on an identical second model:
    same loss every step, same weights after three steps.  And the stage-1 freeze loop of train.py:685-697 switches the
    decoder off again by parameter NAME."""
    from gpt4roi_amd.train import FullTrainer
    lr = 5e-5
    a, ids, prompt, img, boxes, labels = _mini(seed=4)
    b, *_ = _mini(seed=4)
    b.mm_projector.load_state_dict(a.mm_projector.state_dict())
    lm = SPILlavaMPTForCausalLM(a)
    lm.enable_decoder_training()
    names = [n for n, _ in lm.named_parameters()]
    assert any(n.startswith("model.llama_master.layers.0.wqkv") for n in names) and "model.llama_master.lm_head" in names
    assert all(p.requires_grad and p.dtype == torch.float32 for n, p in lm.named_parameters() if "llama_master" in n)
    params = [p for p in lm.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    tr = FullTrainer(b, lr=lr, max_grad_norm=1.0)
    lm.train()
    la, lb = [], []
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        out = lm(input_ids=prompt, labels=labels, images=img, bboxes=boxes)
        out.loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        la.append(out.loss.item())
        lb.append(tr.step(prompt, img, boxes, labels).item())
    print("losses through autograd:", la, "FullTrainer:", lb)
    for x, y in zip(la, lb):
        assert abs(x - y) < 2e-3 * abs(y), (la, lb)                # (atomics order in a few backward kernels)
    assert la[-1] < la[0]
    a._maybe_prepare()                                              # kernel tensors <- the masters the last step moved
    wa, wb = a.llama.export_hf_state_dict(), b.llama.export_hf_state_dict()
    for k in wa:
        d = (wa[k].float() - wb[k].float()).abs().max().item()
        assert d <= 2 ** -7 * wb[k].float().abs().max().item() + 1e-6, (k, d)
    for (k, pa), (_, pb) in zip(a.spi_module.named_parameters(), b.spi_module.named_parameters()):
        d = (pa - pb).abs()
        assert d.max().item() <= 2 * 3 * lr and d.mean().item() <= 0.05 * 3 * lr, k
    # state_dict() (what safe_save_model_for_hf_trainer writes) carries the stepped weights under the reference's names
    sd = lm.state_dict()
    assert torch.equal(sd["model.layers.0.self_attn.o_proj.weight"], wa["model.layers.0.self_attn.o_proj.weight"])
    # stage 1 on the same model: the reference's freeze loop, by name
    for n, p in lm.named_parameters():
        p.requires_grad = "spi_module" in n
    lm.zero_grad(set_to_none=True)
    lm(input_ids=prompt, labels=labels, images=img, bboxes=boxes).loss.backward()
    assert all(p.grad is None for n, p in lm.named_parameters() if "llama_master" in n or "mm_projector" in n)
    assert all(p.grad is not None for n, p in lm.named_parameters() if "spi_module" in n)
    assert a.llama.train_weights is False
