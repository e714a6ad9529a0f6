"""GPU: the gradient exchange of the data-parallel training rows checked BY VALUE on the device (SURVEY.md
8a row a18).  Two ranks share the one MI355X of the test box and exchange over gloo (RCCL refuses two ranks on one device; on
the 8-GPU node the same code runs one rank per GPU over RCCL -- tests/test_rccl_gpu.py executes those collectives on RCCL).
Each rank runs the HIP training step on ITS half of a batch; asserted:

  * the rank-averaged gradients the trainers hand to clip + AdamW equal the gradients of ONE process on the WHOLE batch
    (per parameter: max-norm error and cosine; the only differences are the reduction order and the GEMM tile dispatch of a
    batch of 2 against a batch of 4),
  * the parameters after two steps are bit-identical on the two ranks and equal the one-process run within Adam's step bound,
  * for both exchange algorithms of grad_reduce.GradBucketReducer ("rs_ag" = reduce-scatter + all-gather, "all_reduce"),
  * and for train.FSDPFullTrainer (parameters + gradients + optimizer state sharded per decoder layer) at mini width: the
    gathered parameters after two sharded steps equal the unsharded FullTrainer on the whole batch.

What the reference does here: DDP under HF Trainer (train_stage1.sh:11 `torchrun --nproc_per_node=4`) and FSDP full-shard
(train_stage2.sh:29,51-52) -- mean loss per rank, gradients averaged over the ranks.  The halves hold the same number of
supervised tokens, so the average of the rank means IS the whole-batch mean."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

import _mp  # noqa: E402

LR, STEPS = 5e-5, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tiny(n_img=4, seed=8):
    """the mini-width model of tests/test_train_gpu.py::_tiny_stage2 with a batch of n_img images (3 regions each, one
    prompt structure: every sample supervises the same number of tokens)"""
    from oracle import spi_oracle as S
    from gpt4roi_amd import synthetic as syn
    from gpt4roi_amd.llama import LlamaDecoder
    from gpt4roi_amd.spi_llava import SPILlavaLlamaModel
    from gpt4roi_amd.vit import ClipVisionTower
    dev = "cuda"
    H, P, image = 512, 8, 112
    ids = syn.token_ids(vocab_base=990)
    tower = ClipVisionTower(syn.vit_state(H, 4 * H, 12, image, seed=seed), heads=8, device=dev)
    dec = LlamaDecoder(syn.llama_state(512, 1408, 2, ids.vocab, seed=seed + 1), heads=4, max_positions=256, device=dev)
    model = SPILlavaLlamaModel(tower, dec, ids, embed_dims=H)
    orc = S.MLVLROIQueryOracle(embed_dims=H, P=P)
    orc.roi_align.updims = torch.nn.Linear(1024, 512)
    model.spi_module.load_state_dict(S.synthetic_state(orc, seed + 2))
    g = torch.Generator().manual_seed(seed + 3)
    with torch.no_grad():
        model.mm_projector.weight.copy_(torch.randn(512, H, generator=g) / H ** 0.5)
        model.mm_projector.bias.copy_(torch.randn(512, generator=g) * 0.05)
    img = torch.randn(n_img, 3, image, image, generator=g)
    boxes = [syn.boxes(3, g) for _ in range(n_img)]
    prompt = torch.stack([syn.prompt_ids(ids, P, 3, g, sys_len=6, question_len=9, vocab_base=990) for _ in range(n_img)])
    labels = prompt.clone()
    labels[:, :8 + P * P] = -100
    labels[labels >= 990] = -100
    assert len({int((row != -100).sum()) for row in labels}) == 1          # equal supervised-token counts per sample
    return model, (prompt.to(dev), img.to(dev), [b.to(dev) for b in boxes], labels.to(dev))


def _half(args, rank, world):
    prompt, img, boxes, labels = args
    n = prompt.size(0) // world
    sl = slice(rank * n, (rank + 1) * n)
    return prompt[sl], img[sl], boxes[sl], labels[sl]


def _run(kind, algo, rank=0, world=1):
    """STEPS training steps of trainer `kind` on this rank's share of the batch -> (losses, first-step gradients, final
    parameters), everything by name"""
    from gpt4roi_amd.train import FSDPFullTrainer, FullTrainer, RegionTrainer
    model, args = _tiny()
    mine = _half(args, rank, world)
    if kind == "region":
        tr = RegionTrainer(model, lr=LR, train_projector=True, bucket_bytes=4 << 20, exchange_algo=algo)
    elif kind == "full":
        tr = FullTrainer(model, lr=LR, bucket_bytes=4 << 20, exchange_algo=algo)
    else:
        tr = FSDPFullTrainer(model, lr=LR)
    losses, grads1 = [], None
    for s in range(STEPS):
        if kind == "fsdp":
            losses.append(float(tr.step(*mine)))
            continue
        loss, grads = tr.loss_and_grads(*mine, exchange=tr.reducer is not None)
        if s == 0:
            grads1 = {k: v.detach().float().clone().reshape(-1) for k, v in grads.items()}
        tr.apply(grads, exchanged=tr.reducer is not None)
        losses.append(float(loss))
    if kind == "fsdp":
        params = {k: v.float() for k, v in tr.full_state_dict().items()}
    else:
        params = {k: p.detach().clone() for k, p in tr.params.items()}
        if kind == "full":
            params.update({k: v.detach().float().clone() for k, v in tr.dec_master.items()})
    n_buckets = len(tr.reducer.buckets) if getattr(tr, "reducer", None) is not None else 0
    return losses, grads1, params, n_buckets


def _worker(rank, world, port, kind, algo, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)                                  # both ranks on the one GPU of the test box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        losses, grads1, params, n_buckets = _run(kind, algo, rank, world)
        q.put(_mp.plain((rank, losses, grads1, params, n_buckets)))
    finally:
        dist.destroy_process_group()


def _two_ranks(kind, algo):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, kind, algo, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((_mp.tensors(q.get(timeout=900)) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return res


def _cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float(a @ b / (a.norm() * b.norm()).clamp_min(1e-30))


def _check_params(p0, p1, ref, what):
    for k in ref:
        assert torch.equal(p0[k], p1[k]), f"{what}: {k} differs between the two ranks"
        d = (p0[k].float().cpu() - ref[k].float().cpu()).abs()
        # Adam turns reduction-order noise on ~0 gradients into +-lr steps: bound by the step size, as the one-rank
        # sharded-vs-unsharded tests do (tests/test_train_gpu.py)
        tol = 2 * STEPS * LR + 2 ** -7 * float(ref[k].float().abs().max())        # (+ one bf16 rounding for the fsdp export)
        assert float(d.max()) <= tol and float(d.mean()) <= 0.1 * STEPS * LR + 2 ** -9 * float(ref[k].float().abs().mean()), \
            (what, k, float(d.max()), float(d.mean()))


@pytest.mark.parametrize("kind,algo", [("region", "rs_ag"), ("region", "all_reduce"), ("full", "rs_ag")])
def test_two_ranks_on_halves_equal_one_process_on_the_whole_batch(kind, algo):
    (r0, l0, g0, p0, nb0), (r1, l1, g1, p1, nb1) = _two_ranks(kind, algo)
    assert nb0 == nb1 and nb0 >= (2 if kind == "region" else 3)           # several buckets in flight during the backward
    ref_losses, ref_grads, ref_params, _ = _run(kind, algo)               # ONE process, the whole batch, no exchange
    # the rank-mean losses average to the whole-batch loss
    for s in range(STEPS):
        assert abs((l0[s] + l1[s]) / 2 - ref_losses[s]) < 2e-3 * abs(ref_losses[s]), (s, l0, l1, ref_losses)
    # the exchanged gradients, by value
    worst = []
    for k, want in ref_grads.items():
        assert torch.equal(g0[k], g1[k]), f"{k}: the two ranks hold different averaged gradients"
        want = want.cpu()
        err = float((g0[k].cpu() - want).abs().max() / want.abs().max().clamp_min(1e-12))
        worst.append((round(err, 4), round(1 - _cos(g0[k], want), 6), k))
    worst.sort()
    print(f"{kind}/{algo}: exchanged gradient vs the whole-batch gradient, worst five (max-norm err, 1 - cos):", worst[-5:])
    assert worst[-1][0] < 5e-2 and max(w[1] for w in worst) < 2e-3, worst[-5:]
    _check_params(p0, p1, ref_params, f"{kind}/{algo}")


def test_fsdp_full_shard_two_ranks_equal_the_unsharded_trainer_on_the_whole_batch():
    (r0, l0, _, p0, _), (r1, l1, _, p1, _) = _two_ranks("fsdp", "rs_ag")
    ref_losses, _, ref_params, _ = _run("full", "rs_ag")                  # unsharded FullTrainer, one process, whole batch
    for s in range(STEPS):
        assert abs((l0[s] + l1[s]) / 2 - ref_losses[s]) < 2e-3 * abs(ref_losses[s]), (s, l0, l1, ref_losses)
    assert set(p0) == set(ref_params), set(p0) ^ set(ref_params)
    _check_params(p0, p1, ref_params, "fsdp")
