"""CPU, world_size 2 over gloo: the sharded optimizer of the stage-2 row (gpt4roi_amd/sharded.py).

Checks the part a single GPU cannot: bf16 and fp32 tensors land in separate flat buckets, the live tensors become views of
the parameter buckets, gradients are reduce-scattered as their bucket fills, the global gradient norm is the all-reduced sum
of the per-rank shard sums, each rank updates only its shard (and holds only 1/world of the optimizer state), and the
in-place all-gather leaves BOTH ranks with exactly the parameters an unsharded clip_grad_norm_ + AdamW on the averaged
gradients produces.  The HIP update kernel needs a GPU, so a torch restatement of AdamW is injected as `update_fn`
(tests/test_train_gpu.py checks the fused kernel path of the same class on the device)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _mp

from gpt4roi_amd.sharded import ShardedAdamW

LR, CLIP, BETAS, EPS = 1e-2, 0.5, (0.9, 0.999), 1e-8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tensors():
    g = torch.Generator().manual_seed(0)
    spec = [("embed", (50, 16), torch.bfloat16), ("l0.norm", (16,), torch.float32), ("l0.wqkv", (48, 16), torch.bfloat16),
            ("l0.wd", (16, 45), torch.bfloat16), ("norm", (16,), torch.float32), ("head", (50, 16), torch.bfloat16),
            ("proj.w", (7, 3), torch.float32)]
    return [(n, torch.randn(s, generator=g).to(dt)) for n, s, dt in spec]


def _grad(i, shape, rank, step):
    g = torch.Generator().manual_seed(1000 * step + 10 * i + rank)
    return torch.randn(shape, generator=g)


def torch_update(b, lr, step, betas, eps, wd, total_sq, max_norm):
    """AdamW on the owned shard of one bucket, torch ops (the math of g4r_multi_adamw_f32)."""
    g = b.grad_shard
    if total_sq is not None:
        coef = max_norm / (float(total_sq.sqrt()) + 1e-6)
        if coef < 1.0:
            g = g * coef
    p = b.master
    b.exp_avg.mul_(betas[0]).add_(g, alpha=1 - betas[0])
    b.exp_avg_sq.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
    p.mul_(1 - lr * wd)
    denom = (b.exp_avg_sq / (1 - betas[1] ** step)).sqrt_().add_(eps)
    p.addcdiv_(b.exp_avg / (1 - betas[0] ** step), denom, value=-lr)
    if b.live_dtype == torch.bfloat16:
        b.param_shard.copy_(p)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        live = dict(_tensors())
        entries = list(live.items())

        def rebind(name, view):
            live[name] = view

        opt = ShardedAdamW(entries, rebind, bucket_bytes=2048, betas=BETAS, eps=EPS, update_fn=torch_update)
        layout = [(str(b.live_dtype), [n for n, _ in b.entries], b.shard) for b in opt.buckets]
        for b in opt.buckets:                                   # live tensors are views of the flat parameter bucket
            for (n, _), pv in zip(b.entries, b.pviews):
                assert live[n].data_ptr() == pv.data_ptr()
        norms = []
        for step in range(3):
            opt.reset()
            for i in reversed(range(len(entries))):             # backward order, same on every rank
                n = entries[i][0]
                opt.ready(n, _grad(i, live[n].shape, rank, step))
            norms.append(float(opt.step(LR, CLIP)))
        q.put(_mp.plain((rank, layout, norms, {n: t.clone() for n, t in live.items()}, opt.state_bytes())))
    finally:
        dist.destroy_process_group()


def _unsharded_reference(steps=3):
    """fp32 masters + torch.optim.AdamW + clip_grad_norm_ on the rank-averaged gradients; bf16 tensors rounded per step."""
    tensors = _tensors()
    masters = [torch.nn.Parameter(t.float()) for _, t in tensors]
    opt = torch.optim.AdamW(masters, lr=LR, betas=BETAS, eps=EPS, weight_decay=0.0)
    norms = []
    for step in range(steps):
        for i, m in enumerate(masters):
            m.grad = (_grad(i, m.shape, 0, step) + _grad(i, m.shape, 1, step)) / 2
        norms.append(float(torch.nn.utils.clip_grad_norm_(masters, CLIP)) ** 2)
        opt.step()
    return {n: m.detach().to(t.dtype) for (n, t), m in zip(tensors, masters)}, norms


def test_sharded_adamw_two_ranks_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((_mp.tensors(q.get(timeout=120)) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, lay0, n0, p0, sb0), (_, lay1, n1, p1, sb1) = res
    assert lay0 == lay1 and len(lay0) >= 3
    for dt, names, _ in lay0:                                   # a bucket never mixes bf16 and fp32 tensors
        kinds = {dict(_tensors())[n].dtype for n in names}
        assert kinds == {torch.bfloat16 if "bfloat16" in dt else torch.float32}
    want, want_norms = _unsharded_reference()
    for got in (p0, p1):
        for n, w in want.items():
            if w.dtype == torch.bfloat16:                       # one bf16 ulp: fp32 op order differs from torch.optim's
                torch.testing.assert_close(got[n].float(), w.float(), rtol=2 ** -7, atol=1e-6)
            else:
                torch.testing.assert_close(got[n], w, rtol=1e-5, atol=1e-6)
    for n in want:                                              # both ranks end bit-identical (the all-gather)
        assert torch.equal(p0[n], p1[n])
    for a, b, w in zip(n0, n1, want_norms):
        assert abs(a - w) <= 1e-5 * w and abs(b - w) <= 1e-5 * w
    own, full = sb0
    assert own <= full / 2 + 64 and sb0 == sb1                 # each rank holds half of the optimizer state


def test_world_one_is_a_local_update():
    live = dict(_tensors())
    entries = list(live.items())
    opt = ShardedAdamW(entries, lambda n, v: live.__setitem__(n, v), bucket_bytes=2048, betas=BETAS, eps=EPS,
                       update_fn=torch_update)
    before = {n: t.clone() for n, t in live.items()}
    opt.reset()
    for i in reversed(range(len(entries))):
        opt.ready(entries[i][0], torch.ones(entries[i][1].shape))
    opt.step(LR, None)
    for n, t in live.items():                                   # first AdamW step with unit gradients moves every value by ~lr
        d = (before[n].float() - t.float())
        assert (d >= 0).all() and abs(float(d.mean()) - LR) < 0.2 * LR     # (bf16 rounding per element, lr on average)
    own, full = opt.state_bytes()
    assert own >= full
