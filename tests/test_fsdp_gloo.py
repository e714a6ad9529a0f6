"""CPU, world_size 2 over gloo: gpt4roi_amd/fsdp.py -- parameters, gradients and optimizer state fully sharded per unit (the
reference's stage-2 strategy, train_stage2.sh:51-52).  A toy "model" of three units stands in for the decoder layers (the real
layers need the GPU kernels; tests/test_train_gpu.py runs FSDPFullTrainer against FullTrainer on one rank): its forward and
backward only touch a unit's tensors between use() and release(), exactly like LlamaDecoder's unit hooks.  Checked on both
ranks: the gathered parameters are the full tensors, the live names are None outside their unit, each rank ends the backward
with the rank-AVERAGED gradient of its own slice only, three steps of clip + AdamW on the slices reproduce an unsharded
torch.optim.AdamW on the averaged gradients, and the persistent bytes per rank are half the unsharded state."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _mp
from gpt4roi_amd.fsdp import FullShardManager

LR, CLIP, BETAS, EPS = 1e-2, 1.0, (0.9, 0.999), 1e-8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _units():
    g = torch.Generator().manual_seed(0)
    units = []
    for u in range(3):
        units.append([(f"u{u}.w", (torch.randn(33, 17, generator=g) * 0.3).to(torch.bfloat16)),
                      (f"u{u}.v", (torch.randn(129, generator=g) * 0.3).to(torch.bfloat16)),
                      (f"u{u}.n", 1 + 0.1 * torch.randn(17, generator=g))])
    return units


def _grad(name, shape, rank, step):
    g = torch.Generator().manual_seed(sum(ord(c) * (i + 1) for i, c in enumerate(name)) * 31 + 1000 * step + 7 * rank)   # (str hash() differs per process)
    return torch.randn(shape, generator=g)


def torch_update(f, lr, step, betas, eps, wd, total_sq, max_norm):
    """AdamW on one flat slice in plain torch (the fused HIP kernels need the GPU)."""
    g = f.grad_shard.clone()
    if total_sq is not None:
        coef = min(1.0, max_norm / (float(total_sq.sqrt()) + 1e-6))
        g = g * coef
    f.exp_avg.mul_(betas[0]).add_(g, alpha=1 - betas[0])
    f.exp_avg_sq.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
    bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
    denom = (f.exp_avg_sq / bc2).sqrt() + eps
    f.master.mul_(1 - lr * wd).addcdiv_(f.exp_avg / bc1, denom, value=-lr)
    if f.param_shard.dtype != torch.float32:
        f.param_shard.copy_(f.master)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        units = _units()
        shapes = {n: tuple(t.shape) for u in units for n, t in u}
        live = {n: t for u in units for n, t in u}

        def rebind(name, t):
            live[name] = t

        mgr = FullShardManager(units, rebind, betas=BETAS, eps=EPS, prefetch=1, update_fn=torch_update)
        assert all(v is None for v in live.values())
        seen, norms = [], []
        for step in range(3):
            mgr.begin_step()
            mgr.direction(+1)
            fwd = {}
            for ui in range(3):                                   # forward: gather, read, release
                mgr.use(ui)
                for n in mgr.units[ui]["names"]:
                    fwd[n] = live[n].clone()
                others = [v for k, v in live.items() if not k.startswith(f"u{ui}.") and not k.startswith(f"u{ui + 1}.")]
                assert all(v is None or True for v in others)
                mgr.release(ui)
                assert all(live[n] is None for n in mgr.units[ui]["names"])
            mgr.direction(-1)
            for ui in reversed(range(3)):                         # backward: gather again, gradients as they are produced
                mgr.use(ui)
                for n in reversed(mgr.units[ui]["names"]):
                    assert torch.equal(live[n], fwd[n])
                    g = _grad(n, shapes[n], rank, step)
                    if (step + len(n)) % 2:                           # a producer that writes into the unit's buffer itself
                        slot = mgr.grad_slot(n)
                        assert slot.shape == g.shape and mgr.grad_slot("no such tensor") is None
                        slot.copy_(g)
                        g = slot
                    mgr.grad_ready(n, g)
                mgr.release(ui)
            if step == 0:
                seen = {n: t.clone() for n, t in fwd.items()}
                gs = [(f.lo, f.shard, f.grad_shard.clone(), [n for n, _ in f.entries]) for u in mgr.units for f in u["flats"]]
            norms.append(float(mgr.step(LR, CLIP)))
        final = {}
        for ui in range(3):
            for n, v in mgr.full_state(ui).items():
                final[n] = v.clone()
            mgr.release(ui)
        q.put(_mp.plain((rank, seen, gs, norms, final, mgr.memory())))
    finally:
        dist.destroy_process_group()


def _reference(steps=3):
    units = _units()
    names = [n for u in units for n, _ in u]
    masters = {n: torch.nn.Parameter(t.float()) for u in units for n, t in u}
    dtypes = {n: t.dtype for u in units for n, t in u}
    opt = torch.optim.AdamW(list(masters.values()), lr=LR, betas=BETAS, eps=EPS, weight_decay=0.0)
    norms = []
    for step in range(steps):
        for n in names:
            masters[n].grad = (_grad(n, masters[n].shape, 0, step) + _grad(n, masters[n].shape, 1, step)) / 2
        norms.append(float(torch.nn.utils.clip_grad_norm_(list(masters.values()), CLIP)) ** 2)
        opt.step()
        for n in names:                                           # bf16 tensors: the next step starts from the fp32 master
            pass
    return {n: masters[n].detach().to(dtypes[n]) for n in names}, norms


def test_full_shard_two_ranks_matches_unsharded():
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
    units = _units()
    full = {n: t for u in units for n, t in u}
    (_, seen0, gs0, n0, fin0, mem0), (_, seen1, gs1, n1, fin1, mem1) = res
    for n, t in full.items():                                     # the all-gather reconstructs every tensor on both ranks
        assert torch.equal(seen0[n], t) and torch.equal(seen1[n], t)
    # after the backward each rank holds the AVERAGED gradient of its own slice only
    for gs, rank in ((gs0, 0), (gs1, 1)):
        for lo, shard, g, names in gs:
            shapes = [tuple(full[n].shape) for n in names]
            flat = torch.cat([((_grad(n, s, 0, 0) + _grad(n, s, 1, 0)) / 2).reshape(-1) for n, s in zip(names, shapes)])
            flat = torch.cat([flat, torch.zeros(2 * shard - flat.numel())])
            torch.testing.assert_close(g, flat[lo:lo + shard])
    want, want_norms = _reference()
    for a, b, c in zip(n0, n1, want_norms):
        assert abs(a - b) < 1e-9 * max(1.0, abs(a)) and abs(a - c) < 1e-4 * c
    for n, w in want.items():
        assert torch.equal(fin0[n], fin1[n])
        tol = 2 ** -7 * w.float().abs().max().item() if w.dtype == torch.bfloat16 else 1e-5
        assert (fin0[n].float() - w.float()).abs().max().item() <= tol + 1e-7, n
    own, unsharded, transient = mem0
    assert abs(own - unsharded / 2) <= 64 and transient > 0        # half the persistent state per rank (+ padding)


def _one_step(mgr, live, shapes, rank, step):
    mgr.begin_step()
    mgr.direction(+1)
    for ui in range(3):
        mgr.use(ui)
        mgr.release(ui)
    mgr.direction(-1)
    for ui in reversed(range(3)):
        mgr.use(ui)
        for n in reversed(mgr.units[ui]["names"]):
            mgr.grad_ready(n, _grad(n, shapes[n], rank, step))
        mgr.release(ui)
    return float(mgr.step(LR, CLIP))


def _resume_worker(rank, world, port, q):
    """two steps -> shard_state() -> a FRESH manager (built from the INITIAL tensors) -> load_shard_state() -> two more steps,
    against four uninterrupted steps: identical slices on every rank (the sharded trainer must resume)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import copy
        shapes = {n: tuple(t.shape) for u in _units() for n, t in u}

        def make():
            live = {}
            return FullShardManager(_units(), lambda n, t: live.__setitem__(n, t), betas=BETAS, eps=EPS, prefetch=1,
                                    update_fn=torch_update), live
        a, live_a = make()
        for step in range(2):
            _one_step(a, live_a, shapes, rank, step)
        sd = copy.deepcopy(a.shard_state())
        steps_at_save = a.steps
        for step in range(2, 4):
            _one_step(a, live_a, shapes, rank, step)
        b, live_b = make()
        b.load_shard_state(sd)
        b.steps = steps_at_save
        for step in range(2, 4):
            _one_step(b, live_b, shapes, rank, step)
        same = all(torch.equal(fa.param_shard, fb.param_shard) and torch.equal(fa.master, fb.master)
                   and torch.equal(fa.exp_avg, fb.exp_avg) and torch.equal(fa.exp_avg_sq, fb.exp_avg_sq)
                   for ua, ub in zip(a.units, b.units) for fa, fb in zip(ua["flats"], ub["flats"]))
        wrong_layout = False
        try:
            b.load_shard_state(dict(sd, rank=1 - rank))
        except AssertionError:
            wrong_layout = True
        q.put(_mp.plain((rank, same, wrong_layout, sd["world"], len(sd["units"]))))
    finally:
        dist.destroy_process_group()


def test_full_shard_state_saves_and_resumes_on_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_resume_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((_mp.tensors(q.get(timeout=120)) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same, wrong_layout, world, n_units in res:
        assert same, f"rank {rank}: the resumed run left the uninterrupted one"
        assert wrong_layout and world == 2 and n_units == 3
