"""Multi-GPU glue for the forward/inference path: one process per GPU, images sharded across ranks,
NO data-path collective (every stage of the path is per-image: ViT, region module, RoIAlign by RoI ->
image, splice per sample, decoder per sequence).  The only communication is the
benchmark bookkeeping below (a barrier and a MAX/SUM all-reduce of two scalars), which is backend
agnostic: RCCL ("nccl" on ROCm) on the GPU box, gloo in the CPU tests.

Training (stage 1/2) adds a real exchange step -- the gradient all-reduce; that lives in grad_reduce.py and
train.py, not here.
"""
import time


def shard(n_units, rank, world):
    """Contiguous, balanced split of `n_units` independent units (images): every unit belongs to
    exactly one rank; sizes differ by at most one."""
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def timed_steps(step_fn, steps, sync_fn, dist=None):
    """barrier + device sync, `steps` calls of step_fn, barrier + device sync.
    Returns this rank's elapsed seconds."""
    if dist is not None:
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    if dist is not None:
        dist.barrier()
    sync_fn()
    return time.perf_counter() - t0


def aggregate(units_local, seconds_local, dist=None, device="cpu"):
    """Whole-job numbers: (sum of units over ranks, max of seconds over ranks)."""
    if dist is None:
        return units_local, seconds_local
    import torch
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(u.item()), float(t.item())
