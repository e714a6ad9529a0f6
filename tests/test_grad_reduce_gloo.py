"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce of the data-parallel training rows
(gpt4roi_amd/grad_reduce.py) -- bucket packing in reverse parameter order, out-of-order readiness,
averaging, identical results on both ranks, persistent buffers across steps."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _mp import plain, tensors

from gpt4roi_amd.grad_reduce import GradBucketReducer


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _params():
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 32), (64,), (3, 3, 8, 8), (1000,), (7,), (128, 16)]
    return [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]


def _grad(p_index, shape, rank, step):
    g = torch.Generator().manual_seed(1000 * step + 10 * p_index + rank)
    return torch.randn(shape, generator=g)


def _worker(rank, world, port, q, algo="rs_ag"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = _params()
        red = GradBucketReducer(params, bucket_bytes=4096, algo=algo)   # small buckets -> several of them
        assert red.algo == algo
        desc = red.describe()
        results = []
        for step in range(2):
            red.reset()
            # collectives must be issued in the same order on every rank: both report in backward order
            for i in reversed(range(len(params))):
                red.ready(params[i], _grad(i, params[i].shape, rank, step))
            out = red.finish()
            results.append([out[id(p)].clone() for p in params])
        q.put(plain((rank, desc, results)))
    finally:
        dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("algo", ["rs_ag", "all_reduce"])
def test_bucketed_allreduce_two_ranks(algo):
    """algo="rs_ag" is the branch RCCL runs on the node (in-place reduce_scatter_tensor into a view of its own input, then
    all_gather_into_tensor); gloo executes the very same calls here.  Odd bucket sizes exercise the shard padding."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, algo)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((tensors(q.get(timeout=120)) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, desc0, r0), (_, desc1, r1) = res
    assert desc0 == desc1 and len(desc0) >= 3                       # several buckets, same layout on both ranks
    params = _params()
    for step in range(2):
        for i, p in enumerate(params):
            want = (_grad(i, p.shape, 0, step) + _grad(i, p.shape, 1, step)) / 2
            torch.testing.assert_close(r0[step][i], want)
            torch.testing.assert_close(r1[step][i], want)


def test_single_process_is_a_pass_through():
    params = _params()
    red = GradBucketReducer(params, bucket_bytes=1 << 20)
    red.reset()
    grads = [torch.full_like(p, float(i)) for i, p in enumerate(params)]
    for p, g in zip(params, grads):
        red.ready(p, g)
    out = red.finish()
    for p, g in zip(params, grads):
        assert torch.equal(out[id(p)], g)
    red.reset()
    red.ready(params[0], grads[0])
    try:
        red.finish()
        raise AssertionError("finish() must refuse incomplete buckets")
    except RuntimeError:
        pass
