"""CPU, world_size 2 over gloo: the gradient exchange exactly as the trainers call it (train.exchange_gradients):
name-keyed gradients of mixed kinds -- nn.Parameters in the reference layout (region module) and plain fp32 master
tensors in kernel layouts (stage-2 decoder, GradBucketReducer(trainable_only=False)) -- averaged over the ranks and
handed back under the same names."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _mp

from gpt4roi_amd.grad_reduce import GradBucketReducer
from gpt4roi_amd.train import exchange_gradients


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tensors():
    g = torch.Generator().manual_seed(0)
    t = {"spi_module.roi_align.updims.weight": torch.nn.Parameter(torch.randn(32, 16, generator=g)),
         "spi_module.mlvl_fuse.fuse_convs.0.conv.weight": torch.nn.Parameter(torch.randn(8, 8, 3, 3, generator=g)),
         "spi_module.mlvl_fuse.fuse_convs.0.gn.bias": torch.nn.Parameter(torch.randn(8, generator=g)),
         "llama.0.wqkv": torch.randn(48, 16, generator=g),               # plain fp32 masters (kernel layout)
         "llama.norm": torch.randn(16, generator=g),
         "llama.embed_tokens": torch.randn(100, 16, generator=g)}
    return t


def _grads(tensors, rank, step):
    out = {}
    for i, (k, v) in enumerate(tensors.items()):
        g = torch.Generator().manual_seed(100 * step + 10 * i + rank)
        out[k] = torch.randn(v.shape, generator=g)
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tensors = _tensors()
        red = GradBucketReducer(list(tensors.values()), bucket_bytes=2048, comm_dtype=torch.float32, trainable_only=False)
        res = []
        for step in range(2):
            avg = exchange_gradients(red, tensors, _grads(tensors, rank, step))
            res.append({k: v.clone() for k, v in avg.items()})
        q.put(_mp.plain((rank, len(red.buckets), res)))
    finally:
        dist.destroy_process_group()


def test_named_gradient_exchange_two_ranks():
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
    tensors = _tensors()
    assert res[0][1] > 1                                               # several buckets
    for step in range(2):
        want = {k: (_grads(tensors, 0, step)[k] + _grads(tensors, 1, step)[k]) / 2 for k in tensors}
        for rank in range(2):
            got = res[rank][2][step]
            assert list(got) == list(tensors)
            for k in tensors:
                assert got[k].shape == tensors[k].shape
                assert torch.allclose(got[k], want[k], atol=1e-6), (step, rank, k)
