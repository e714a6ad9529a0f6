"""CPU, world_size 2 over gloo: the N > 1 bookkeeping of the sharded forward path
(gpt4roi_amd/replicas.py) -- disjoint/complete sharding, barrier-bracketed timing, SUM/MAX
aggregation -- exactly what bench.py runs over RCCL on the GPU node."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpt4roi_amd import replicas


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = list(replicas.shard(7, rank, world))
        done = []

        def step():
            time.sleep(0.02 * (rank + 1))        # rank 1 is the slow one
            done.append(1)
        dt = replicas.timed_steps(step, 3, lambda: None, dist)
        units, tmax = replicas.aggregate(len(mine) * 32, dt, dist)
        out.put((rank, mine, dt, units, tmax, len(done)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, dt0, u0, t0, n0), (r1, s1, dt1, u1, t1, n1) = res
    assert sorted(s0 + s1) == list(range(7)) and not set(s0) & set(s1)      # every image exactly once
    assert abs(len(s0) - len(s1)) <= 1
    assert n0 == n1 == 3
    assert u0 == u1 == 7 * 32                                               # SUM of region tokens
    assert t0 == t1 and t0 >= max(dt0, dt1) - 1e-9 and t0 >= 0.11           # MAX over ranks (slow rank ~0.12 s)
    # the closing barrier makes the fast rank wait for the slow one
    assert dt0 >= 0.11


def test_single_process_path_needs_no_process_group():
    assert replicas.aggregate(64, 0.5) == (64, 0.5)
    assert list(replicas.shard(5, 0, 1)) == [0, 1, 2, 3, 4]
    assert [len(replicas.shard(10, r, 4)) for r in range(4)] == [3, 3, 2, 2]
