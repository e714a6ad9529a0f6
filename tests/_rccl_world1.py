"""Executed by tests/test_rccl_gpu.py in a process of its own: a ONE-rank "nccl" (= RCCL on ROCm) process group on the MI355X,
through which the exact collective calls of the training exchange run on device buffers -- the in-place
reduce_scatter_tensor into a view of its own input and the all_gather_into_tensor back (grad_reduce.py), the sharded
optimizer's reduce-scatter / all-gather (sharded.py), and bench.py's aggregate().  With one rank the results are the
identities; what this proves is that RCCL accepts these calls, buffer aliasing included (world > 1 runs over gloo in the CPU
tests; the driver's 8-GPU box is the only place they can run over xGMI)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt4roi_amd import replicas                      # noqa: E402
from gpt4roi_amd.grad_reduce import GradBucketReducer  # noqa: E402
from gpt4roi_amd.sharded import ShardedAdamW           # noqa: E402

os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29611"), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
assert dist.get_backend() == "nccl"
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
params = [torch.nn.Parameter(torch.randn(s, generator=g, device=dev)) for s in [(64, 32), (1000,), (7,), (128, 16)]]
grads = [torch.randn(p.shape, generator=g, device=dev) for p in params]
for algo in ("rs_ag", "all_reduce"):
    red = GradBucketReducer(params, bucket_bytes=4096, algo=algo)
    red.world = 1
    red.reset()
    for p, gr in zip(reversed(params), reversed(grads)):
        red.ready(p, gr)
    # world 1 skips the launch in ready(); run the bucket collectives explicitly on the communication stream
    for b in red.buckets:
        with torch.cuda.stream(red.comm_stream):
            red._reduce(b)
    torch.cuda.synchronize()
    out = red.finish()
    for p, gr in zip(params, grads):
        assert torch.equal(out[id(p)], gr), algo
# sharded optimizer: reduce-scatter of the gradient bucket, all-gather of the parameter bucket
live = {f"t{i}": torch.randn(s, generator=g, device=dev).to(torch.bfloat16 if i % 2 else torch.float32)
        for i, s in enumerate([(256, 64), (512,), (33, 7)])}
before = {k: v.clone() for k, v in live.items()}
opt = ShardedAdamW(list(live.items()), lambda n, v: live.__setitem__(n, v), bucket_bytes=8192)
opt.reset()
for n in reversed(list(live)):
    opt.ready(n, torch.ones(live[n].shape, device=dev))
for b in opt.buckets:
    opt._reduce_scatter(b)
    dist.all_gather_into_tensor(b.param, b.param_shard)
torch.cuda.synchronize()
for k in live:
    assert torch.equal(live[k], before[k])
u, t = replicas.aggregate(32.0, 0.5, dist, device=dev)
assert (u, t) == (32.0, 0.5)
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
