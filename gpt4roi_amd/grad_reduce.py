"""Bucketed, overlapped gradient all-reduce for the data-parallel training rows.

The reference has no explicit collective: stage 1 relies on HF Trainer wrapping the model in
torch DDP (/root/reference/train_stage1.sh:11), stage 2 on FSDP (train_stage2.sh:51-52).  What the
path needs on an 8 x MI355X node is one sum-all-reduce of the trainable gradients per step
(stage 1: the 299 M-parameter `spi_module` = 1.2 GB fp32; stage 2: ~14 GB bf16), overlapped with
the backward pass.  This module is that exchange step, written for xGMI rather than translated from
DDP's defaults:

  * xGMI is point-to-point (7 links x ~153 GB/s per GPU); a ring all-reduce of S bytes is bound by
    one link (t ~ 1.75*S/153 GB/s) while reduce-scatter + all-gather keeps all 7 links busy
    (t ~ 2*(S/8)/153 GB/s).  Buckets are therefore LARGE (default 256 MiB, not DDP's 25 MiB: fewer,
    bigger collectives amortise RCCL launch latency and HBM is 288 GB) and each bucket is reduced as
    reduce_scatter_tensor + all_gather_into_tensor on a dedicated communication stream;
  * gradients are packed into one flat, persistently allocated buffer per bucket in REVERSE
    parameter order (the order backward produces them), so a bucket launches as soon as its last
    gradient has been written while the backward kernels of earlier layers are still running;
  * `finish()` makes the compute stream wait on the communication stream and hands back views of the
    flat buffers (averaged), no extra copy.

Backend agnostic: RCCL ("nccl" on ROCm) on the node, gloo in the CPU tests
(tests/test_grad_reduce_gloo.py, tests/test_train_exchange_gloo.py).  Fed by gpt4roi_amd/train.py.
"""
import torch
import torch.distributed as dist


class _Bucket:
    def __init__(self, params, dtype, device, world):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        pad = (-self.numel) % world                      # reduce_scatter needs equal shards
        self.flat = torch.zeros(self.numel + pad, dtype=dtype, device=device)
        self.views, off = [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        self.pending = len(params)
        self.work = None
        self.event = None


class GradBucketReducer:
    """reducer = GradBucketReducer(params); per step: reducer.reset(); reducer.ready(p, grad) for every
    parameter as its gradient is produced (any order); reducer.finish() -> {param: averaged grad view}."""

    def __init__(self, params, bucket_bytes=256 << 20, group=None, comm_dtype=None, average=True,
                 trainable_only=True, algo="rs_ag"):
        """algo: "rs_ag" = reduce_scatter_tensor + all_gather_into_tensor in place (every xGMI link carries 1/world of
        the bucket in each phase; the default on every backend, so the gloo tests execute the code RCCL runs), or
        "all_reduce" = one all-reduce per bucket (the measured alternative)."""
        assert algo in ("rs_ag", "all_reduce")
        self.algo = algo
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.average = average
        if trainable_only:                      # nn.Parameters; plain tensors (kernel-layout masters) pass False
            params = [p for p in params if p.requires_grad]
        assert params, "no trainable parameters"
        self.device = params[0].device
        self.dtype = comm_dtype or params[0].dtype
        # backward produces gradients roughly in reverse registration order
        order = list(reversed(params))
        self.buckets, cur, cur_bytes = [], [], 0
        esize = torch.tensor([], dtype=self.dtype).element_size()
        for p in order:
            cur.append(p)
            cur_bytes += p.numel() * esize
            if cur_bytes >= bucket_bytes:
                self.buckets.append(_Bucket(cur, self.dtype, self.device, self.world))
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(_Bucket(cur, self.dtype, self.device, self.world))
        self.where = {}
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self.where[id(p)] = (b, i)
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def reset(self):
        for b in self.buckets:
            b.pending = len(b.params)
            b.work = None
            b.event = None

    def slot(self, param):
        """The view of the flat bucket that receives `param`'s gradient: a producer that writes its result there (the
        weight-gradient GEMMs take an output tensor) saves the copy in `ready` -- one read + one write of every gradient."""
        b, i = self.where[id(param)]
        return b.views[i]

    def ready(self, param, grad):
        """Copy `grad` into its slot (unless it was produced there); launch the bucket's collective when the bucket is
        complete."""
        b, i = self.where[id(param)]
        if grad.data_ptr() != b.views[i].data_ptr() or grad.dtype != b.views[i].dtype:
            b.views[i].copy_(grad)
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        if self.world == 1:
            return
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record()                                   # the copies above, on the compute stream
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self._reduce(b)
                b.event = torch.cuda.Event()
                b.event.record()
        else:
            self._reduce(b)

    def _reduce(self, b):
        if self.algo == "rs_ag":
            # direct algorithm over all xGMI links: reduce-scatter then all-gather, in place
            shard = b.flat.numel() // self.world
            rank = dist.get_rank(self.group)
            mine = b.flat[rank * shard:(rank + 1) * shard]
            dist.reduce_scatter_tensor(mine, b.flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                mine.div_(self.world)
            dist.all_gather_into_tensor(b.flat, mine, group=self.group)
        else:
            dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                b.flat.div_(self.world)

    def finish(self):
        """Wait for every bucket; returns {id(param): averaged gradient view into the flat buffers}."""
        for b in self.buckets:
            if b.pending != 0:
                raise RuntimeError(f"{b.pending} gradients of a bucket were never reported ready")
            if b.event is not None:
                torch.cuda.current_stream(self.device).wait_event(b.event)
        out = {}
        for b in self.buckets:
            for p, v in zip(b.params, b.views):
                out[id(p)] = v
        return out

    def describe(self):
        esize = torch.tensor([], dtype=self.dtype).element_size()
        return [dict(params=len(b.params), mbytes=round(b.flat.numel() * esize / 2 ** 20, 1)) for b in self.buckets]
