"""Full sharding of parameters, gradients AND optimizer state per decoder layer -- the memory strategy the reference's stage 2
actually launches: `--fsdp "full_shard auto_wrap" --fsdp_transformer_layer_cls_to_wrap LlamaDecoderLayer`
(/root/reference/train_stage2.sh:51-52; the FSDP patch at gpt4roi/train/train.py:655-676), SURVEY.md 8f-3.

sharded.py shards the optimizer state only (ZeRO-1/2: a 7 B replica + its gradients fit the 288 GB of an MI355X).  This
module is the ZeRO-3 form that makes the 13 B models of the reference's README fit and frees HBM for the batch:

  * a UNIT is what FSDP's auto-wrap policy wraps: one LlamaDecoderLayer (its kernel tensors wqkv, wo, wgu, wd in bf16 and the
    two RMSNorm weights in fp32), plus one root unit for what sits outside the layers (embedding table, final norm, lm_head);
  * every unit's tensors are laid out in ONE flat buffer per dtype, padded to a multiple of the world size.  Each rank keeps
    PERSISTENTLY only its 1/world slice of it: the parameter shard (bf16 or fp32), the fp32 master + Adam moments of that
    slice, and -- between backward and step -- the fp32 gradient slice;
  * forward of unit u: `gather(u)` all-gathers the shards into a transient full buffer taken from a small pool (the next
    unit's gather is issued on the communication stream while this unit computes: prefetch), `use(u)` makes the compute stream
    wait for it and re-points the live tensors the kernels read at views of that buffer, `release(u)` returns the buffer;
  * backward of unit u (units in reverse): the same gather (FSDP reshards after the forward), the layer's weight gradients are
    written into a transient full fp32 gradient buffer as the backward produces them, and as soon as the unit is complete the
    buffer is REDUCE-SCATTERED: every rank is left with the averaged gradient of its own slice only;
  * `step()`: global-norm clip (each rank sums the squares of its slices, one 8-byte all-reduce) and AdamW on the owned slices
    -- the fused kernels of sharded.py's optimizer -- writing the updated parameter shard in its storage type.

Per-rank bytes for P parameters in bf16 units: 2P/world (shards) + 12P/world (master + moments) + 4P/world (gradient slices)
+ the pool: `prefetch + 1` full units of parameters and one of gradients (a LLaMA-7B layer: 0.4 GB bf16, 0.8 GB fp32).
Collectives per step and unit: 2 all-gathers of the parameter bytes (forward, backward) + 1 reduce-scatter of the fp32
gradient bytes, each over all 7 xGMI links (1/world of the unit per peer) -- RCCL on the node ("nccl"), gloo in
tests/test_fsdp_gloo.py.  The compute itself is unchanged: the same kernels read the gathered buffer.
"""
import torch
import torch.distributed as dist

from . import kernels as K


class _Flat:
    """The tensors of one dtype of a unit, flattened: layout, this rank's persistent slices, pointers into pool buffers."""

    def __init__(self, entries, dtype, device, world, rank):
        self.entries = entries                                   # [(name, shape)]
        self.dtype = dtype
        self.numel = sum(int(torch.Size(s).numel()) for _, s in entries)
        self.padded = self.numel + ((-self.numel) % world)
        self.shard = self.padded // world
        self.lo = rank * self.shard
        self.offsets, off = [], 0
        for _, s in entries:
            self.offsets.append(off)
            off += int(torch.Size(s).numel())
        self.param_shard = torch.zeros(self.shard, dtype=dtype, device=device)      # persistent
        self.grad_shard = torch.zeros(self.shard, dtype=torch.float32, device=device)
        self.master = None                                        # fp32 master of the shard (the shard itself when fp32)
        self.exp_avg = torch.zeros(self.shard, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(self.shard, dtype=torch.float32, device=device)
        self.full = None                                          # transient gathered parameters (a pool buffer)
        self.gfull = None                                         # transient full gradients

    def views(self, flat):
        return [flat[o:o + int(torch.Size(s).numel())].view(s) for o, (_, s) in zip(self.offsets, self.entries)]


class FullShardManager:
    """units: list (forward order) of lists [(name, live_tensor)].  `rebind(name, tensor_or_None)` re-points the tensor the
    kernels read for `name` (None when the unit is released: using a released tensor is a bug, not a stale read).
    `update_fn` (tests only) replaces the fused AdamW kernels, which need the GPU."""

    def __init__(self, units, rebind, group=None, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, prefetch=1, update_fn=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rebind = rebind
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.update_fn = update_fn
        self.prefetch = prefetch
        self.steps = 0
        dev = units[0][0][1].device
        self.device = dev
        self.units = []
        for ents in units:
            flats = []
            for dtype in (torch.bfloat16, torch.float32):
                sel = [(n, t) for n, t in ents if t.dtype == dtype]
                if not sel:
                    continue
                f = _Flat([(n, tuple(t.shape)) for n, t in sel], dtype, dev, self.world, self.rank)
                # this rank's slice of the flat image of the live tensors
                img = torch.zeros(f.padded, dtype=dtype, device=dev)
                for v, (_, t) in zip(f.views(img), sel):
                    v.copy_(t)
                f.param_shard.copy_(img[f.lo:f.lo + f.shard])
                f.master = f.param_shard if dtype == torch.float32 else f.param_shard.float()
                del img
                flats.append(f)
            assert sum(len(f.entries) for f in flats) == len(ents), "bf16 / fp32 tensors only"
            self.units.append(dict(flats=flats, names=[n for n, _ in ents], pending=0, gathered=False, event=None, gevent=None))
            for n, _ in ents:
                rebind(n, None)                                   # the full tensors are gone: only the shards persist
        self.where = {n: (ui, fi, ei) for ui, u in enumerate(self.units) for fi, f in enumerate(u["flats"])
                      for ei, (n, _) in enumerate(f.entries)}
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._pool = {}                                           # (dtype, padded numel) -> free full buffers
        self._fused = None
        self.total_sq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.peak_transient = 0
        self._live_transient = 0

    # ---- pool of transient full buffers -----------------------------------------------------------------------------------
    def _take(self, dtype, n):
        free = self._pool.setdefault((dtype, n), [])
        if free:
            buf, busy = free.pop()
            if busy is not None:                                  # its last reader ran on the OTHER stream: order behind it
                torch.cuda.current_stream(self.device).wait_event(busy)
        else:
            buf = torch.empty(n, dtype=dtype, device=self.device)
        self._live_transient += buf.numel() * buf.element_size()
        self.peak_transient = max(self.peak_transient, self._live_transient)
        return buf

    def _give(self, buf, busy=None):
        """Back to the pool; `busy` = event after which the buffer may be overwritten (a collective still reading it)."""
        self._live_transient -= buf.numel() * buf.element_size()
        self._pool[(buf.dtype, buf.numel())].append((buf, busy))

    def _on_comm(self, fn):
        """Run fn on the communication stream behind everything the compute stream has issued; returns the event to wait on."""
        if self.comm_stream is None:
            fn()
            return None
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ev)
            fn()
            done = torch.cuda.Event()
            done.record()
        return done

    # ---- parameters -----------------------------------------------------------------------------------------------------------
    def gather(self, ui):
        """Issue the all-gather of unit ui's parameter shards (asynchronously on the communication stream).  Idempotent."""
        u = self.units[ui]
        if u["gathered"]:
            return
        for f in u["flats"]:
            f.full = self._take(f.dtype, f.padded)

        def run():
            for f in u["flats"]:
                if self.world > 1:
                    dist.all_gather_into_tensor(f.full, f.param_shard, group=self.group)
                else:
                    f.full[:f.shard].copy_(f.param_shard)
        u["event"] = self._on_comm(run)
        u["gathered"] = True

    def use(self, ui):
        """The compute stream waits for unit ui's gather; the live tensors become views of the gathered buffer.  Issues the
        gathers of the next `prefetch` units in the direction of travel first (`direction` is set by forward()/backward())."""
        self.gather(ui)
        for d in range(1, self.prefetch + 1):
            nxt = ui + d * self._dir
            if 0 <= nxt < len(self.units) and nxt != self._root:
                self.gather(nxt)
        u = self.units[ui]
        if u["event"] is not None:
            torch.cuda.current_stream(self.device).wait_event(u["event"])
            u["event"] = None
        for f in u["flats"]:
            for (n, _), v in zip(f.entries, f.views(f.full)):
                self.rebind(n, v)

    def release(self, ui):
        u = self.units[ui]
        if not u["gathered"]:
            return
        for f in u["flats"]:
            for n, _ in f.entries:
                self.rebind(n, None)
            self._give(f.full)              # (the next gather into it is ordered behind this stream's work by _on_comm)
            f.full = None
        u["gathered"] = False

    _dir, _root = 1, -1

    def direction(self, d, root=-1):
        """+1 while the forward walks the units, -1 in the backward; `root` = a unit that stays gathered for the whole step."""
        self._dir, self._root = d, root

    # ---- gradients --------------------------------------------------------------------------------------------------------------
    def begin_step(self):
        for u in self.units:
            u["pending"] = len(u["names"])
            u["gevent"] = None

    def _gfull(self, f):
        if f.gfull is None:
            f.gfull = self._take(torch.float32, f.padded)
            if f.padded != f.numel:
                f.gfull[f.numel:].zero_()
        return f.gfull

    def grad_slot(self, name):
        """The view of the unit's transient full gradient buffer that receives `name`'s gradient, for producers that can
        write their result in place (no copy in grad_ready); None for an unknown name."""
        if name not in self.where:
            return None
        ui, fi, ei = self.where[name]
        f = self.units[ui]["flats"][fi]
        return f.views(self._gfull(f))[ei]

    def grad_ready(self, name, grad):
        """Write `grad` into the unit's transient full gradient buffer (unless it was produced there); reduce-scatter the
        unit when it is complete.  Returns True: the caller may drop its tensor."""
        ui, fi, ei = self.where[name]
        u = self.units[ui]
        f = u["flats"][fi]
        dst = f.views(self._gfull(f))[ei]
        if grad.data_ptr() != dst.data_ptr() or grad.dtype != dst.dtype:
            dst.copy_(grad.reshape(f.entries[ei][1]))
        u["pending"] -= 1
        if u["pending"] == 0:
            self._reduce_unit(u)
        return True

    def _reduce_unit(self, u):
        flats = [f for f in u["flats"] if f.gfull is not None]
        assert len(flats) == len(u["flats"]), "a unit's gradients arrive together"

        def run():
            for f in flats:
                if self.world > 1:
                    dist.reduce_scatter_tensor(f.grad_shard, f.gfull, op=dist.ReduceOp.SUM, group=self.group)
                    f.grad_shard.div_(self.world)
                else:
                    f.grad_shard.copy_(f.gfull[f.lo:f.lo + f.shard])
        u["gevent"] = self._on_comm(run)
        for f in flats:
            self._give(f.gfull, busy=u["gevent"])                 # the reduce-scatter reads it on the communication stream
            f.gfull = None

    # ---- update -------------------------------------------------------------------------------------------------------------------
    def step(self, lr, max_grad_norm=None):
        for u in self.units:
            if u["pending"] != 0:
                raise RuntimeError(f"{u['pending']} gradients of a unit were never reported")
            if u["gevent"] is not None:
                torch.cuda.current_stream(self.device).wait_event(u["gevent"])
                u["gevent"] = None
        self.steps += 1
        flats = [f for u in self.units for f in u["flats"]]
        clip = max_grad_norm is not None and max_grad_norm > 0
        total = None
        if self.update_fn is not None:
            if clip:
                self.total_sq.copy_(sum((f.grad_shard.double() ** 2).sum() for f in flats).reshape(1))
                if self.world > 1:
                    dist.all_reduce(self.total_sq, group=self.group)
                total = self.total_sq
            for f in flats:
                self.update_fn(f, lr, self.steps, self.betas, self.eps, self.weight_decay, total, max_grad_norm)
        else:
            if self._fused is None:
                self._fused = K.MultiTensorAdamW([f.master for f in flats],
                                                 [(f.param_shard if f.dtype == torch.bfloat16 else None) for f in flats],
                                                 self.betas, self.eps, self.weight_decay)
                for f, m, v in zip(flats, self._fused.exp_avg, self._fused.exp_avg_sq):
                    m.copy_(f.exp_avg)
                    v.copy_(f.exp_avg_sq)
                    f.exp_avg, f.exp_avg_sq = m, v
                self._fused.steps = self.steps - 1
            gs = [f.grad_shard for f in flats]
            if clip:
                total = self._fused.grad_norm_sq(gs)
                if self.world > 1:
                    dist.all_reduce(total, group=self.group)
            self._fused.step(gs, lr, max_grad_norm if clip else None, total_sq=total)
        return total if clip else None

    # ---- accounting -----------------------------------------------------------------------------------------------------------------
    def memory(self):
        """(persistent bytes on this rank, bytes of the same state unsharded, peak transient bytes of the pool so far)."""
        flats = [f for u in self.units for f in u["flats"]]
        esz = {torch.bfloat16: 2, torch.float32: 4}
        own = sum(f.shard * (esz[f.dtype] + 4 + 8 + (4 if f.dtype == torch.bfloat16 else 0)) for f in flats)
        full = sum(f.numel * (esz[f.dtype] + 4 + 8 + (4 if f.dtype == torch.bfloat16 else 0)) for f in flats)
        return own, full, self.peak_transient

    # ---- this rank's shards, for checkpoints (torch FSDP's sharded state dict: resumable on the same (rank, world) layout) ----
    def shard_state(self):
        units = []
        for u in self.units:
            units.append([dict(names=[n for n, _ in fl.entries], dtype=str(fl.dtype), param_shard=fl.param_shard.clone(),
                               master=fl.master.clone(), exp_avg=fl.exp_avg.clone(), exp_avg_sq=fl.exp_avg_sq.clone())
                          for fl in u["flats"]])
        return {"rank": self.rank, "world": self.world, "units": units}

    def load_shard_state(self, sd):
        assert sd["world"] == self.world and sd["rank"] == self.rank, "the sharded state is per (rank, world)"
        assert len(sd["units"]) == len(self.units)
        for u, su in zip(self.units, sd["units"]):
            assert len(u["flats"]) == len(su)
            for fl, s in zip(u["flats"], su):
                assert [n for n, _ in fl.entries] == s["names"] and str(fl.dtype) == s["dtype"]
                fl.param_shard.copy_(s["param_shard"])
                if fl.master is not fl.param_shard:
                    fl.master.copy_(s["master"])
                fl.exp_avg.copy_(s["exp_avg"])                 # (after the first step these ARE the fused optimizer's moments)
                fl.exp_avg_sq.copy_(s["exp_avg_sq"])

    def full_state(self, ui):
        """name -> full tensor of unit ui (gathers it; for checkpoints / tests).  The caller must release(ui)."""
        self.direction(1)
        keep, self.prefetch = self.prefetch, 0                    # this unit only: no look-ahead gather left behind
        self.use(ui)
        self.prefetch = keep
        u = self.units[ui]
        return {n: v for f in u["flats"] for (n, _), v in zip(f.entries, f.views(f.full))}
