"""Data-parallel gradient averaging over RCCL / xGMI (one process per GPU), replacing apex DDP
(apex/apex/parallel/distributed.py:129-639) as used by train_concap.py:246-253.

Differences that follow from the engine's design, not from the collective library:
  * gradients already live in one flat fp32 arena, so a bucket is a RANGE of it: no flatten / unflatten /
    multi_tensor_scale copies (distributed.py:425-475), no first-iteration bucket discovery (:367-390);
  * the backward command list is cut at sub-layer boundaries; after each cut an event is recorded and the
    bucket's reduction is queued on a side stream, overlapping the rest of the backward (:513-556);
  * a bucket is ONE contiguous collective: parameter slots are 1024-element aligned and their padding
    gradients are always zero, so every span is extended to its slot end and neighbouring slots coalesce.
Parameters are broadcast from rank 0 at wrap time as ONE flat buffer (distributed.py:253).

Reduction modes (`mode=` / VK_DDP_MODE, `wire=` / VK_DDP_WIRE):
  allreduce (default)  one `all_reduce(AVG)` per bucket -- apex's semantics (:451-454) with the library's algorithm choice;
  rs_ag                `reduce_scatter_tensor(AVG)` into this rank's 1/world shard of the bucket, then
                       `all_gather_into_tensor` back: the two halves of a direct all-reduce issued separately, so
                       every rank exchanges its shard with all peers at once over the 7 xGMI links;
  zero1                the reduce-scatter half of rs_ag only: each rank keeps the averaged gradient of ITS 1/world shard of every bucket
                       (shards are whole 1024-element slots; the few slots a bucket does not divide into are all-reduced and stay
                       replicated), `volta_amd.AdamW` then steps that shard alone -- 1/world of the optimizer's 30 bytes per parameter and
                       of its two moment arenas' traffic -- and the updated fp32 MASTER weights are all-gathered (the bf16 copies the
                       GEMMs read are re-cast locally).  Bytes on the links equal rs_ag's; what shrinks is the optimizer pass.  The
                       gradient norm is shard-decomposable (per-slot sums of squares, vk_grad_sqnorm_chunks): replicas and the
                       unsharded path agree bit for bit.  Moments of foreign shards are not kept up to date: `AdamW.state_dict()` gathers them;
  wire = "bf16"        (allreduce / rs_ag) the bucket crosses the links as bf16 -- half the bytes -- but is summed in fp32:
                       every rank receives its shard's `world` bf16 pieces (`all_to_all_single`), adds them in fp32,
                       and the averaged shard is gathered back as bf16.  Off by default: the reference averages in fp32."""
import os

import torch
import torch.distributed as dist
from torch import nn

SLOT = 1024        # parameter slots of the flat arenas are aligned to this many elements (engine.CHUNK)


def plan_buckets(spans, ready, n_stages, cap_bytes, elem_bytes=4, pad_to=1, total=None):
    """spans: {name: (offset, numel)} in one flat arena; ready[name]: index of the backward stage after which
    the gradient of `name` is final.  Returns [(stage, [(lo, hi), ...])]: after backward stage `stage` the
    listed element ranges (coalesced, each a bucket of >= cap_bytes except the last) can be reduced.
    `pad_to`: every span end is rounded up to this multiple (slot padding holds zero gradients), capped at `total`."""
    by_stage = {}
    for name, (off, n) in spans.items():
        hi = off + n
        if pad_to > 1:
            hi = -(-hi // pad_to) * pad_to
            if total is not None:
                hi = min(hi, total)
        by_stage.setdefault(ready[name], []).append((off, hi))
    out, pending, size = [], [], 0
    for s in range(n_stages):
        for r in by_stage.get(s, []):
            pending.append(r)
            size += (r[1] - r[0]) * elem_bytes
        if pending and (size >= cap_bytes or s == n_stages - 1):
            pending.sort()
            merged = [list(pending[0])]
            for lo, hi in pending[1:]:
                if lo <= merged[-1][1]:
                    merged[-1][1] = max(merged[-1][1], hi)
                else:
                    merged.append([lo, hi])
            out.append((s, [tuple(m) for m in merged]))
            pending, size = [], 0
    return out


class BucketReducer:
    """Averages ranges of a flat gradient tensor across the process group, asynchronously on CUDA."""

    def __init__(self, flat, process_group=None, mode=None, wire=None, gradient_average=True, gradient_predivide_factor=1.0):
        self.flat, self.pg = flat, process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.cuda = flat.is_cuda
        self.stream = None
        if self.cuda:
            # a stream on a hardware queue of its own: one that shares the queue of the compute stream or of the weight-gradient side stream
            # serialises them behind its waits (volta_amd/streams.py)
            from .streams import independent_stream
            with torch.cuda.device(flat.device):
                self.stream = independent_stream(device=flat.device)
        # apex's scaling (distributed.py:445-454): g *= 1 / predivide, sum over ranks, g *= predivide / world unless gradient_average is off.
        # The default (average, predivide 1) is one AVG collective; anything else sums and scales explicitly.
        self.pre_scale = 1.0 / float(gradient_predivide_factor)
        self.post_scale = float(gradient_predivide_factor) / self.world if gradient_average else 1.0
        self.plain_mean = bool(gradient_average) and float(gradient_predivide_factor) == 1.0
        self.use_avg = self.cuda and dist.get_backend(process_group) == "nccl" and self.plain_mean
        self.mode = mode or os.environ.get("VK_DDP_MODE", "allreduce")
        self.wire = wire or os.environ.get("VK_DDP_WIRE", "fp32")
        if self.mode not in ("allreduce", "rs_ag", "zero1") or self.wire not in ("fp32", "bf16"):
            raise ValueError("BucketReducer: mode %r / wire %r (allreduce | rs_ag | zero1, fp32 | bf16)" % (self.mode, self.wire))
        if self.mode == "zero1" and self.wire != "fp32":
            raise ValueError("BucketReducer: mode zero1 reduces and gathers in fp32")
        self.bytes_on_wire = 0          # per step, sent by this rank (algorithmic: 2 (w-1)/w x bucket bytes for either mode)
        self._ws = {}
        # "gate": flags + one-wave gates instead of stream events (vk_store_u64 / vk_gate_value).  Measured equal to the event form once the
        # communication stream is clear of the compute stream's pipe (profiles/r04_experiments.md); kept as a switch
        self.gated = os.environ.get("VK_DDP_ORDER", "event") == "gate"
        self._epoch = 0
        if self.cuda:
            self._flags = torch.zeros(2 * 256, dtype=torch.int64, device=flat.device)       # (compute, side) flag per bucket of a pass
            self._err = torch.zeros(1, dtype=torch.int32, device=flat.device)
        self.begin_step()

    def begin_step(self):
        """zero1 bookkeeping of one backward: `sharded` = [(lo, hi, shard)] ranges whose rank-th piece of `shard` elements this rank owns,
        `replicated` = [(lo, hi)] ranges every rank holds (and steps) in full."""
        self.bytes_on_wire = 0
        self.sharded, self.replicated = [], []
        self._epoch += 1                # what this pass's flags are set to and its gates wait for
        self._bucket = 0

    def owned(self):
        """Element ranges of the flat arena this rank's optimizer steps under zero1, in arena order."""
        return sorted([(lo + self.rank * s, lo + (self.rank + 1) * s) for lo, hi, s in self.sharded] + list(self.replicated))

    def gather(self, flat, layout=None):
        """All-gather the owned shards of a tensor laid out like the gradient arena (updated master weights; optimizer moments for a
        checkpoint): afterwards every rank holds the whole tensor.  `layout`: a `sharded` list of an earlier step (default: the current one)."""
        for lo, hi, s in (self.sharded if layout is None else layout):
            dist.all_gather_into_tensor(flat[lo:hi], flat[lo + self.rank * s:lo + (self.rank + 1) * s], group=self.pg)
            self.bytes_on_wire += (self.world - 1) * s * flat.element_size()

    def reduce(self, ranges, join=None, side=None):
        """Queue the reduction of `ranges` on the communication stream, behind (a) everything enqueued so far on the compute stream and
        (b) the other producer of the bucket, the executor's weight-gradient side stream.
        `side`: that stream's raw handle -- the ordering is then made of FLAGS and GATES (vk_store_u64 on the compute and the side stream,
        vk_gate_value on the communication stream), not of stream events: an unsatisfied `wait_event` enqueued a step ahead sits at the
        head of the communication queue for milliseconds and stalls the queues that share its command-processor pipe (measured: 19-25 ms
        per step instead of 16.9, by which pool stream the reducer happened to get; profiles/r04_experiments.md).
        `join` (without `side`): the event form -- called with the communication stream current, to make it wait for the side stream."""
        if self.cuda and side is not None and self.gated:
            import ctypes as C
            from . import _lib as L
            k = self._bucket
            self._bucket += 1
            if k >= self._flags.numel() // 2:
                raise RuntimeError("BucketReducer: more than %d buckets in one backward pass" % (self._flags.numel() // 2))
            f_main, f_side = self._flags.data_ptr() + 16 * k, self._flags.data_ptr() + 16 * k + 8
            L.check(L.lib.vk_store_u64(C.c_void_p(f_main), self._epoch, L.stream_ptr()))
            L.check(L.lib.vk_store_u64(C.c_void_p(f_side), self._epoch, C.c_void_p(side)))
            with torch.cuda.stream(self.stream):
                L.check(L.lib.vk_gate_value(C.c_void_p(f_main), self._epoch, 10000000, L.ptr(self._err), L.stream_ptr()))
                L.check(L.lib.vk_gate_value(C.c_void_p(f_side), self._epoch, 10000000, L.ptr(self._err), L.stream_ptr()))
                self._reduce(ranges)
        elif self.cuda:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                if join is not None:
                    join()
                self._reduce(ranges)
        else:
            if join is not None:
                join()
            self._reduce(ranges)

    def _buf(self, key, n, dtype):
        t = self._ws.get(key)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(n, dtype=dtype, device=self.flat.device)
            self._ws[key] = t
        return t[:n]

    def _reduce(self, ranges):
        w = self.world
        for lo, hi in ranges:
            view = self.flat[lo:hi]
            n = hi - lo
            sharded = w > 1 and n % w == 0 and n >= w
            esz = 2 if (self.wire == "bf16" and sharded) else 4
            if self.mode != "zero1":
                self.bytes_on_wire += 2 * (w - 1) * n * esz // w
            if self.pre_scale != 1.0:
                view.mul_(self.pre_scale)
            if self.mode == "zero1":
                self._reduce_zero1(lo, hi)
                continue
            if self.wire == "bf16" and sharded:
                self._reduce_bf16(view, n)
            elif self.mode == "rs_ag" and sharded:
                shard = view[self.rank * (n // w):(self.rank + 1) * (n // w)]
                if self.use_avg:
                    dist.reduce_scatter_tensor(shard, view, op=dist.ReduceOp.AVG, group=self.pg)
                else:
                    tmp = self._buf("rs", n // w, view.dtype)
                    dist.reduce_scatter_tensor(tmp, view, op=dist.ReduceOp.SUM, group=self.pg)
                    shard.copy_(tmp)
                    if self.post_scale != 1.0:
                        shard.mul_(self.post_scale)
                dist.all_gather_into_tensor(view, shard, group=self.pg)
            elif self.use_avg:
                dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg)
            else:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
                if self.post_scale != 1.0:
                    view.mul_(self.post_scale)

    def _reduce_zero1(self, lo, hi):
        """Reduce-scatter of whole slots: the first world * shard elements of the range are cut into `world` equal pieces of whole
        1024-element slots, the rest (fewer than `world` slots) is all-reduced and stays replicated."""
        w, n = self.world, hi - lo
        s = (n // (w * SLOT)) * SLOT
        main = w * s
        if s > 0:
            view = self.flat[lo:lo + main]
            shard = view[self.rank * s:(self.rank + 1) * s]
            if self.use_avg:
                dist.reduce_scatter_tensor(shard, view, op=dist.ReduceOp.AVG, group=self.pg)
            else:
                tmp = self._buf("rs", s, view.dtype)
                dist.reduce_scatter_tensor(tmp, view, op=dist.ReduceOp.SUM, group=self.pg)
                shard.copy_(tmp)
                if self.post_scale != 1.0:
                    shard.mul_(self.post_scale)
            self.sharded.append((lo, lo + main, s))
            self.bytes_on_wire += (w - 1) * s * 4
        if main < n:
            rest = self.flat[lo + main:hi]
            if self.use_avg:
                dist.all_reduce(rest, op=dist.ReduceOp.AVG, group=self.pg)
            else:
                dist.all_reduce(rest, op=dist.ReduceOp.SUM, group=self.pg)
                if self.post_scale != 1.0:
                    rest.mul_(self.post_scale)
            self.replicated.append((lo + main, hi))
            self.bytes_on_wire += 2 * (w - 1) * (n - main) * 4 // w

    def _reduce_bf16(self, view, n):
        """bf16 on the links, fp32 sum: piece j of every rank's bucket goes to rank j (all-to-all), is summed there in fp32,
        and the averaged shard comes back through an all-gather in bf16."""
        w, s = self.world, n // self.world
        send = self._buf("a2a_send", n, torch.bfloat16)
        send.copy_(view)
        recv = self._buf("a2a_recv", n, torch.bfloat16)
        dist.all_to_all_single(recv, send, group=self.pg)
        mean = recv.view(w, s).float().sum(0).mul_(self.post_scale)
        shard = self._buf("ag_shard", s, torch.bfloat16)
        shard.copy_(mean)
        dist.all_gather_into_tensor(send, shard, group=self.pg)
        view.copy_(send)

    def finish(self):
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)


class DistributedDataParallel(nn.Module):
    """apex.parallel.DistributedDataParallel's constructor (apex/apex/parallel/distributed.py:151-173), argument by argument:
      message_size               honoured: minimum elements per bucket;
      delay_allreduce            honoured: True reduces the whole arena once, after the backward (train_task.py:253);
      gradient_average,
      gradient_predivide_factor  honoured with apex's arithmetic (:445-454): g *= 1 / predivide before the sum, g *= predivide / world after
                                 it unless gradient_average is False;
      allreduce_always_fp32      accepted: gradients are fp32 here and are summed in fp32 under every mode / wire;
      shared_param               ValueError, as in apex (:196-197);
      allreduce_trigger_params, retain_allreduce_buffers, num_allreduce_streams > 1, allreduce_communicators,
      gradient_average_split_factor, prof
                                 NotImplementedError when set: buckets are ranges of the engine's gradient arena cut at sub-layer
                                 boundaries, there are no per-parameter hooks, no separate allreduce buffers and one communication stream.
    `mode` / `wire` / `process_group` are this implementation's own (module docstring)."""

    def __init__(self, module, message_size=10000000, delay_allreduce=False, shared_param=None, allreduce_trigger_params=None,
                 retain_allreduce_buffers=False, allreduce_always_fp32=False, num_allreduce_streams=1, allreduce_communicators=None,
                 gradient_average=True, gradient_predivide_factor=1.0, gradient_average_split_factor=None, prof=False,
                 process_group=None, mode=None, wire=None):
        super().__init__()
        if shared_param is not None:
            raise ValueError("shared_param is no longer supported as an option.  It was misleadingly named from the start.  It turns out overlapping "
                             "communication with computation should work fine with shared parameters.  If you still wish to delay communication to "
                             "the end of the backward pass, use delay_allreduce=True|False instead.")
        unsupported = dict(allreduce_trigger_params=allreduce_trigger_params is not None, retain_allreduce_buffers=bool(retain_allreduce_buffers),
                           num_allreduce_streams=num_allreduce_streams != 1, allreduce_communicators=allreduce_communicators is not None,
                           gradient_average_split_factor=gradient_average_split_factor is not None, prof=bool(prof))
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError("volta_amd.DistributedDataParallel: %s not supported (buckets are ranges of the engine's gradient arena)" % ", ".join(bad))
        if not float(gradient_predivide_factor) > 0.0:
            raise ValueError("gradient_predivide_factor must be positive")
        self.module = module
        self.message_size = message_size          # elements per bucket, as apex's argument (distributed.py:164)
        self.delay_allreduce = bool(delay_allreduce)
        self.pg = process_group
        arena = module.materialize()
        dist.broadcast(arena.master, 0, group=process_group)
        arena.refresh_shadow(force=True)
        self.reducer = BucketReducer(arena.grad, process_group, mode=mode, wire=wire, gradient_average=gradient_average,
                                     gradient_predivide_factor=gradient_predivide_factor)
        module.__dict__["_ddp"] = self

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    @staticmethod
    def _side_handle(owner):
        """Raw handle of the executor's weight-gradient side stream of compute stream `owner` (None on CPU)."""
        if owner is None:
            return None
        import ctypes as C
        from . import _lib as L
        h = L.lib.vk_side_stream(C.c_void_p(owner))
        if not h:
            raise RuntimeError(L.lib.vk_last_error().decode())
        return int(h)

    def _plan(self, eng):
        plan = getattr(eng, "_ddp_plan", None)      # lives and dies with the engine it describes
        key = (self.message_size, self.delay_allreduce)
        if plan is None or plan[0] != key:
            arena = eng.arena
            spans = {}
            for n in arena.params:
                numel = 1
                for d in arena.shape[n]:
                    numel *= d
                spans[n] = (arena.offset[n], numel)
            n_stages = len(eng.bwd_marks)
            cap = (1 << 62) if self.delay_allreduce else self.message_size * 4      # delay_allreduce: one bucket, cut after the last stage
            buckets = plan_buckets(spans, eng.param_ready_stage, n_stages, cap, pad_to=SLOT, total=arena.total)
            plan = (key, [(eng.bwd_marks[s], ranges) for s, ranges in buckets])
            eng._ddp_plan = plan
        return plan[1]

    def gather_params(self, arena):
        """zero1, after the optimizer stepped this rank's shards: every rank receives the other shards' fp32 master weights and re-casts
        the bf16 copies the GEMMs read (the owner's copy came out of its AdamW launch already)."""
        from . import _lib as L
        red = self.reducer
        red.gather(arena.master)
        for lo, hi, s in red.sharded:
            L.check(L.lib.vk_cast_f32_bf16(arena.master.data_ptr() + 4 * lo, arena.shadow.data_ptr() + 2 * lo, hi - lo, L.stream_ptr()))

    def run_backward(self, eng):
        if self.reducer.mode == "zero1":
            # after a sharded reduction the gradient arena holds the AVERAGE only inside this rank's shards: any consumer other than
            # volta_amd.AdamW / clip_grad_norm_ (which know the shard layout) would step on un-averaged local gradients and nothing would
            # gather the parameters -- refuse to start such a step
            opt = getattr(eng.arena, "_vk_adamw", None)
            if opt is None or opt() is None:
                raise RuntimeError("DistributedDataParallel(mode='zero1') shards the gradient reduction for volta_amd.optimization.AdamW: build that "
                                   "optimizer on the model's parameters (and run one step, or call optimizer._setup()) before the first backward")
        start = 0
        self.reducer.begin_step()
        owner = torch.cuda.current_stream().cuda_stream if self.reducer.cuda else None      # the executor keeps one side stream per caller stream
        for end, ranges in self._plan(eng):
            eng.bwd.run(start, end)
            # the bucket's weight gradients were computed on the executor's side stream: the communication stream waits
            # for them, the compute stream carries on with the backward
            self.reducer.reduce(ranges, join=lambda: eng.bwd.join_side(owner), side=self._side_handle(owner))
            start = end
        eng.bwd.run(start, len(eng.bwd.ops))
        self.reducer.finish()
