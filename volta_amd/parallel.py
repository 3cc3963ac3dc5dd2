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
  wire = "bf16"        (either mode) the bucket crosses the links as bf16 -- half the bytes -- but is summed in fp32:
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

    def __init__(self, flat, process_group=None, mode=None, wire=None):
        self.flat, self.pg = flat, process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.cuda = flat.is_cuda
        self.stream = torch.cuda.Stream(device=flat.device) if self.cuda else None
        self.use_avg = self.cuda and dist.get_backend(process_group) == "nccl"
        self.mode = mode or os.environ.get("VK_DDP_MODE", "allreduce")
        self.wire = wire or os.environ.get("VK_DDP_WIRE", "fp32")
        if self.mode not in ("allreduce", "rs_ag") or self.wire not in ("fp32", "bf16"):
            raise ValueError("BucketReducer: mode %r / wire %r (allreduce | rs_ag, fp32 | bf16)" % (self.mode, self.wire))
        self.bytes_on_wire = 0          # per step, sent by this rank (algorithmic: 2 (w-1)/w x bucket bytes for either mode)
        self._ws = {}

    def reduce(self, ranges, join=None):
        """`join`: called with the communication stream current, to make IT (not the compute stream) wait for other
        producers of the bucket (the executor's weight-gradient side stream)."""
        if self.cuda:
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
            self.bytes_on_wire += 2 * (w - 1) * n * esz // w
            if self.wire == "bf16" and sharded:
                self._reduce_bf16(view, n)
            elif self.mode == "rs_ag" and sharded:
                shard = view[self.rank * (n // w):(self.rank + 1) * (n // w)]
                if self.use_avg:
                    dist.reduce_scatter_tensor(shard, view, op=dist.ReduceOp.AVG, group=self.pg)
                else:
                    tmp = self._buf("rs", n // w, view.dtype)
                    dist.reduce_scatter_tensor(tmp, view, op=dist.ReduceOp.SUM, group=self.pg)
                    shard.copy_(tmp).mul_(1.0 / w)
                dist.all_gather_into_tensor(view, shard, group=self.pg)
            elif self.use_avg:
                dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg)
            else:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
                view.mul_(1.0 / w)

    def _reduce_bf16(self, view, n):
        """bf16 on the links, fp32 sum: piece j of every rank's bucket goes to rank j (all-to-all), is summed there in fp32,
        and the averaged shard comes back through an all-gather in bf16."""
        w, s = self.world, n // self.world
        send = self._buf("a2a_send", n, torch.bfloat16)
        send.copy_(view)
        recv = self._buf("a2a_recv", n, torch.bfloat16)
        dist.all_to_all_single(recv, send, group=self.pg)
        mean = recv.view(w, s).float().sum(0).mul_(1.0 / w)
        shard = self._buf("ag_shard", s, torch.bfloat16)
        shard.copy_(mean)
        dist.all_gather_into_tensor(send, shard, group=self.pg)
        view.copy_(send)

    def finish(self):
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)


class DistributedDataParallel(nn.Module):
    def __init__(self, module, message_size=10000000, process_group=None, mode=None, wire=None, **unused):
        super().__init__()
        self.module = module
        self.message_size = message_size          # elements per bucket, as apex's argument (distributed.py:164)
        self.pg = process_group
        arena = module.materialize()
        dist.broadcast(arena.master, 0, group=process_group)
        arena.refresh_shadow(force=True)
        self.reducer = BucketReducer(arena.grad, process_group, mode=mode, wire=wire)
        module.__dict__["_ddp"] = self

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def _plan(self, eng):
        plan = getattr(eng, "_ddp_plan", None)      # lives and dies with the engine it describes
        if plan is None or plan[0] != self.message_size:
            arena = eng.arena
            spans = {}
            for n in arena.params:
                numel = 1
                for d in arena.shape[n]:
                    numel *= d
                spans[n] = (arena.offset[n], numel)
            n_stages = len(eng.bwd_marks)
            buckets = plan_buckets(spans, eng.param_ready_stage, n_stages, self.message_size * 4, pad_to=SLOT, total=arena.total)
            plan = (self.message_size, [(eng.bwd_marks[s], ranges) for s, ranges in buckets])
            eng._ddp_plan = plan
        return plan[1]

    def run_backward(self, eng):
        start = 0
        self.reducer.bytes_on_wire = 0
        owner = torch.cuda.current_stream().cuda_stream if self.reducer.cuda else None      # the executor keeps one side stream per caller stream
        for end, ranges in self._plan(eng):
            eng.bwd.run(start, end)
            # the bucket's weight gradients were computed on the executor's side stream: the communication stream waits
            # for them, the compute stream carries on with the backward
            self.reducer.reduce(ranges, join=lambda: eng.bwd.join_side(owner))
            start = end
        eng.bwd.run(start, len(eng.bwd.ops))
        self.reducer.finish()
