"""Data-parallel gradient averaging over RCCL / xGMI (one process per GPU), replacing apex DDP
(apex/apex/parallel/distributed.py:129-639) as used by train_concap.py:246-253.

Differences that follow from the engine's design, not from the collective library:
  * gradients already live in one flat fp32 arena, so a bucket is a RANGE of it: no flatten / unflatten /
    multi_tensor_scale copies (distributed.py:425-475), no first-iteration bucket discovery (:367-390);
  * the backward command list is cut at sub-layer boundaries; after each cut an event is recorded and the
    bucket's all-reduce is queued on a side stream, overlapping the rest of the backward (:513-556);
  * averaging uses the collective's AVG reduction where the backend has it (RCCL), else SUM + scale.
Parameters are broadcast from rank 0 at wrap time as ONE flat buffer (distributed.py:253)."""
import torch
import torch.distributed as dist
from torch import nn


def plan_buckets(spans, ready, n_stages, cap_bytes, elem_bytes=4):
    """spans: {name: (offset, numel)} in one flat arena; ready[name]: index of the backward stage after which
    the gradient of `name` is final.  Returns [(stage, [(lo, hi), ...])]: after backward stage `stage` the
    listed element ranges (coalesced, each a bucket of >= cap_bytes except the last) can be all-reduced."""
    by_stage = {}
    for name, (off, n) in spans.items():
        by_stage.setdefault(ready[name], []).append((off, off + n))
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

    def __init__(self, flat, process_group=None):
        self.flat, self.pg = flat, process_group
        self.world = dist.get_world_size(process_group)
        self.cuda = flat.is_cuda
        self.stream = torch.cuda.Stream(device=flat.device) if self.cuda else None
        self.use_avg = self.cuda and dist.get_backend(process_group) == "nccl"

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

    def _reduce(self, ranges):
        for lo, hi in ranges:
            view = self.flat[lo:hi]
            if self.use_avg:
                dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.pg)
            else:
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg)
                view.mul_(1.0 / self.world)

    def finish(self):
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)


class DistributedDataParallel(nn.Module):
    def __init__(self, module, message_size=10000000, process_group=None, **unused):
        super().__init__()
        self.module = module
        self.message_size = message_size          # elements per bucket, as apex's argument (distributed.py:164)
        self.pg = process_group
        arena = module.materialize()
        dist.broadcast(arena.master, 0, group=process_group)
        arena.refresh_shadow(force=True)
        self.reducer = BucketReducer(arena.grad, process_group)
        self._plans = {}
        module.__dict__["_ddp"] = self

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def _plan(self, eng):
        key = id(eng)
        if key not in self._plans:
            arena = eng.arena
            spans = {}
            for n in arena.params:
                numel = 1
                for d in arena.shape[n]:
                    numel *= d
                spans[n] = (arena.offset[n], numel)
            n_stages = len(eng.bwd_marks)
            buckets = plan_buckets(spans, eng.param_ready_stage, n_stages, self.message_size * 4)
            self._plans[key] = [(eng.bwd_marks[s], ranges) for s, ranges in buckets]
        return self._plans[key]

    def run_backward(self, eng):
        start = 0
        for end, ranges in self._plan(eng):
            eng.bwd.run(start, end)
            # the bucket's weight gradients were computed on the executor's side stream: the communication stream waits
            # for them, the compute stream carries on with the backward
            self.reducer.reduce(ranges, join=eng.bwd.join_side)
            start = end
        eng.bwd.run(start, len(eng.bwd.ops))
        self.reducer.finish()
