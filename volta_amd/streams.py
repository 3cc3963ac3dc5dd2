"""Streams that really run beside each other.

HIP multiplexes its streams onto a few hardware queues (ROCclr: GPU_MAX_HW_QUEUES, 4 by default; PyTorch alone opens 32 pool streams per
priority), and the hardware queues onto the command processor's pipes.  Measured on MI355X (profiles/r04_experiments.md, tools/probe_streams.py):
  * two streams on ONE hardware queue are processed in enqueue order -- a kernel of the one holds back every later launch of the other;
  * two streams on different queues of ONE pipe take turns at the pipe: while the one is active (a running kernel, or an unsatisfied
    `wait_event` at the head of its queue) every launch of the other takes 1.6x as long (3.7 -> 6 us).  For a helper stream that is
    busy or waiting for the whole step -- the gradient reducer's communication stream, a pipelined optimizer -- on the pipe of the
    COMPUTE stream that is not a detail: 24 ms per step instead of 16.9 (the launches of the backward's dgrads fall behind the weight
    gradients of the side stream, and both run at half speed beside each other), in 6 of 10 `torch.cuda.Stream()` draws.
So the streams the engine opens beside a command list are not taken blindly: `independent_stream()` PROBES candidates against the streams
they must not disturb -- a train of tiny launches on the one while a one-wave kernel runs on the other -- and keeps the first candidate
that neither delays them nor is delayed by them."""
import ctypes as C
import os
import time
import warnings

import torch

from . import _lib as L

HOLD_US = 2000          # how long the probe holds the stream under test


def _raw(s):
    return s if isinstance(s, int) else int(s.cuda_stream)


def shares_queue(a, b):
    """True when work enqueued on stream `b` (torch stream or raw handle) waits for work enqueued earlier on stream `a`."""
    a, b = _raw(a), _raw(b)
    torch.cuda.synchronize()
    L.check(L.lib.vk_hold_cus(1, HOLD_US, 0, C.c_void_p(a)))
    L.check(L.lib.vk_hold_cus(1, 1, 0, C.c_void_p(b)))
    ev = torch.cuda.Event()
    t0 = time.perf_counter()
    with torch.cuda.stream(torch.cuda.ExternalStream(b)) if b else _null():
        ev.record()
    ev.synchronize()
    dt = (time.perf_counter() - t0) * 1e6
    torch.cuda.synchronize()
    return dt > HOLD_US * 0.5


def active_cost(victim, s, n=300, hold_us=3000):
    """Time (us) of a train of `n` tiny launches on stream `victim` while a one-wave kernel runs on stream `s` for `hold_us`, against
    the same train with `s` idle: (busy, idle).  Queues that share a command-processor pipe take turns at it: every launch of the one
    waits for the pipe while the other is active (measured: 3.7 -> 8-11 us per launch)."""
    victim, sr = _raw(victim), _raw(s)
    out = []
    for busy in (True, False, True, False):
        torch.cuda.synchronize()
        if busy:
            L.check(L.lib.vk_hold_cus(1, hold_us, 0, C.c_void_p(sr)))
        done = torch.cuda.Event()
        t0 = time.perf_counter()
        for _ in range(n):
            L.check(L.lib.vk_hold_cus(1, 1, 0, C.c_void_p(victim)))
        with torch.cuda.stream(torch.cuda.ExternalStream(victim)) if victim else _null():
            done.record()
        done.synchronize()
        out.append((time.perf_counter() - t0) * 1e6)
        torch.cuda.synchronize()
    return min(out[0], out[2]), min(out[1], out[3])


def blocked_wait_cost(victim, s, helper=None, n=300):
    """Time (us) of a train of `n` tiny launches on stream `victim` while stream `s` sits on an unsatisfied `wait_event`, against the same
    train with nothing pending: (blocked, free).  A stream whose pending wait slows another stream's dispatch shares more than a queue
    with it (the command processor's pipe); it must not carry the early-enqueued waits of a gradient reducer or a pipelined optimizer."""
    victim = _raw(victim)
    helper = helper or torch.cuda.Stream()
    out = []
    for blocked in (True, False, True, False):
        torch.cuda.synchronize()
        if blocked:
            L.check(L.lib.vk_hold_cus(1, 3000, 0, C.c_void_p(_raw(helper))))
            ev = torch.cuda.Event()
            ev.record(helper)
            s.wait_event(ev)
        done = torch.cuda.Event()
        t0 = time.perf_counter()
        for _ in range(n):
            L.check(L.lib.vk_hold_cus(1, 1, 0, C.c_void_p(victim)))
        with torch.cuda.stream(torch.cuda.ExternalStream(victim)) if victim else _null():
            done.record()
        done.synchronize()
        out.append((time.perf_counter() - t0) * 1e6)
        torch.cuda.synchronize()
    return min(out[0], out[2]), min(out[1], out[3])


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def engine_streams(owner=None):
    """Raw handles a new helper stream must stay clear of: the stream the command lists run on and its weight-gradient side stream."""
    own = torch.cuda.current_stream().cuda_stream if owner is None else _raw(owner)
    side = L.lib.vk_side_stream(C.c_void_p(own))
    return [own] + ([int(side)] if side else [])


def independent_stream(avoid=None, device=None, tries=24, soft_avoid=()):
    """A torch stream that shares neither a hardware queue nor a command-processor pipe with any of `avoid` (torch streams / raw handles;
    default: engine_streams()): launches on them take the same time whether or not the candidate is active, and vice versa.
    `soft_avoid`: streams it should also stay clear of when a candidate allows it (second choice: clear of `avoid` only)."""
    avoid = engine_streams() if avoid is None else [_raw(x) for x in avoid]
    soft = [_raw(x) for x in soft_avoid]
    if os.environ.get("VK_STREAM_PROBE", "1") == "0":
        return torch.cuda.Stream(device=device)

    def disturbs(a, s):
        for _ in range(2):                                  # a candidate must look clean twice: one noisy idle baseline must not let it through
            busy, idle = active_cost(a, s, n=200, hold_us=2000)
            if busy > 1.2 * idle + 40.0:
                return True
            busy, idle = active_cost(s, a, n=200, hold_us=2000)
            if busy > 1.2 * idle + 40.0:
                return True
        return False

    seen = set()
    first = second = None
    for _ in range(tries):
        s = torch.cuda.Stream(device=device)
        first = first or s
        if s.cuda_stream in seen or s.cuda_stream in avoid:
            continue
        seen.add(s.cuda_stream)
        if any(disturbs(a, s) for a in avoid):
            continue
        if not any(disturbs(a, s) for a in soft):
            return s
        second = second or s
    if second is not None:
        return second
    warnings.warn("volta_amd: no stream clear of the engine's hardware queues / pipes among %d candidates (GPU_MAX_HW_QUEUES=%s): helper "
                  "launches may slow the compute stream's dispatch" % (len(seen), os.environ.get("GPU_MAX_HW_QUEUES", "default")))
    return first
