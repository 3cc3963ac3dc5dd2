"""Optimizer-side drop-ins for the reference's pre-training driver (train_concap.py:21,227-234,307-311):

  AdamW                  pytorch_transformers.optimization.AdamW  (same constructor / param-group semantics)
  WarmupLinearSchedule   pytorch_transformers.optimization.WarmupLinearSchedule
  clip_grad_norm_        torch.nn.utils.clip_grad_norm_

All parameters of a volta_amd model are views of one flat arena, so `step()` is ONE fused kernel over
(p, g, m, v) that also refreshes the bf16 weight copies used by the GEMMs, and the gradient norm / clip
coefficient never leave the device.  The arithmetic (decay after the Adam update with the un-corrected lr,
eps outside the bias correction) follows pytorch-transformers 1.1.0 and is pinned by the oracle tests."""
import ctypes as C
import math
import os

import torch
import weakref

from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR

from . import _lib as L


def _arena_of(params):
    owners = {id(getattr(p, "_vk_owner", None)): getattr(p, "_vk_owner", None) for p in params}
    if len(owners) != 1 or None in owners.values():
        raise RuntimeError("volta_amd optimizers need parameters owned by one volta_amd model on the GPU: call "
                           "model.cuda() and model.materialize() (or run one forward) before building the optimizer")
    model = next(iter(owners.values()))
    arena = model.materialize()
    return model, arena


def _chunks_of(arena, name):
    numel = 1
    for d in arena.shape[name]:
        numel *= d
    return arena.offset[name] // 1024, (arena.offset[name] + numel + 1023) // 1024


def _check_shared_chunks(arena, included, what, among=None):
    """Skipping works on whole 1024-element chunks; only the fused Q|K|V slot packs several tensors into one chunk, and
    those are frozen / trained together in every reference driver."""
    inc, exc = set(), set()
    for n in arena.params:
        if among is not None and n not in among:
            continue
        c0, c1 = _chunks_of(arena, n)
        (inc if n in included else exc).update(range(c0, c1))
    both = inc & exc
    if both:
        raise RuntimeError("volta_amd: parameters sharing an arena chunk (the fused query / key / value slot) must all be %s or none "
                           "of them (chunk %d)" % (what, min(both)))


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True, overlap_with_forward=False, overlap_ranges=8):
        """overlap_with_forward: issue the update as `overlap_ranges` launches over consecutive arena ranges (= forward order) on a stream
        of its own; the model's next forward waits range by range for the weights it is about to read, so the HBM-bound update of the late
        layers runs under the MFMA-bound forward of the early ones.  Same arithmetic, same results.  Opt-in because host code that reads
        parameters right after step() WITHOUT a device synchronisation (`p.cpu()` on the current stream) would race with that stream:
        call `optimizer.synchronize()` (or torch.cuda.synchronize()) before such reads."""
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        self._fused = None
        self._overlap = (bool(overlap_with_forward), int(overlap_ranges))
        # known to the model's arena from construction on (not only from the first step): DistributedDataParallel(mode="zero1") refuses to
        # start a backward whose sharded gradients no volta_amd optimizer would pick up, and the reference's order is model -> DDP -> optimizer
        # -> first backward (train_concap.py:227-253).  Parameters that are not (yet) a materialised volta_amd model's stay lazy (_setup).
        try:
            _, arena = _arena_of([p for g in self.param_groups for p in g["params"]])
            arena._vk_adamw = weakref.ref(self)
        except RuntimeError:
            pass

    def _setup(self):
        allp = [p for g in self.param_groups for p in g["params"]]
        model, arena = _arena_of(allp)
        byptr = {p.data_ptr(): n for n, p in arena.params.items()}
        # classes: groups with identical (initial lr, wd, betas, eps, correct_bias) share one class; chunks of parameters the
        # optimizer was not given (frozen: train_concap.py:200,213 builds its groups from requires_grad parameters) are skipped
        classes, cls_of_chunk = [], torch.full((arena.total // 1024,), L.CHUNK_SKIP, dtype=torch.uint8)
        hyper = None
        spans = []
        for g in self.param_groups:
            h = (tuple(g["betas"]), g["eps"], g["correct_bias"])
            hyper = hyper or h
            if h != hyper:
                raise RuntimeError("volta_amd.AdamW: betas / eps / correct_bias must be common to all groups")
            key = (g.get("initial_lr", g["lr"]), g["weight_decay"])
            if key not in [c[0] for c in classes]:
                if len(classes) == 8:
                    raise RuntimeError("volta_amd.AdamW: more than 8 distinct (lr, weight_decay) classes")
                classes.append((key, g))
            ci = [c[0] for c in classes].index(key)
            for p in g["params"]:
                n = byptr[p.data_ptr()]
                c0, c1 = _chunks_of(arena, n)
                cls_of_chunk[c0:c1] = ci
                spans.append((p, n, c0, c1))
        _check_shared_chunks(arena, {n for _, n, _, _ in spans}, "given to the optimizer")
        arena._vk_adamw = weakref.ref(self)       # clip_grad_norm_ hands its coefficient to this optimizer's next step (see there)
        self._fused = dict(model=model, arena=arena, classes=classes, base_class=cls_of_chunk, spans=spans, masks={},
                           chunk_class=cls_of_chunk.to(arena.device),
                           m=torch.zeros_like(arena.master), v=torch.zeros_like(arena.master), step=0)

    def _chunk_class_for_step(self):
        """Parameters without a gradient this step are skipped like pytorch_transformers' AdamW does (`if p.grad is None:
        continue`): no moment update, no decay, and never a stale gradient left in the arena by an earlier step."""
        f = self._fused
        arena = f["arena"]
        missing = []
        for i, (p, n, c0, c1) in enumerate(f["spans"]):
            g = p.grad
            if g is None:
                missing.append(i)
            elif g.data_ptr() != arena.grad.data_ptr() + 4 * arena.offset[n]:
                raise RuntimeError("volta_amd.AdamW: the gradient of %s is not the engine's arena view (a foreign tensor was "
                                   "assigned to .grad); run backward through the model, or zero_grad(set_to_none=True)" % n)
        if not missing:
            return f["chunk_class"]
        key = tuple(missing)
        if key not in f["masks"]:
            if len(f["masks"]) > 16:
                f["masks"].clear()
            _check_shared_chunks(arena, {f["spans"][i][1] for i in range(len(f["spans"])) if i not in set(missing)}, "with a gradient this step",
                                 among={sp[1] for sp in f["spans"]})
            cc = f["base_class"].clone()
            for i in missing:
                cc[f["spans"][i][2]:f["spans"][i][3]] = L.CHUNK_SKIP
            f["masks"][key] = cc.to(arena.device)
        return f["masks"][key]

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        if self._fused is None:
            self._setup()
        f = self._fused
        arena = f["arena"]
        f["step"] += 1
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        a = L.AdamwArgs()
        a.p, a.g, a.m, a.v = arena.master.data_ptr(), arena.grad.data_ptr(), f["m"].data_ptr(), f["v"].data_ptr()
        a.shadow, a.chunk_class = arena.shadow.data_ptr(), self._chunk_class_for_step().data_ptr()
        pend = getattr(arena, "pending_clip", None)
        clip = None
        if pend is not None:
            if pend[1] == self._grad_names():
                clip = pend[0]                     # the coefficient was computed over exactly the gradients this step consumes: folded into the pass
            else:
                flush_clip(arena)                  # another parameter set: applied to its gradients first
        a.clip = clip.data_ptr() if clip is not None else None
        arena.pending_clip = None
        a.n = arena.total
        for i, (key, grp) in enumerate(f["classes"]):
            a.cls_lr_mult[i] = grp["lr"]          # current (scheduled) lr of the class
            a.cls_wd[i] = grp["weight_decay"]
        a.lr, a.beta1, a.beta2, a.eps = 1.0, b1, b2, g0["eps"]
        t = f["step"]
        a.step_mult = math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t) if g0["correct_bias"] else 1.0
        a.grad_scale = grad_scale
        arena.sync_optimizer()         # an earlier pipelined step nobody waited for (two steps without a forward in between)
        red = self._zero1_reducer()
        if red is not None:
            self._step_zero1(a, arena, red)
        elif not self._overlap[0]:
            L.check(L.lib.vk_adamw_step(C.byref(a), L.stream_ptr()))
        else:
            self._step_pipelined(a, arena, clip)
        arena.mark_shadow_fresh()      # the kernel refreshed the bf16 copies itself
        return loss

    def _grad_names(self):
        """Names of this optimizer's parameters that carry a gradient now (the set a deferred clip coefficient must have been computed over)."""
        return frozenset(n for p, n, _, _ in self._fused["spans"] if p.grad is not None)

    def _zero1_reducer(self):
        """The data-parallel wrapper's reducer when it runs in mode "zero1" and the last backward left this rank with shards to step."""
        ddp = getattr(self._fused["model"], "_ddp", None) if self._fused else None
        red = getattr(ddp, "reducer", None)
        if red is None or red.mode != "zero1" or not (red.sharded or red.replicated):
            return None
        return red

    def _step_zero1(self, a, arena, red):
        """Sharded step (volta_amd/parallel.py, mode "zero1"): the fused kernel runs over the element ranges this rank owns -- its 1/world
        shard of every bucket plus the few replicated slots -- then the updated master weights are all-gathered and the bf16 copies re-cast.
        The moments of foreign shards are left alone; should the shard layout change (another batch shape compiles another bucket plan),
        they are gathered under the old layout first."""
        f = self._fused
        layout = tuple(red.sharded)
        prev = f.get("zero1_layout")
        if prev is not None and prev != layout:
            red.gather(f["m"], prev)
            red.gather(f["v"], prev)
        f["zero1_layout"] = layout
        base = (a.p, a.g, a.m, a.v, a.shadow, a.chunk_class)
        for lo, hi in red.owned():
            assert lo % 1024 == 0 and hi % 1024 == 0 and hi > lo, (lo, hi)
            a.p, a.g, a.m, a.v = base[0] + 4 * lo, base[1] + 4 * lo, base[2] + 4 * lo, base[3] + 4 * lo
            a.shadow, a.chunk_class = base[4] + 2 * lo, base[5] + lo // 1024
            a.n = hi - lo
            L.check(L.lib.vk_adamw_step(C.byref(a), L.stream_ptr()))
        f["model"]._ddp.gather_params(arena)

    def _step_pipelined(self, a, arena, clip):
        f = self._fused
        if "stream" not in f:
            n = max(1, min(self._overlap[1], arena.total // 1024))
            nchunks = arena.total // 1024
            from .streams import independent_stream, engine_streams
            ddp = getattr(f["model"], "_ddp", None)
            comm = getattr(getattr(ddp, "reducer", None), "stream", None)
            with torch.cuda.device(arena.device):          # clear of the compute, weight-gradient and communication streams' hardware queues
                f["stream"] = independent_stream(engine_streams(), device=arena.device, soft_avoid=[comm] if comm is not None else [])
            f["bounds"] = [nchunks * (i + 1) // n for i in range(n)]
            f["events"] = [torch.cuda.Event() for _ in range(n)]
            # Switch, off by default: the ranges from index VK_OPT_NARROW_FROM on are stepped by VK_OPT_CUS RESIDENT workgroups, one per compute
            # unit (vk_adamw_step_on), on the CUs the forward's persistent GEMM launches leave alone.  Built because clip + AdamW add 1.25 ms to
            # the step whether pipelined or not (a full-width launch only gets CUs when a GEMM launch gives them up); measured
            # (profiles/r04_experiments.md 10): one CU streams this update at 50 GB/s alone and 37 under the forward, so 24 of them need 7 ms
            # for what the forward waits for after 5 -- 18.4 ms per step with every range but the first on 24 CUs, 16.95-17.05 with only the
            # last two to five ranges, against 16.65-16.7 full-width.
            f["narrow"] = int(os.environ.get("VK_OPT_CUS", "0"))
            f["narrow_from"] = int(os.environ.get("VK_OPT_NARROW_FROM", "4"))
        side, bounds, events = f["stream"], f["bounds"], f["events"]
        side.wait_stream(torch.cuda.current_stream())          # gradients, their norm and the clip coefficient are final
        if clip is not None:
            clip.record_stream(side)
        base = (a.p, a.g, a.m, a.v, a.shadow, a.chunk_class)
        with torch.cuda.stream(side):
            sp = C.c_void_p(side.cuda_stream)
            lo = 0
            for idx, (hi, ev) in enumerate(zip(bounds, events)):
                if hi > lo:
                    off = lo * 1024
                    a.p, a.g, a.m, a.v = base[0] + 4 * off, base[1] + 4 * off, base[2] + 4 * off, base[3] + 4 * off
                    a.shadow, a.chunk_class = base[4] + 2 * off, base[5] + lo
                    a.n = (hi - lo) * 1024
                    if idx < f["narrow_from"] or f["narrow"] <= 0:
                        L.check(L.lib.vk_adamw_step(C.byref(a), sp))
                    else:
                        L.check(L.lib.vk_adamw_step_on(C.byref(a), f["narrow"], sp))
                ev.record(side)
                lo = hi
        arena.opt_pending = (bounds, events)

    def synchronize(self):
        """Current stream waits for a pipelined step (needed only before host code reads parameters without a device sync)."""
        if self._fused is not None:
            self._fused["arena"].sync_optimizer()

    # ---- checkpoint interchange (volta/train_utils.py:295-340 saves optimizer.state_dict() of pytorch_transformers'
    # AdamW: per parameter {"step", "exp_avg", "exp_avg_sq"}, indexed in param_groups order)
    def _spans(self):
        arena = self._fused["arena"]
        byptr = {p.data_ptr(): n for n, p in arena.params.items()}
        out = []
        for g in self.param_groups:
            for p in g["params"]:
                n = byptr[p.data_ptr()]
                numel = 1
                for d in arena.shape[n]:
                    numel *= d
                out.append((arena.offset[n], numel, tuple(arena.shape[n])))
        return out

    def consolidate_state_dict(self):
        """Data-parallel mode "zero1": every rank holds the moments of its own shards only.  COLLECTIVE -- every rank calls it -- gathers
        them so that `state_dict()` (which the reference calls on rank 0 alone, volta/train_utils.py:295-316) has the whole state.
        A no-op without a sharded optimizer, so a driver may call it unconditionally in front of its `if default_gpu:` save."""
        self.synchronize()
        if self._fused is None:
            return
        f = self._fused
        red = self._zero1_reducer()
        if red is not None and f.get("zero1_layout"):
            red.gather(f["m"], f["zero1_layout"])
            red.gather(f["v"], f["zero1_layout"])
        f["zero1_consolidated_at"] = f["step"]

    def state_dict(self):
        """Collective-free (the reference's checkpoint flow enters it on one rank).  Under "zero1" the moments must have been gathered by
        `consolidate_state_dict()` since the last step; otherwise this raises instead of saving other ranks' stale shards."""
        self.synchronize()
        if self._fused is not None and self._fused.get("zero1_layout") and self._fused.get("zero1_consolidated_at") != self._fused["step"]:
            raise RuntimeError("volta_amd.AdamW.state_dict(): the optimizer state is sharded over the data-parallel ranks (mode 'zero1'); "
                               "call optimizer.consolidate_state_dict() on EVERY rank first, then state_dict() on the saving rank")
        sd = super().state_dict()
        if self._fused is not None and self._fused["step"] > 0:
            f = self._fused
            sd["state"] = {i: {"step": f["step"], "exp_avg": f["m"][o:o + n].view(shape).clone(), "exp_avg_sq": f["v"][o:o + n].view(shape).clone()}
                           for i, (o, n, shape) in enumerate(self._spans())}
        return sd

    def load_state_dict(self, state_dict):
        self.synchronize()
        state = state_dict.get("state", {})
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        if not state:
            return
        if self._fused is None:
            self._setup()
        f = self._fused
        spans = self._spans()
        if len(state) != len(spans):
            raise ValueError("volta_amd.AdamW.load_state_dict: the checkpoint holds %d parameter states, the optimizer %d parameters" % (len(state), len(spans)))
        steps = set()
        for i, (o, n, shape) in enumerate(spans):
            st = state[i] if i in state else state[str(i)]
            f["m"][o:o + n].copy_(st["exp_avg"].reshape(-1))
            f["v"][o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(st["step"]))
        if len(steps) != 1:
            raise ValueError("volta_amd.AdamW.load_state_dict: parameters at different step counts %s (one fused launch updates all of them)" % sorted(steps))
        f["step"] = steps.pop()

    def zero_grad(self, set_to_none=True):
        if self._fused is not None:
            self._fused["arena"].pending_clip = None       # a coefficient nobody consumed dies with the gradients it was computed for
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()


class WarmupLinearSchedule(LambdaLR):
    """Linear warm-up from 0 to 1 over `warmup_steps`, then linear decay to 0 at `t_total`."""

    def __init__(self, optimizer, warmup_steps, t_total, last_epoch=-1):
        self.warmup_steps = warmup_steps
        self.t_total = t_total
        super().__init__(optimizer, self.lr_lambda, last_epoch=last_epoch)

    def lr_lambda(self, step):
        if step < self.warmup_steps:
            return float(step) / float(max(1, self.warmup_steps))
        return max(0.0, float(self.t_total - step) / float(max(1.0, self.t_total - self.warmup_steps)))


def flush_clip(arena):
    """Apply a clip coefficient that is still waiting for an optimizer step to the gradients it was computed over (and forget it)."""
    pend = getattr(arena, "pending_clip", None)
    arena.pending_clip = None
    if pend is None:
        return
    out, names = pend
    if len(names) == sum(p.grad is not None for _, p in arena.param_list()) == len(arena.param_list()):
        arena.grad.mul_(out[1])
    else:
        for n in names:
            arena.view(n, "grad").mul_(out[1])


def clip_grad_norm_(parameters, max_norm, norm_type=2.0, defer_to_optimizer=None, pre_scale=1.0):
    """Global L2 norm of the gradients and clipping by max_norm / (norm + 1e-6) when that is < 1 (torch.nn.utils.clip_grad_norm_ as
    train_concap.py:307 calls it).  Returns the norm as a 0-dim DEVICE tensor (no host synchronisation).
    Where the scaling happens: when a `volta_amd.AdamW` has been built on this model's parameters, the coefficient is handed to its
    next `step()`, which folds it into its single pass over the gradients (saves one read + write of the 968 MB gradient arena); it is
    applied at once instead when no such optimizer exists, when the next step consumes another set of gradients, before the next
    backward, and by `flush_clip(model.materialize())` for code that reads `.grad` between the two calls.  `defer_to_optimizer=False`
    forces the eager form, `True` the deferred one."""
    if float(norm_type) != 2.0:
        raise NotImplementedError("only the L2 norm is supported")
    model = None
    whole = False
    owner = getattr(parameters, "vk_model", None)          # modeling.ArenaParameters: `model.parameters()` of a volta_amd model
    if owner is not None and not parameters.started and owner.__dict__.get("_arena") is not None:
        model, arena = owner, owner.materialize()          # the whole arena, no need to walk 600 tensors
        whole = True
    if model is None:
        params = [parameters] if isinstance(parameters, torch.Tensor) else list(parameters)
        model, arena = _arena_of(params)
        given = {p.data_ptr() for p in params}
    flush_clip(arena)                                       # two clips in a row: the first one's scaling is part of what the second measures
    # parameters without a gradient (frozen, or an unused head) are left out, as torch.nn.utils.clip_grad_norm_ does
    plist, gviews = arena.param_list(), arena.grad_views()
    have = []
    for (n, p), g in zip(plist, gviews):
        if not whole and p.data_ptr() not in given:
            continue
        if p.grad is None:
            continue
        if not (p.grad is g or p.grad.data_ptr() == g.data_ptr()):
            raise RuntimeError("clip_grad_norm_: the gradient of %s is not attached to the engine's arena (run backward first)" % n)
        have.append(n)
    if not have:
        raise RuntimeError("clip_grad_norm_: gradients are not attached to the engine's arena (run backward first)")
    mask = None
    if len(have) != len(plist):
        cache = arena.__dict__.setdefault("_clip_masks", {})
        key = tuple(have)
        if key not in cache:
            if len(cache) > 16:
                cache.clear()
            _check_shared_chunks(arena, set(have), "part of the norm")
            cc = torch.full((arena.total // 1024,), L.CHUNK_SKIP, dtype=torch.uint8)
            for n in have:
                c0, c1 = _chunks_of(arena, n)
                cc[c0:c1] = 0
            cache[key] = cc.to(arena.device)
        mask = cache[key]
    # per-slot sums of squares, then one fixed-order sum: a function of the gradient values alone, so a rank that owns only part of the
    # arena (data-parallel mode "zero1") and receives the other ranks' slot sums obtains the same bits as one that holds everything
    nchunks = arena.total // 1024
    npad = nchunks + (nchunks & 1)                     # an even count: the first-level sums follow the slot sums as doubles
    if not hasattr(arena, "norm_chunks"):
        arena.norm_chunks = torch.zeros(npad + 256, device=arena.device)
    sums = arena.norm_chunks[:npad]
    out = torch.empty(2, device=arena.device)
    ddp = getattr(model, "_ddp", None)
    red = getattr(ddp, "reducer", None)
    if red is not None and red.mode == "zero1" and (red.sharded or red.replicated):
        sums.zero_()
        mine = [(lo + red.rank * s, lo + (red.rank + 1) * s) for lo, hi, s in red.sharded] + (list(red.replicated) if red.rank == 0 else [])
        for lo, hi in mine:
            L.check(L.lib.vk_grad_sqnorm_chunks(L.ptr(arena.grad), lo // 1024, (hi - lo) // 1024, L.ptr(mask), L.ptr(sums), L.stream_ptr()))
        torch.distributed.all_reduce(sums, group=red.pg)        # every slot has one contributor: the sum adds zeros, exactly
    else:
        L.check(L.lib.vk_grad_sqnorm_chunks(L.ptr(arena.grad), 0, nchunks, L.ptr(mask), L.ptr(sums), L.stream_ptr()))
    L.check(L.lib.vk_grad_norm_from_chunks(L.ptr(sums), npad, pre_scale, float(max_norm), L.ptr(out), L.stream_ptr()))
    names = frozenset(have)
    if defer_to_optimizer is None:
        # the reference's plain call: defer when the optimizer that will consume these gradients is ours and steps exactly this set
        opt = getattr(arena, "_vk_adamw", None)
        opt = opt() if opt is not None else None
        defer_to_optimizer = opt is not None and opt._fused is not None and opt._fused["arena"] is arena and opt._grad_names() == names
    arena.pending_clip = (out, names)
    if not defer_to_optimizer:
        flush_clip(arena)
    return out[0]
