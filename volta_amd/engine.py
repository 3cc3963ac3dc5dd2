"""Host side of the MI355X engine: flat parameter arenas and the static execution plan.

Design: for a given (batch, text length, region count) every activation, weight and gradient
buffer is fixed, so a pre-training step is compiled ONCE into two command lists (forward, backward) of
pre-bound kernel launches (`vk_op`, include/volta_hip.h) that the C++ executor replays each step.  No
autograd graph, no per-op Python: `BertForVLPreTraining.forward` costs one C call, `backward` another.

Parameters live in one flat fp32 arena (each nn.Parameter is a view of it), mirrored by a bf16 shadow
arena (MFMA operands), a fp32 gradient arena (wgrad GEMMs write straight into it; DDP buckets are ranges
of it) and, in the optimizer, two moment arenas.  Q/K/V weights of a sub-layer are adjacent, so the three
projections run as one [3H, H] GEMM.

Dropout sites are numbered in the order of the reference's forward (embeddings; per sub-layer: tt, tv,
vv, vt probabilities, text output, vision output; pooled) -- the contract the mask-replay tests rely on.
"""
import ctypes as C
import math
import os

import torch

from . import _lib as L
from ._lib import check, ptr
from .modules import sublayer_schedule

CHUNK = 1024
VIS_TARGET_WIDTH = {"0": 1601, "1": 2048, "2": 2048, "3": 1600, "4": 400, "5": 2048, "6": 1601}      # volta/losses.py:129-137
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def _round_up(x, m):
    return (x + m - 1) // m * m


# ======================================================================================== arenas
class ParamArena:
    """Flat master / shadow / gradient storage; re-points every Parameter at its slice."""

    def __init__(self, model, device):
        named = list(model.named_parameters())
        byname = dict(named)
        slots, seen = [], set()
        for name, p in named:
            if name in seen:
                continue
            group = [name]
            for q in ("query", "v_query"):
                tag = ".attention_self.%s." % q
                if tag in name:
                    pre, suf = name.split(tag)
                    kq = q.replace("query", "")
                    group = ["%s.attention_self.%s%s.%s" % (pre, kq, k, suf) for k in ("query", "key", "value")]
            assert all(g in byname for g in group), group
            seen.update(group)
            slots.append(group)
        self.offset, self.shape = {}, {}
        classes = []
        off = 0
        for group in slots:
            start = off
            for g in group:
                self.offset[g] = off
                self.shape[g] = tuple(byname[g].shape)
                off += byname[g].numel()
            off = _round_up(off, CHUNK)
            nd = any(k in group[0] for k in NO_DECAY)
            assert all(any(k in g for k in NO_DECAY) == nd for g in group)
            classes += [1 if nd else 0] * ((off - start) // CHUNK)
        self.total = off
        self.device = device
        self.master = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.shadow = torch.zeros(self.total, dtype=torch.bfloat16, device=device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.chunk_class = torch.tensor(classes, dtype=torch.uint8, device=device)
        self.names = [n for n, _ in named]
        self.params = byname
        with torch.no_grad():
            for name, p in named:
                v = self.view(name)
                v.copy_(p.data.to(device=device, dtype=torch.float32))
                p.data = v
                p.grad = None
        self.shadow_version = -1
        self.weights_epoch = 0          # bumped whenever the master weights changed (torch-side edits, fused AdamW)
        self.fp8_sites = {}             # data_ptr of an fp32 master view -> (view [N, K], q uint8 [N, Kp], scale [N])
        self.fp8_epoch = -1

    def view(self, name, which="master"):
        buf = getattr(self, which)
        o = self.offset[name]
        n = 1
        for d in self.shape[name]:
            n *= d
        return buf[o:o + n].view(self.shape[name])

    def param_list(self):
        if getattr(self, "_plist", None) is None or len(self._plist) != len(self.params):
            self._plist = list(self.params.items())
        return self._plist

    def grad_views(self):
        """One view of the gradient arena per parameter, created once (the hot loop only re-attaches them)."""
        if getattr(self, "_gviews", None) is None or len(self._gviews) != len(self.params):
            self._gviews = [self.view(n, "grad") for n in self.params]
        return self._gviews

    def span(self, names, which, shape):
        """Contiguous view over consecutive tensors of one slot (the fused Q|K|V block)."""
        buf = getattr(self, which)
        o = self.offset[names[0]]
        n = 1
        for d in shape:
            n *= d
        for a, b in zip(names[:-1], names[1:]):
            assert self.offset[b] == self.offset[a] + int(torch.tensor(self.shape[a]).prod())
        return buf[o:o + n].view(shape)

    def intact(self):
        return all(p.data_ptr() == self.master.data_ptr() + 4 * self.offset[n] for n, p in self.params.items())

    def param_version(self):
        """Sum of the parameters' own version counters.  Every Parameter was re-pointed with `p.data = view`, which gives
        it a counter of ITS OWN: `load_state_dict`, `p.copy_()`, `p.add_()` (torch optimizers, EMA under no_grad) bump it
        and leave `master._version` alone, so the base tensor's counter says nothing about them."""
        return sum(p._version for p in self.params.values()) + self.master._version

    def refresh_shadow(self, force=False):
        """bf16 copies of the weights, rebuilt whenever a parameter was written through torch since the last refresh.
        The fused AdamW refreshes the shadow inside its own launch and calls mark_shadow_fresh().  Not seen by any version
        counter: in-place edits through `p.data` (`p.data.mul_()`): call invalidate_shadow() after those."""
        v = self.param_version()
        if force or self.shadow_version != v:
            self.sync_optimizer()
            check(L.lib.vk_cast_f32_bf16(ptr(self.master), ptr(self.shadow), self.total, L.stream_ptr()))
            self.shadow_version = v
            self.weights_epoch += 1

    def mark_shadow_fresh(self):
        self.shadow_version = self.param_version()
        self.weights_epoch += 1

    def fp8_weight(self, w):
        """e4m3 copy (+ per-output-channel scales) of the fp32 master weight view `w` [N, K], registered for refresh_fp8()."""
        key = w.data_ptr()
        if key not in self.fp8_sites:
            N, K = w.shape
            Kp = _round_up(K, 128)
            self.fp8_sites[key] = (w, torch.zeros(N, Kp, dtype=torch.uint8, device=self.device), torch.ones(N, dtype=torch.float32, device=self.device))
            self.fp8_epoch = -1
        return self.fp8_sites[key][1:]

    def refresh_fp8(self):
        """Re-quantise the registered weights from the fp32 masters when they changed (one row kernel per weight matrix)."""
        if self.fp8_epoch == self.weights_epoch:
            return
        for w, q, sc in self.fp8_sites.values():
            N, K = w.shape
            check(L.lib.vk_quant_rows_fp8(ptr(w), 1, w.stride(0), ptr(q), q.stride(0), ptr(sc), N, K, None, L.stream_ptr()))
        self.fp8_epoch = self.weights_epoch

    def invalidate_shadow(self):
        self.shadow_version = -1

    def sync_optimizer(self):
        """Make the current stream wait for a pipelined optimizer step still in flight on its own stream."""
        pend = getattr(self, "opt_pending", None)
        if pend:
            cur = torch.cuda.current_stream()
            for ev in pend[1]:
                cur.wait_event(ev)
            self.opt_pending = None


# ======================================================================================== plan
class Plan:
    def __init__(self):
        self.ops, self.keep = [], []
        self.c_ops = None
        self.timing = None

    def add(self, kind, a, i0=0, i1=0, i2=0, b=None, c=None):
        self.ops.append((kind, i0, i1, i2, a, b, c))

    def freeze(self):
        arr = (L.Op * max(1, len(self.ops)))()
        for i, (kind, i0, i1, i2, a, b, c) in enumerate(self.ops):
            arr[i] = L.Op(kind, i0, i1, i2, _addr(a), _addr(b), _addr(c))
        self.c_ops = arr
        return self

    def run(self, start=0, end=None):
        end = len(self.ops) if end is None else end
        if end > start:
            base = C.cast(C.byref(self.c_ops, start * C.sizeof(L.Op)), C.POINTER(L.Op))
            if self.timing is not None:
                ms = C.cast(C.byref(self.timing, start * C.sizeof(C.c_float)), C.POINTER(C.c_float))
                check(L.lib.vk_run_ops_timed(base, end - start, L.stream_ptr(), ms))
            else:
                check(L.lib.vk_run_ops(base, end - start, L.stream_ptr()))

    def join_side(self, owner=None):
        """Make the current stream wait for the side-stream work (weight gradients) issued so far by lists run on stream `owner`
        (a raw stream handle; default: the current stream itself)."""
        if owner is None:
            check(L.lib.vk_side_join(L.stream_ptr()))
        else:
            check(L.lib.vk_side_join_from(C.c_void_p(owner), L.stream_ptr()))

    def enable_timing(self, on=True):
        """Per-op HIP-event timing (synchronises the stream on every run; profiling passes only)."""
        self.timing = (C.c_float * max(1, len(self.ops)))() if on else None


def _addr(o):
    if o is None:
        return None
    if isinstance(o, int):
        if 0 < o < (1 << 16):        # a count or an index handed over where a buffer was meant: never a device address
            raise ValueError("engine: %d is not a device address" % o)
        return o
    if isinstance(o, torch.Tensor):
        return o.data_ptr()
    return C.addressof(o)


class Stream:
    """Per-modality geometry: 0 = text, 1 = vision."""

    def __init__(self, L_, B, H):
        self.L, self.M, self.H = L_, B * L_, H


class StepEngine:
    """Buffers + forward / backward command lists of one (B, T, Rv, train) shape."""

    H8_MUL = 8.0           # static scale of the fp8 copy of the GELU output: |h| <= 56 representable, 2^-9 absolute resolution near 0

    def __init__(self, cfg, arena, B, T, Rv, train, heads="pretrain", fp8=False, task=None, task_dropout=0.1, attn_maps=False):
        """fp8: the forward Q|K|V, FFN-up and FFN-down projections of every sub-layer run on the e4m3 MFMA path (csrc/fp8.hip); inputs are
        quantised per row right before the GEMM, weights per output channel whenever they change; the backward stays bf16.
        heads: "pretrain" = the three pre-training heads and losses (BertForVLPreTraining); "tasks" = poolers only, the
        sequence and pooled outputs leave the engine and their gradients enter it (BertForVLTasks)."""
        self.cfg, self.arena, self.B, self.T, self.Rv, self.train = cfg, arena, B, T, Rv, train
        self.heads = heads
        self.task = task              # heads == "tasks": (task id, its task_cfg entry) -- the classifier built behind the poolers
        self.attn_maps = bool(attn_maps)      # keep every attention sub-layer's probabilities (config.visualization, encoders.py:342-358): generic attention kernels
        self.attn_map_info = []
        self.task_dropout = float(task_dropout)      # BertForVLTasks(dropout_prob=...): nn.Dropout on the fused pooled vector / region states (encoders.py:1118-1122)
        self.fp8 = bool(fp8)
        # soft boundaries between dependent GEMMs (VK_GEMM_SOFT_START + row-block counters).  OFF by default: measured neutral in the step
        # (profiles/r04_experiments.md: on gfx950 a barrier-less dispatch starts on an XCD only when that XCD's workgroups of the launch in
        # front are done, so there is no tail overlap to win, and the write-through hand-off costs what the shorter boundary saves)
        self.soft = os.environ.get("VK_SOFT", "0") == "1"
        # FFN-up -> FFN-down as ONE persistent launch with row-block hand-off (vk_gemm_chain): "0" (default) two launches; "fwd" the forward
        # pair; "all" also the backward pair (FFN-down dgrad x gelu' -> FFN-up dgrad).  Measured (profiles/r04_experiments.md 2 and 9,
        # r04_chain_stamps.txt): the launch boundary goes, but the workgroups with three producer tiles set the pace -- -0.08 ms per step for
        # "fwd" while the consumer's poll was a bare counter read; with the acquire the memory model asks for behind the poll (buffer_inv:
        # every waiting tile drops its XCD's cached operand lines) "fwd" costs +0.15 ms, "all" more.  Off: the default step has no
        # intra-launch hand-off in it.
        self.chain = os.environ.get("VK_CHAIN", "0")
        self.side_delay_us = int(os.environ.get("VK_SIDE_DELAY_US", "0"))
        if os.environ.get("VK_RESERVE_CUS"):                  # study knob: the persistent GEMM launches leave this many CUs unclaimed (process-wide)
            L.lib.vk_gemm_reserve_cus(int(os.environ["VK_RESERVE_CUS"]))
        # How a sub-layer's weight-gradient block (side stream) is started: "event" (default) -- the fork event of rounds 1-3; "gate" -- a
        # one-wave gate at the head of the block that the sub-layer's last dgrad releases as its first workgroup retires, no stream event
        # (vk_gemm_problem::retire_flag + vk_gate_wait).  Built to take the cross-queue wake-up latency out of the schedule; measured
        # 0.1-0.2 ms per step slower than the event form and no cure for the slow steps it was built against (those came from a helper
        # stream on the compute stream's pipe: volta_amd/streams.py, profiles/r04_experiments.md).  Kept as a switch.
        self.side_gate = os.environ.get("VK_SIDE_START", "event") == "gate"
        if self.side_gate:
            # a gate on the hardware queue of the stream that releases it would wait for its own releaser (until its timeout): only with the side
            # stream on a queue of its own
            from . import streams as S
            own, side = S.engine_streams()
            if S.shares_queue(own, side) or S.shares_queue(side, own):
                self.side_gate = False
        dev = arena.device
        self.dev = dev
        H, Hv = cfg.hidden_size, cfg.v_hidden_size
        # Widths: the two streams may differ (config/vilbert_base.json: 768 text, 1024 vision) and an attention sub-layer may project to a
        # width of its own (sublayer2attn_hidden_size); every width is a multiple of 64, LayerNorm widths at most 1024.  Head sizes: 64 on
        # the MFMA attention kernels, 32 / 96 / 128 on the generic ones.
        if H % 64 or Hv % 64 or H > 1024 or Hv > 1024:
            raise NotImplementedError("hidden sizes must be multiples of 64, at most 1024 (got %d / %d)" % (H, Hv))
        self.wide = H != Hv or bool(cfg.sublayer2attn_hidden_size or cfg.sublayer2num_attention_heads or cfg.sublayer2intermediate_size or
                                    cfg.sublayer2v_attn_hidden_size or cfg.sublayer2v_num_attention_heads or cfg.sublayer2v_intermediate_size) or \
            H // cfg.num_attention_heads != 64 or Hv // cfg.v_num_attention_heads != 64
        if self.wide and fp8:
            raise NotImplementedError("the fp8 projection path covers the single-width (ctrl_*) geometry")
        if self.wide and cfg.image_embeddings != "vilbert":
            raise NotImplementedError("different stream widths are built for the ViLBERT embeddings (config/vilbert_base.json)")
        if cfg.hidden_act != "gelu" or cfg.v_hidden_act != "gelu":
            raise NotImplementedError("engine supports gelu activations (every reference config)")
        if cfg.fusion_method not in ("mul", "sum", "text", "vl-bert_vqa", "none"):
            raise ValueError("Invalid fusion method: %s" % cfg.fusion_method)
        self.unused_params = set()    # parameters no launch of this plan reads: their .grad stays None, as under the reference's autograd
        # rows beyond the MFMA attention tiles (64 text tokens, 128 regions: VCR's 80-token captions, 200 / 256 / 306 regions of the
        # grounding tasks, config_tasks/all_tasks.yml) run on the generic attention kernels: up to 512 keys per query row
        if T + Rv > 512:
            raise NotImplementedError("more than 512 keys per query row (%d text + %d vision rows)" % (T, Rv))
        self.H, self.I, self.nh = H, cfg.intermediate_size, cfg.num_attention_heads
        self.st = [Stream(T, B, H), Stream(Rv, B, Hv)]
        self.R = Rv - (1 if cfg.add_global_imgfeat is not None else 0)
        self.bufs = {}
        self.keep = []
        self.fwd, self.bwd = Plan(), Plan()
        self.bwd_groups = []          # per forward stage: list of op tuples, replayed in reverse stage order
        self.site = 0
        self.seed = torch.zeros(1, dtype=torch.int64, device=dev)
        self.inputs = {}              # name -> list of (struct, field) patched every step
        self.taps = {}
        self._build()

    # ---------------------------------------------------------------- helpers
    def buf(self, name, shape, dtype=torch.bfloat16, zero=False):
        assert name not in self.bufs, name
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.dev)
        self.bufs[name] = t
        return t

    def tmp(self, name, shape, dtype=torch.bfloat16):
        """Backward temporaries are shared by all sub-layers (launches on one stream are ordered).  Ops that are being built for a
        side-stream block run NEXT to main-stream launches: they get temporaries of their own (`_aside`)."""
        name += getattr(self, "_aside", "")
        if name not in self.bufs:
            self.bufs[name] = torch.empty(shape, dtype=dtype, device=self.dev)
        t = self.bufs[name]
        assert tuple(t.shape) == tuple(shape) and t.dtype == dtype, (name, t.shape, shape)
        return t

    def drop(self, p):
        site = self.site
        self.site += 1
        return L.dropout_cfg(self.seed.data_ptr(), site, p if self.train else 0.0)

    def k(self, obj):
        self.keep.append(obj)
        return obj

    def patch(self, name, struct, field, index=None):
        self.inputs.setdefault(name, []).append((struct, field, index))

    def W(self, name):
        return self.arena.view(name, "shadow")

    def Pm(self, name):
        return self.arena.view(name, "master")

    def G(self, name):
        return self.arena.view(name, "grad")

    def gemm(self, plan_ops, layout, epi, probs, geometry=0):
        """geometry: tile code of vk_gemm_grouped_ex carried in the op's i0 above the layout (0 = the library's heuristic)"""
        arr = self.k((L.GemmProblem * len(probs))(*probs))
        plan_ops.append((L.OP_GEMM, layout | (geometry << 8), epi, len(probs), arr, None, None))

    def gemm_chain(self, plan_ops, layout, epi_p, producers, epi_c, consumers):
        ap = self.k((L.GemmProblem * len(producers))(*producers))
        ac = self.k((L.GemmProblem * len(consumers))(*consumers))
        plan_ops.append((L.OP_GEMM_CHAIN, layout, epi_p | (epi_c << 8), len(producers) | (len(consumers) << 8), ap, ac, None))

    def prob(self, A, B, Cout, M, N, K, lda, ldb, ldc, bias=None, R=None, ldr=0, C2=None, bias_grad=None, dyn=None, n_store=0):
        return L.GemmProblem(_addr(A), _addr(B), _addr(Cout), _addr(C2), _addr(bias), _addr(R), _addr(bias_grad), _addr(dyn),
                             M, N, K, lda, ldb, ldc, ldr, n_store)

    # ---- split accumulations (include/volta_hip.h, vk_gemm_problem::ws): K-slices of one product in one launch, summed by the last arriver
    # bytes of partial-tile workspace per stream (launches on one stream are ordered: they share it).  Sized by the first request -- the only user
    # is the image projection's weight gradient, one product per launch: nparts x tiles x 320 KiB = 84 MB for [1024 x 2048] at B = 256 --
    # instead of a fixed 768 MiB per cached engine; VK_SPLIT_WS_MB overrides
    SPLIT_WS_MIN = 32 << 20

    def _split_alloc(self, tag, layout, M, N, nparts, geometry):
        """(ws address, cnt address) for one split accumulation of the launch being built on stream `tag` ("main" / "side")."""
        tiles = C.c_int(0)
        nbytes = L.lib.vk_gemm_split_workspace_bytes(layout, M, N, nparts, geometry, C.byref(tiles))
        assert nbytes > 0 and tiles.value > 0, (layout, M, N, nparts, geometry)
        aside, self._aside = getattr(self, "_aside", ""), ""          # the arenas belong to a stream, not to a block of ops
        if "split_ws_" + tag not in self.bufs:
            mb = os.environ.get("VK_SPLIT_WS_MB")
            cap = (int(mb) << 20) if mb else max(self.SPLIT_WS_MIN, _round_up(int(nbytes * 1.25), 1 << 20))
            self.bufs["split_ws_" + tag] = torch.empty(cap, dtype=torch.uint8, device=self.dev)
        ws = self.bufs["split_ws_" + tag]
        self._aside = aside
        if "split_cnt_" + tag not in self.bufs:
            self.bufs["split_cnt_" + tag] = torch.zeros(1 << 16, dtype=torch.int32, device=self.dev)      # zero once: every launch leaves them zero
        cnt = self.bufs["split_cnt_" + tag]
        cur = self.__dict__.setdefault("_split_cur", {}).setdefault(tag, [0, 0])
        if cur[0] + nbytes > ws.numel() or cur[1] + tiles.value > cnt.numel():
            raise RuntimeError("split-accumulation workspace too small (%d + %d of %d bytes): set VK_SPLIT_WS_MB" % (cur[0], nbytes, ws.numel()))
        out = (ws.data_ptr() + cur[0], cnt.data_ptr() + 4 * cur[1])
        cur[0] += _round_up(nbytes, 256)
        cur[1] += tiles.value
        return out

    # ---- soft boundaries (include/volta_hip.h, VK_GEMM_SOFT_START): a GEMM whose A operand is the output of the GEMM launched right before it
    # is enqueued without the stream-order barrier; its tiles wait for row-block counters that the producer's tiles raise
    SOFT_ROWS = 1 << 16

    def soft_counters(self, nrb, backward=False):
        """nrb int32 counters (one per 256-row block) out of the engine's counter arena, which one fill launch at the head of each command
        list zeroes; plus the address of the error word a timed-out poll raises."""
        cnt = self.bufs["soft_cnt"]
        n = _round_up(nrb, 32)                               # a producer's counters on lines of their own
        assert self._soft_cur + n <= self._soft_cur_b, "hand-off counter arena too small"
        if backward:
            self._soft_cur_b -= n
            cur = self._soft_cur_b
            self._soft_zero_b.p[0] = cnt.data_ptr() + 4 * cur
            self._soft_zero_b.n[0] = 4 * (self.SOFT_ROWS - cur)
            if not getattr(self, "_soft_zero_b_listed", False):
                self.bwd_pro.append((L.OP_GENERIC, 0, 0, 0, self._soft_zero_b, None, None))
                self._soft_zero_b_listed = True
        else:
            cur = self._soft_cur
            self._soft_cur = cur + n
            self._soft_zero.n[0] = 4 * (self._soft_cur - 32)     # the fill launch at the head of the forward list covers every counter handed out
        return cnt.data_ptr() + 4 * cur, cnt.data_ptr()

    def soft_error(self):
        """Non-zero when a guarded tile (1) or a weight-gradient gate (2) gave up waiting in some step since the engine was built (host
        synchronisation: tests and diagnostics)."""
        e = int(self.bufs["soft_cnt"][0].item()) if "soft_cnt" in self.bufs else 0
        return e | (int(self.bufs["gate_err"][0].item()) if "gate_err" in self.bufs else 0)

    def _split_launch_done(self, tag):
        self.__dict__.setdefault("_split_cur", {})[tag] = [0, 0]

    @staticmethod
    def split_geometry(widths):
        """One tile geometry for a launch whose outputs are `widths` columns wide: 256 x 192 tiles when they cover every width without a
        ragged column tile and 256-wide ones would not (N = 768), 256 x 256 otherwise."""
        return 259 if all(n % 192 == 0 for n in widths) and any(n % 256 for n in widths) else 258

    def prob_parts(self, tag, geometry, layout, slices, Cout, M, N, ldc, **kw):
        """The problems of ONE product cut into len(slices) parts; slices: [(A address, B address, K, lda, ldb)]."""
        if len(slices) == 1:
            A, B, K, lda, ldb = slices[0]
            return [self.prob(A, B, Cout, M, N, K, lda, ldb, ldc, **kw)]
        ws, cnt = self._split_alloc(tag, layout, M, N, len(slices), geometry)
        out = []
        for i, (A, B, K, lda, ldb) in enumerate(slices):
            q = self.prob(A, B, Cout, M, N, K, lda, ldb, ldc, **kw)
            q.ws, q.cnt, q.part, q.nparts = ws, cnt, i, len(slices)
            out.append(q)
        return out

    def gemm_fp8(self, plan_ops, epi, specs):
        """specs: [(A bf16 [M, K] or (A8, scale_a) already quantised, master weight view [N, K], C, bias, C2)] -- one fp8 launch for all of
        them, preceded by the row quantisation of every bf16 A."""
        probs = []
        for idx, spec in enumerate(specs):
            A, Wm, Cout, bias, C2 = spec[:5]
            c8 = spec[5] if len(spec) > 5 else None            # (h8 uint8 [M, N], multiplier): e4m3 copy of the GELU output
            w8, ws = self.arena.fp8_weight(Wm)
            N, K = Wm.shape
            if isinstance(A, tuple):
                a8, sa = A
                M = a8.shape[0]
            else:
                M = A.shape[0]
                a8 = self.tmp("fp8_a%d_%d_%d" % (idx, M, K), (M, _round_up(K, 128)), torch.uint8)      # consumed by the launch that follows: shared by all sub-layers
                if K % 128:
                    a8[:, K:].zero_()               # the K padding is multiplied into the product: it must be zero, not stale bytes
                sa = self.tmp("fp8_sa%d_%d_%d" % (idx, M, K), (M,), torch.float32)
                plan_ops.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_QUANT_ROWS, p=(A, a8, sa, None), n=(M, K, A.stride(0), a8.stride(0), 0)), None, None))
            probs.append(L.GemmFp8Problem(L.GemmProblem(_addr(a8), _addr(w8), _addr(Cout), _addr(C2), _addr(bias), None, None, None,
                                                        M, N, K, a8.stride(0), w8.stride(0), Cout.stride(0), 0, 0), _addr(sa), _addr(ws),
                                          _addr(c8[0]) if c8 else None, c8[1] if c8 else 0.0, c8[0].stride(0) if c8 else 0))
        arr = self.k((L.GemmFp8Problem * len(probs))(*probs))
        plan_ops.append((L.OP_GEMM_FP8, 0, epi, len(probs), arr, None, None))

    def generic(self, fn, p=(), n=(), f=(), drop=None):
        g = L.GenericArgs()
        g.fn = fn
        for i, x in enumerate(p):
            g.p[i] = _addr(x)
        for i, x in enumerate(n):
            g.n[i] = int(x)
        for i, x in enumerate(f):
            g.f[i] = float(x)
        g.drop = drop or L.dropout_cfg(None, 0, 0.0)
        return self.k(g)

    def ln_args(self, d, x, gname, bname, y, z, mean, rstd, M, drop, post=0, out_scale=1.0, addvec=None, dyn=None, segs=None, fp8_out=None, H=None):
        """fp8_out = (q uint8 [M, ld8], scale fp32 [M]): the kernel also leaves a row-quantised e4m3 copy of y (the next projection's A operand)."""
        a = L.LnArgs(_addr(d), _addr(x), _addr(addvec), _addr(self.Pm(gname)), _addr(self.Pm(bname)), _addr(y), _addr(z), _addr(mean),
                     _addr(rstd), _addr(dyn), M, H or self.H, M, post, out_scale, drop, _mk_segs(drop, segs))
        if fp8_out is not None:
            a.y8, a.y8_scale, a.ld8 = _addr(fp8_out[0]), _addr(fp8_out[1]), fp8_out[0].stride(0)
        return self.k(a)

    def fp8_hidden(self, m):
        """Per-modality e4m3 copy of the current hidden state, written by the LayerNorm that produces it and read by the next
        sub-layer's first projection (one buffer per modality: consumed before the next LayerNorm of that modality runs)."""
        M, H = self.st[m].M, self.H
        q = self.tmp("fp8_x%d" % m, (M, _round_up(H, 128)), torch.uint8)
        if H % 128:
            q.zero_()
        return q, self.tmp("fp8_xs%d" % m, (M,), torch.float32)

    def ln_bwd_args(self, dy, z, mean, rstd, gname, bname, dz, dd, M, drop, post=0, out_scale=1.0, dyn=None, segs=None, accumulate=0, defer=False, own_partial=False,
                    H=None):
        """`defer`: the dgamma / dbeta column reduction is left to an OP_LN_FINALIZE that the next _wgrad() places in its
        side-stream block; the partial records then need a buffer of their own.  `own_partial`: this launch runs inside a side-stream
        block, next to main-stream LayerNorm backwards -- it cannot share their scratch records either."""
        H = H or self.H
        Hmax = max(self.st[0].H, self.st[1].H)
        if (own_partial or getattr(self, "_aside", "")) and not defer:
            self._n_ln_partial = getattr(self, "_n_ln_partial", 0) + 1
            partial = self.buf("ln_partial_%d" % self._n_ln_partial, (L.lib.vk_ln_bwd_partial_rows(M) * 2 * H,), torch.float32)
        elif defer:
            self._n_ln_partial = getattr(self, "_n_ln_partial", 0) + 1
            partial = self.buf("ln_partial_%d" % self._n_ln_partial, (L.lib.vk_ln_bwd_partial_rows(M) * 2 * H,), torch.float32)
            accumulate |= 2
        else:
            partial = self.tmp("ln_partial", (L.lib.vk_ln_bwd_partial_rows(max(self.st[0].M, self.st[1].M)) * 2 * Hmax,), torch.float32)
        a = L.LnBwdArgs(_addr(dy), _addr(z), _addr(mean), _addr(rstd), _addr(self.Pm(gname)), _addr(dz), _addr(dd), _addr(partial),
                        _addr(self.G(gname)), _addr(self.G(bname)), _addr(dyn), M, H, M, post, out_scale, accumulate, drop,
                        _mk_segs(drop, segs))
        a = self.k(a)
        if defer:
            self._deferred_ln = getattr(self, "_deferred_ln", []) + [a]
        return a

    # ---------------------------------------------------------------- build
    def _build(self):
        cfg, B, H = self.cfg, self.B, self.H
        st = self.st
        f = self.fwd.ops
        if self.soft or self.chain != "0":
            # row-block counters of the hand-offs: word 0 is the error word (never cleared), the counters start on the next line and
            # are zeroed by ONE fill launch per command list, in front of everything (its length grows as _build hands counters out);
            # the forward's counters grow up from the start of the arena, the backward's down from its end
            self.bufs["soft_cnt"] = torch.zeros(self.SOFT_ROWS, dtype=torch.int32, device=self.dev)
            self._soft_cur = 32
            self._soft_cur_b = self.SOFT_ROWS
            self._soft_zero = self.generic(L.FN_MEMSET, p=(self.bufs["soft_cnt"].data_ptr() + 128,), n=(0, 0))
            self._soft_zero_b = self.generic(L.FN_MEMSET, p=(self.bufs["soft_cnt"].data_ptr() + 4 * self.SOFT_ROWS,), n=(0, 0))
            f.append((L.OP_GENERIC, 0, 0, 0, self._soft_zero, None, None))
        # per-step inputs (static staging copies are avoided: the few ops that read them are patched)
        self.masks = [self.buf("mask_t", (B, self.T), torch.float32), self.buf("mask_v", (B, self.Rv), torch.float32)]
        for m, name in ((0, "attention_mask"), (1, "image_attention_mask")):
            g = self.generic(L.FN_MASK_PREP, p=(None, self.masks[m]), n=(B * st[m].L,))
            self.patch(name, g, "p", 0)
            f.append((L.OP_GENERIC, 0, 0, 0, g, None, None))
        self.x = [None, None]        # current hidden state buffers
        self.x8 = [None, None]       # fp8 path: (e4m3 copy, row scales) of x[m] when its producer wrote one
        self.level = [0, 0]          # number of sub-layers that transformed x[m] so far (ping-pong parity of dX)
        self.bwd_pro = []            # zero-fills of gradient tensors that are accumulated with atomics
        kind = cfg.image_embeddings
        bwd_stages = []
        nemb = 1 if kind in ("visualbert", "vl-bert") else 2
        self.stage_prefix = [["bert.embeddings.", "bert.v_embeddings."]] * nemb + [["bert.encoder.layer.%d." % n] for n, _ in sublayer_schedule(cfg)]
        # ViLBERT's first sub-layers are text-only (ctrl_vilbert_base: 0-11): the image embedding does not depend on them, nor they on it.
        # Its forward runs on the executor's side stream next to them (joined before the first sub-layer that touches the vision
        # stream), its backward likewise as soon as that sub-layer's backward has produced the vision gradient -- instead of at the very
        # end of the list, after the text-only sub-layers, whose launches (240 tiles or fewer) leave CUs idle.
        sched = list(sublayer_schedule(cfg))
        uses_v = lambda n, typ: (n in cfg.tv_attn_sublayers or n in cfg.vt_attn_sublayers or n in cfg.vv_attn_sublayers) if typ == "attn" else n in cfg.v_ff_sublayers
        k_vis = next((k for k, (n, typ) in enumerate(sched) if uses_v(n, typ)), len(sched))
        emb_image_aside = kind in ("vilbert", "lxmert") and 0 < k_vis < len(sched)
        emb_image_bwd = None
        if kind in ("vilbert", "lxmert"):
            bwd_stages.append(self._emb_text("bert.embeddings."))
            i0 = len(f)
            self._aside = "_aside" if emb_image_aside else ""           # its backward runs in a side-stream block: own temporaries and LayerNorm scratch
            img_bwd = (self._emb_image_vilbert if kind == "vilbert" else self._emb_image_lxmert)("bert.v_embeddings.")
            self._aside = ""
            if emb_image_aside:
                f.insert(i0, (L.OP_SIDE_BEGIN, 0, 0, 0, None, None, None))
                f.append((L.OP_SIDE_END, 15, 0, 0, None, None, None))
                emb_image_bwd = [(L.OP_SIDE_BEGIN, 0, 0, 0, None, None, None)] + img_bwd + [(L.OP_SIDE_END, 14, 0, 0, None, None, None)]
                bwd_stages.append([])
            else:
                bwd_stages.append(img_bwd)
        elif kind == "uniter":
            bwd_stages.append(self._emb_text("bert.embeddings."))
            bwd_stages.append(self._emb_image_uniter("bert.embeddings."))
        elif kind == "visualbert":
            bwd_stages.append(self._emb_visualbert("bert.embeddings."))
        elif kind == "vl-bert":
            bwd_stages.append(self._emb_vlbert("bert.embeddings."))
        else:
            raise NotImplementedError("image_embeddings=%r" % kind)
        self.taps["emb_t"], self.taps["emb_v"] = self.x[0], self.x[1]
        # Weight gradients run on the executor's side stream (VK_OP_SIDE_*): sub-layer number k (forward order) keeps
        # its backward temporaries in buffer set k % 2 and records side event k % 8 after its wgrad; its backward
        # first waits for the event of sub-layer k + 2, the previous user of that buffer set.
        self.n_sub = len(list(sublayer_schedule(cfg)))
        self.fwd_sub_start = []       # forward op index at which sub-layer k begins (the optimizer overlap cuts the list there)
        self.sublayer_ids = [n for n, _ in sched]       # taps "t<n>" / "v<n>": both streams' states after sub-layer n, forward order
        for k, (n, typ) in enumerate(sched):
            self.sub_k = k
            self.fwd_sub_start.append(len(self.fwd.ops))
            if emb_image_aside and k == k_vis:
                f.append((L.OP_WAIT_SIDE, 15, 0, 0, None, None, None))       # the vision stream enters here: its embedding must be complete
            ops = self._attn_sublayer(n) if typ == "attn" else self._ffn_sublayer(n)
            if k + 2 < self.n_sub:
                ops.insert(0, (L.OP_WAIT_SIDE, (k + 2) % 8, 0, 0, None, None, None))
            if emb_image_aside and k == k_vis:
                ops = ops + emb_image_bwd                                   # d(loss)/d(vision embedding) is final after this sub-layer's backward
            bwd_stages.append(ops)
            self.taps["t%d" % n], self.taps["v%d" % n] = self.x[0], self.x[1]
        self.fwd_heads_start = len(self.fwd.ops)
        head_bwd = self._heads() if self.heads == "pretrain" else self._heads_tasks()
        # backward list: zero-fills, heads, then stages in reverse; bwd_marks[s] = op index at which backward
        # stage s is complete (stage 0 = heads), param_ready_stage[name] = stage after which its gradient is final
        self.bwd.ops = list(self.bwd_pro) + list(head_bwd)
        self.bwd_marks = [len(self.bwd.ops)]
        for ops in reversed(bwd_stages):
            self.bwd.ops += ops
            self.bwd_marks.append(len(self.bwd.ops))
        self.bwd.ops.append((L.OP_JOIN, 0, 0, 0, None, None, None))
        prefixes = [["bert.t_pooler.", "bert.v_pooler.", "cls.", "clfs_dict."]] + [pf for pf in reversed(self.stage_prefix)]
        self.param_ready_stage = {}
        for name in self.arena.params:
            self.param_ready_stage[name] = max(i for i, pf in enumerate(prefixes) if any(name.startswith(q) for q in pf))
        self.fwd.freeze()
        self.bwd.freeze()

    # -- gradient buffers of the hidden states.  dX[m] ping-pongs between two buffers: the k-th sub-layer (in
    # forward order) that transforms x[m] reads d(loss)/d(its output) from buffer k%2 and writes the gradient of
    # its input to buffer (k-1)%2; the embeddings read buffer 0, the heads fill buffer (final k)%2.
    def _dx(self, m, parity):
        aside, self._aside = getattr(self, "_aside", ""), ""          # the hidden-state gradients are shared by definition
        t = self.tmp("dx%d_%d" % (m, parity), (self.st[m].M, self.st[m].H))
        self._aside = aside
        return t

    def _dx_step(self, m):
        self.level[m] += 1
        k = self.level[m]
        return self._dx(m, k % 2), self._dx(m, (k - 1) % 2)

    def _zero_grad(self, t):
        self.bwd_pro.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_MEMSET, p=(t,), n=(t.numel() * t.element_size(), 0)), None, None))

    # ---------------------------------------------------------------- embeddings
    def _emb_text(self, pre):
        cfg, st, H = self.cfg, self.st[0], self.H
        f = self.fwd.ops
        z = self.buf("emb_t_z", (st.M, H))
        y = self.buf("emb_t_y", (st.M, H))
        mean, rstd = self.buf("emb_t_mean", (st.M,), torch.float32), self.buf("emb_t_rstd", (st.M,), torch.float32)
        ea = self.k(L.EmbedArgs(None, None, None, _addr(self.Pm(pre + "word_embeddings.weight")), _addr(self.Pm(pre + "position_embeddings.weight")),
                                _addr(self.Pm(pre + "token_type_embeddings.weight")), None, _addr(z), st.M, st.L, H,
                                cfg.vocab_size, cfg.max_position_embeddings, cfg.type_vocab_size))
        self.patch("input_ids", ea, "ids")
        self.patch("token_type_ids", ea, "type_ids")
        f.append((L.OP_EMBED_FWD, 0, 0, 0, ea, None, None))
        dr = self.drop(cfg.hidden_dropout_prob)
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(z, None, pre + "LayerNorm.weight", pre + "LayerNorm.bias", y, None, mean, rstd, st.M, dr, post=1), None, None))
        self.x[0] = y
        # backward
        b = []
        dz = self.tmp("dz0", (st.M, H))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(self._dx(0, 0), z, mean, rstd, pre + "LayerNorm.weight", pre + "LayerNorm.bias", dz, None, st.M, dr, post=1), None, None))
        gpos, gtyp = self.G(pre + "position_embeddings.weight"), self.G(pre + "token_type_embeddings.weight")
        self._zero_grad(gpos)
        self._zero_grad(gtyp)
        eb = self.k(L.EmbedBwdArgs(_addr(dz), None, None, None, _addr(self.G(pre + "word_embeddings.weight")), _addr(gpos), _addr(gtyp),
                                   st.M, st.L, H, cfg.type_vocab_size, cfg.vocab_size, cfg.max_position_embeddings))
        self.patch("input_ids", eb, "ids")
        self.patch("token_type_ids", eb, "type_ids")
        b.append((L.OP_WAIT_SIDE, 12, 0, 0, None, None, None))      # the tied LM decoder's weight gradient (heads, side stream) initialises the table's gradient
        b.append((L.OP_EMBED_BWD, 0, 0, 0, eb, None, None))
        return b

    def _img_proj(self, pre, wname, tag):
        """feat (fp32) -> bf16 -> [Mv, Hv] = feat W^T + b ; returns (proj, feat_bf16)."""
        cfg, st, H = self.cfg, self.st[1], self.st[1].H
        f = self.fwd.ops
        F_ = cfg.v_feature_size
        if F_ % 64:
            raise NotImplementedError("v_feature_size must be a multiple of 64")
        featb = self.buf(tag + "_feat_bf16", (st.M, F_))
        g = self.generic(L.FN_CAST, p=(None, featb), n=(st.M * F_,))
        self.patch("image_feat", g, "p", 0)
        f.append((L.OP_GENERIC, 0, 0, 0, g, None, None))
        proj = self.buf(tag + "_proj", (st.M, H))
        self.gemm(f, L.NT, L.EPI_BF16, [self.prob(featb, self.W(pre + wname + ".weight"), proj, st.M, H, F_, F_, F_, H, bias=self.Pm(pre + wname + ".bias"))])
        return proj, featb

    def _loc_proj(self, pre, tag):
        cfg, st, H = self.cfg, self.st[1], self.st[1].H
        out = self.buf(tag + "_locproj", (st.M, H))
        g = self.generic(L.FN_LOC_FWD, p=(None, self.Pm(pre + "image_location_embeddings.weight"), self.Pm(pre + "image_location_embeddings.bias"), out),
                         n=(st.M, H, cfg.num_locs))
        self.patch("image_loc", g, "p", 0)
        self.fwd.ops.append((L.OP_GENERIC, 0, 0, 0, g, None, None))
        return out

    def _img_proj_bwd(self, b, pre, wname, dz, featb):
        st, H, F_ = self.st[1], self.st[1].H, self.cfg.v_feature_size
        # [H x F] output (24 tiles of 256 x 256) over B * Rv rows: row chunks as parts of one split accumulation fill the chip (146 -> 87 us, profiles/r03_ops_per_launch.txt)
        nparts = max(1, min(8, st.M // 1024))
        step = _round_up(-(-st.M // nparts), 64)
        tag = "side" if getattr(self, "_aside", "") else "main"
        geo = self.split_geometry([F_])
        slices = [(_addr(dz[r0:]), _addr(featb[r0:]), min(step, st.M - r0), H, F_) for r0 in range(0, st.M, step)]
        probs = self.prob_parts(tag, geo, L.TN, slices, self.G(pre + wname + ".weight"), H, F_, F_, bias_grad=self.G(pre + wname + ".bias"))
        self.gemm(b, L.TN, L.EPI_F32, probs, geometry=geo if len(probs) > 1 else 0)
        self._split_launch_done(tag)

    def _loc_proj_bwd(self, b, pre, dz):
        st, H = self.st[1], self.st[1].H
        part = self.tmp("loc_partial", (L.lib.vk_rows32(st.M) * 9 * H,), torch.float32)
        g = self.generic(L.FN_LOC_BWD, p=(dz, None, part, self.G(pre + "image_location_embeddings.weight"), self.G(pre + "image_location_embeddings.bias")),
                         n=(st.M, H, self.cfg.num_locs))
        self.patch("image_loc", g, "p", 1)
        b.append((L.OP_GENERIC, 0, 0, 0, g, None, None))

    def _emb_image_vilbert(self, pre):
        cfg, st, H = self.cfg, self.st[1], self.st[1].H
        proj, featb = self._img_proj(pre, "image_embeddings", "emb_v")
        loc = self._loc_proj(pre, "emb_v")
        y = self.buf("emb_v_y", (st.M, H))
        mean, rstd = self.buf("emb_v_mean", (st.M,), torch.float32), self.buf("emb_v_rstd", (st.M,), torch.float32)
        dr = self.drop(cfg.v_hidden_dropout_prob)
        self.fwd.ops.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(proj, loc, pre + "LayerNorm.weight", pre + "LayerNorm.bias", y, proj, mean, rstd, st.M, dr, post=1, H=H), None, None))
        self.x[1] = y
        b = []
        dz = self.tmp("dz1", (st.M, H))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(self._dx(1, 0), proj, mean, rstd, pre + "LayerNorm.weight", pre + "LayerNorm.bias", dz, None, st.M, dr, post=1, H=H), None, None))
        self._img_proj_bwd(b, pre, "image_embeddings", dz, featb)
        self._loc_proj_bwd(b, pre, dz)
        return b

    def _emb_image_lxmert(self, pre):
        cfg, st, H = self.cfg, self.st[1], self.H
        f = self.fwd.ops
        proj, featb = self._img_proj(pre, "image_embeddings", "emb_v")
        loc = self._loc_proj(pre, "emb_v")
        nodrop = L.dropout_cfg(None, 0, 0.0)
        a_n, b_n = self.buf("emb_v_imgn", (st.M, H)), self.buf("emb_v_locn", (st.M, H))
        st_a = [self.buf("emb_v_%s" % s, (st.M,), torch.float32) for s in ("mean_a", "rstd_a", "mean_b", "rstd_b")]
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(proj, None, pre + "ImgLayerNorm.weight", pre + "ImgLayerNorm.bias", a_n, None, st_a[0], st_a[1], st.M, nodrop), None, None))
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(loc, None, pre + "LocLayerNorm.weight", pre + "LocLayerNorm.bias", b_n, None, st_a[2], st_a[3], st.M, nodrop), None, None))
        y = self.buf("emb_v_y", (st.M, H))
        dr = self.drop(cfg.v_hidden_dropout_prob)
        f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_ADD_DROPOUT, p=(a_n, b_n, y), n=(st.M, H, 0), f=(0.5,), drop=dr), None, None))
        self.x[1] = y
        b = []
        g = self.tmp("dz1", (st.M, H))
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_ADD_DROPOUT, p=(self._dx(1, 0), None, g), n=(st.M, H, 1), f=(0.5,), drop=dr), None, None))
        dza, dzb = self.tmp("dd1", (st.M, H)), self.tmp("dctx1", (st.M, H))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(g, proj, st_a[0], st_a[1], pre + "ImgLayerNorm.weight", pre + "ImgLayerNorm.bias", dza, None, st.M, nodrop), None, None))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(g, loc, st_a[2], st_a[3], pre + "LocLayerNorm.weight", pre + "LocLayerNorm.bias", dzb, None, st.M, nodrop), None, None))
        self._img_proj_bwd(b, pre, "image_embeddings", dza, featb)
        self._loc_proj_bwd(b, pre, dzb)
        return b

    def _emb_image_uniter(self, pre):
        cfg, st, H = self.cfg, self.st[1], self.H
        f = self.fwd.ops
        proj, featb = self._img_proj(pre, "image_embeddings", "emb_v")
        loc = self._loc_proj(pre, "emb_v")
        nodrop = L.dropout_cfg(None, 0, 0.0)
        a_n, b_n = self.buf("emb_v_imgn", (st.M, H)), self.buf("emb_v_locn", (st.M, H))
        st_a = [self.buf("emb_v_%s" % s, (st.M,), torch.float32) for s in ("mean_a", "rstd_a", "mean_b", "rstd_b", "mean", "rstd")]
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(proj, None, pre + "image_layer_norm.weight", pre + "image_layer_norm.bias", a_n, None, st_a[0], st_a[1], st.M, nodrop), None, None))
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(loc, None, pre + "image_location_layer_norm.weight", pre + "image_location_layer_norm.bias", b_n, None, st_a[2], st_a[3], st.M, nodrop), None, None))
        y, z = self.buf("emb_v_y", (st.M, H)), self.buf("emb_v_z", (st.M, H))
        dr = self.drop(cfg.hidden_dropout_prob)
        type1 = self.Pm(pre + "token_type_embeddings.weight")[1]
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(a_n, b_n, pre + "v_LayerNorm.weight", pre + "v_LayerNorm.bias", y, z, st_a[4], st_a[5], st.M, dr, post=1, addvec=type1), None, None))
        self.x[1] = y
        b = []
        dz = self.tmp("dz1", (st.M, H))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(self._dx(1, 0), z, st_a[4], st_a[5], pre + "v_LayerNorm.weight", pre + "v_LayerNorm.bias", dz, None, st.M, dr, post=1), None, None))
        # the broadcast token-type row 1 receives the column sum of dz (the text side zeroed / filled its table first)
        part = self.tmp("colsum_partial", (L.lib.vk_rows32(st.M) * H,), torch.float32)
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_COLSUM, p=(dz, part, self.G(pre + "token_type_embeddings.weight")[1]), n=(st.M, H, 1)), None, None))
        dza, dzb = self.tmp("dd1", (st.M, H)), self.tmp("dctx1", (st.M, H))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(dz, proj, st_a[0], st_a[1], pre + "image_layer_norm.weight", pre + "image_layer_norm.bias", dza, None, st.M, nodrop), None, None))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(dz, loc, st_a[2], st_a[3], pre + "image_location_layer_norm.weight", pre + "image_location_layer_norm.bias", dzb, None, st.M, nodrop), None, None))
        self._img_proj_bwd(b, pre, "image_embeddings", dza, featb)
        self._loc_proj_bwd(b, pre, dzb)
        return b

    def _emb_visualbert(self, pre):
        """One LayerNorm over the per-sample concatenation [text | vision] (embeddings.py:389-392): run as two
        row kernels sharing gamma/beta; the single dropout site sees rows b*(T+Rv)+t and b*(T+Rv)+T+r."""
        cfg, H, B = self.cfg, self.H, self.B
        st_t, st_v = self.st
        f = self.fwd.ops
        T, Rv = st_t.L, st_v.L
        zt = self.buf("emb_t_z", (st_t.M, H))
        ea = self.k(L.EmbedArgs(None, None, None, _addr(self.Pm(pre + "word_embeddings.weight")), _addr(self.Pm(pre + "position_embeddings.weight")),
                                _addr(self.Pm(pre + "token_type_embeddings.weight")), None, _addr(zt), st_t.M, T, H,
                                cfg.vocab_size, cfg.max_position_embeddings, cfg.type_vocab_size))
        self.patch("input_ids", ea, "ids")
        self.patch("token_type_ids", ea, "type_ids")
        f.append((L.OP_EMBED_FWD, 0, 0, 0, ea, None, None))
        proj, featb = self._img_proj(pre, "projection", "emb_v")
        # addvec = position_embeddings_visual[0] + token_type_embeddings_visual[1], rebuilt every step by one list op: the two rows are
        # two "slabs" of the fp32 master arena, a fixed distance apart
        vec = self.buf("emb_v_addvec", (H,), torch.float32)
        rows = sorted((self.Pm(pre + "position_embeddings_visual.weight")[0], self.Pm(pre + "token_type_embeddings_visual.weight")[1]), key=lambda r: r.data_ptr())
        gap = rows[1].data_ptr() - rows[0].data_ptr()
        if gap % 16:
            raise NotImplementedError("hidden size must be a multiple of 4")
        f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_SUM_SLABS, p=(vec, rows[0]), n=(gap // 4, 2, H)), None, None))
        yt, yv = self.buf("emb_t_y", (st_t.M, H)), self.buf("emb_v_y", (st_v.M, H))
        zv = self.buf("emb_v_z", (st_v.M, H))
        stats = [self.buf("emb_%s" % s, (m,), torch.float32) for s, m in (("t_mean", st_t.M), ("t_rstd", st_t.M), ("v_mean", st_v.M), ("v_rstd", st_v.M))]
        dr = self.drop(cfg.hidden_dropout_prob)
        seg_t = [(dr.site, T, T + Rv, 0), (dr.site, 0, 0, 0)]
        seg_v = [(dr.site, Rv, T + Rv, T), (dr.site, 0, 0, 0)]
        gn, bn = pre + "LayerNorm.weight", pre + "LayerNorm.bias"
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(zt, None, gn, bn, yt, None, stats[0], stats[1], st_t.M, dr, post=1, segs=seg_t), None, None))
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(proj, None, gn, bn, yv, zv, stats[2], stats[3], st_v.M, dr, post=1, addvec=vec, segs=seg_v), None, None))
        self.x = [yt, yv]
        b = []
        dzt, dzv = self.tmp("dz0", (st_t.M, H)), self.tmp("dz1", (st_v.M, H))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(self._dx(0, 0), zt, stats[0], stats[1], gn, bn, dzt, None, st_t.M, dr, post=1, segs=seg_t), None, None))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(self._dx(1, 0), zv, stats[2], stats[3], gn, bn, dzv, None, st_v.M, dr, post=1, segs=seg_v, accumulate=1), None, None))
        for nm in ("position_embeddings.weight", "token_type_embeddings.weight", "position_embeddings_visual.weight", "token_type_embeddings_visual.weight"):
            self._zero_grad(self.G(pre + nm))
        part = self.tmp("colsum_partial", (L.lib.vk_rows32(st_v.M) * H,), torch.float32)
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_COLSUM, p=(dzv, part, self.G(pre + "position_embeddings_visual.weight")[0]), n=(st_v.M, H, 0)), None, None))
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_COLSUM, p=(dzv, part, self.G(pre + "token_type_embeddings_visual.weight")[1]), n=(st_v.M, H, 0)), None, None))
        self._img_proj_bwd(b, pre, "projection", dzv, featb)
        eb = self.k(L.EmbedBwdArgs(_addr(dzt), None, None, None, _addr(self.G(pre + "word_embeddings.weight")), _addr(self.G(pre + "position_embeddings.weight")),
                                   _addr(self.G(pre + "token_type_embeddings.weight")), st_t.M, T, H, cfg.type_vocab_size, cfg.vocab_size,
                                   cfg.max_position_embeddings))
        self.patch("input_ids", eb, "ids")
        self.patch("token_type_ids", eb, "type_ids")
        b.append((L.OP_WAIT_SIDE, 12, 0, 0, None, None, None))      # after the tied LM decoder's weight gradient (heads, side stream)
        b.append((L.OP_EMBED_BWD, 0, 0, 0, eb, None, None))
        return b

    def _emb_vlbert(self, pre):
        """VL-BERT embeddings (volta/embeddings.py:240-301): box geometry sin/cos + appearance -> dropout -> Linear(2F -> H)
        -> ReLU = final; vision token = LN_obj(final) + (object / END) embedding + position + type 2; text token =
        word + LN_text(final of the LAST region of the sample) + position + type; ONE LayerNorm over [text | vision].
        Position ids follow the reference, including its expanded-view quirk (see prepare_step)."""
        cfg, H, B = self.cfg, self.H, self.B
        st_t, st_v = self.st
        T, K = st_t.L, st_v.L
        F_, dim = cfg.v_feature_size, cfg.v_coordinate_embeddings_dim
        W = 8 * dim + F_
        if W != 2 * F_ or W % 64 or cfg.v_hidden_size != cfg.hidden_size:
            raise NotImplementedError("VL-BERT embedding geometry outside the reference configs (8*dim must equal v_feature_size)")
        mvrc = cfg.visual_target_weights.get("6", 0) > 0      # masked regions get a word of their own (embeddings.py:191,262-263)
        nword = 3 if mvrc else 2
        f, dev = self.fwd.ops, self.dev
        i64 = dict(dtype=torch.int64, device=dev)
        nodrop = L.dropout_cfg(None, 0, 0.0)
        # ---- static index tensors and per-step position ids (filled in prepare_step)
        is_last = torch.zeros(B, K, **i64)
        is_last[:, -1] = 1
        self.bufs["vl_is_last"] = is_last = is_last.view(-1).contiguous()
        self.bufs["vl_twos"] = twos = torch.full((st_v.M,), 2, **i64)
        self.bufs["vl_tpos"] = tpos = torch.zeros(st_t.M, **i64)
        self.bufs["vl_opos"] = opos = torch.zeros(st_v.M, **i64)
        self.bufs["vl_last_rows"] = last_rows = (torch.arange(B, device=dev, dtype=torch.int32) * K + (K - 1)).contiguous()
        self.bufs["vl_row2b"] = row2b = (torch.arange(st_t.M, device=dev, dtype=torch.int32) // T).contiguous()
        self.bufs["vl_cntB"] = cntB = torch.tensor([B], device=dev, dtype=torch.int32)
        self.bufs["vl_cntMt"] = cntMt = torch.tensor([st_t.M], device=dev, dtype=torch.int32)
        vtab = self.buf("vl_vtab", (nword, H), torch.float32)       # rows: object word, END word (last region), masked-region word
        dvtab = self.buf("vl_dvtab", (nword, H), torch.float32)
        # ---- forward: position ids from this step's input_ids, and the (object | END | masked) word table from the parameters
        g = self.generic(L.FN_VLBERT_POSITIONS, p=(None, tpos, opos), n=(B, T, K))
        self.patch("input_ids", g, "p", 0)
        f.append((L.OP_GENERIC, 0, 0, 0, g, None, None))
        for r, nm in enumerate(("object_linguistic_embeddings", "end_embedding", "object_mask_word_embedding")[:nword]):
            f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_COPY, p=(vtab[r], self.Pm(pre + nm + ".weight")[0]), n=(H * 4,)), None, None))
        x4 = self.buf("vl_x4096", (st_v.M, W))
        zflag = self.buf("vl_zero_flag", (st_v.M,), torch.int32)
        dr0 = self.drop(cfg.v_attention_probs_dropout_prob)
        g = self.generic(L.FN_VLBERT_PREP, p=(None, None, self.Pm(pre + "object_mask_visual_embedding.weight"), x4, zflag), n=(st_v.M, F_, dim, cfg.num_locs), drop=dr0)
        self.patch("image_loc", g, "p", 0)
        self.patch("image_feat", g, "p", 1)
        f.append((L.OP_GENERIC, 0, 0, 0, g, None, None))
        word_ids = is_last
        if mvrc:
            word_ids = self.buf("vl_word_ids", (st_v.M,), torch.int64)
            f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_VLBERT_OBJ_IDS, p=(zflag, word_ids), n=(st_v.M, K)), None, None))
        final = self.buf("vl_final", (st_v.M, H))
        wds = pre + "obj_downsample.1"
        self.gemm(f, L.NT, L.EPI_RELU, [self.prob(x4, self.W(wds + ".weight"), final, st_v.M, H, W, W, W, H, bias=self.Pm(wds + ".bias"))])
        obj_vis = self.buf("vl_obj_vis", (st_v.M, H))
        so = [self.buf("vl_%s" % n_, (m,), torch.float32) for n_, m in (("mean_o", st_v.M), ("rstd_o", st_v.M), ("mean_x", B), ("rstd_x", B),
                                                                       ("mean_t", st_t.M), ("rstd_t", st_t.M), ("mean_v", st_v.M), ("rstd_v", st_v.M))]
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(final, None, pre + "visual_ln_object.weight", pre + "visual_ln_object.bias", obj_vis, None, so[0], so[1], st_v.M, nodrop), None, None))
        vz = self.buf("emb_v_z", (st_v.M, H))
        ptab, ttab = pre + "position_embeddings.weight", pre + "token_type_embeddings.weight"
        ev = self.k(L.EmbedArgs(_addr(word_ids), _addr(twos), _addr(opos), _addr(vtab), _addr(self.Pm(ptab)), _addr(self.Pm(ttab)), _addr(obj_vis), _addr(vz),
                                st_v.M, K, H, nword, cfg.max_position_embeddings, cfg.type_vocab_size))
        f.append((L.OP_EMBED_FWD, 0, 0, 0, ev, None, None))
        flast = self.buf("vl_final_last", (B, H))
        f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_GATHER, p=(final, last_rows, cntB, flast), n=(H, B)), None, None))
        tv = self.buf("vl_tv", (B, H))
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(flast, None, pre + "visual_ln_text.weight", pre + "visual_ln_text.bias", tv, None, so[2], so[3], B, nodrop), None, None))
        tvx = self.buf("vl_tv_exp", (st_t.M, H))
        f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_GATHER, p=(tv, row2b, cntMt, tvx), n=(H, st_t.M)), None, None))
        tz = self.buf("emb_t_z", (st_t.M, H))
        et = self.k(L.EmbedArgs(None, None, _addr(tpos), _addr(self.Pm(pre + "word_embeddings.weight")), _addr(self.Pm(ptab)), _addr(self.Pm(ttab)), _addr(tvx), _addr(tz),
                                st_t.M, T, H, cfg.vocab_size, cfg.max_position_embeddings, cfg.type_vocab_size))
        self.patch("input_ids", et, "ids")
        self.patch("token_type_ids", et, "type_ids")
        f.append((L.OP_EMBED_FWD, 0, 0, 0, et, None, None))
        yt, yv = self.buf("emb_t_y", (st_t.M, H)), self.buf("emb_v_y", (st_v.M, H))
        dr1 = self.drop(cfg.hidden_dropout_prob)
        seg_t = [(dr1.site, T, T + K, 0), (dr1.site, 0, 0, 0)]
        seg_v = [(dr1.site, K, T + K, T), (dr1.site, 0, 0, 0)]
        gn, bn = pre + "LayerNorm.weight", pre + "LayerNorm.bias"
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(tz, None, gn, bn, yt, None, so[4], so[5], st_t.M, dr1, post=1, segs=seg_t), None, None))
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(vz, None, gn, bn, yv, None, so[6], so[7], st_v.M, dr1, post=1, segs=seg_v), None, None))
        self.x = [yt, yv]
        # ---- backward
        b = []
        dzt, dzv = self.tmp("dz0", (st_t.M, H)), self.tmp("dz1", (st_v.M, H))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(self._dx(0, 0), tz, so[4], so[5], gn, bn, dzt, None, st_t.M, dr1, post=1, segs=seg_t), None, None))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(self._dx(1, 0), vz, so[6], so[7], gn, bn, dzv, None, st_v.M, dr1, post=1, segs=seg_v, accumulate=1), None, None))
        self._zero_grad(self.G(ptab))
        self._zero_grad(self.G(ttab))
        # text tokens: word / position / type tables, and the per-sample visual vector
        eb = self.k(L.EmbedBwdArgs(_addr(dzt), None, None, _addr(tpos), _addr(self.G(pre + "word_embeddings.weight")), _addr(self.G(ptab)), _addr(self.G(ttab)),
                                   st_t.M, T, H, cfg.type_vocab_size, cfg.vocab_size, cfg.max_position_embeddings))
        self.patch("input_ids", eb, "ids")
        self.patch("token_type_ids", eb, "type_ids")
        b.append((L.OP_WAIT_SIDE, 12, 0, 0, None, None, None))      # after the tied LM decoder's weight gradient (heads, side stream)
        b.append((L.OP_EMBED_BWD, 0, 0, 0, eb, None, None))
        dtv = self.buf("vl_dtv", (B, H))
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_ROWGROUP_SUM, p=(dzt, dtv), n=(B, T, H)), None, None))
        dflast = self.buf("vl_dfinal_last", (B, H))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(dtv, flast, so[2], so[3], pre + "visual_ln_text.weight", pre + "visual_ln_text.bias", dflast, None, B, nodrop), None, None))
        # vision tokens: (object | END) embedding, position, type 2, then LN_obj
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_MEMSET, p=(dvtab,), n=(dvtab.numel() * 4, 0)), None, None))
        evb = self.k(L.EmbedBwdArgs(_addr(dzv), _addr(word_ids), _addr(twos), _addr(opos), _addr(dvtab), _addr(self.G(ptab)), _addr(self.G(ttab)),
                                    st_v.M, K, H, cfg.type_vocab_size, nword, cfg.max_position_embeddings))
        b.append((L.OP_EMBED_BWD, 0, 0, 0, evb, None, None))
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_COPY, p=(self.G(pre + "object_linguistic_embeddings.weight"), dvtab[0]), n=(H * 4,)), None, None))
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_COPY, p=(self.G(pre + "end_embedding.weight"), dvtab[1]), n=(H * 4,)), None, None))
        if mvrc:
            b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_COPY, p=(self.G(pre + "object_mask_word_embedding.weight"), dvtab[2]), n=(H * 4,)), None, None))
        dfinal = self.tmp("dd1", (st_v.M, H))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(dzv, final, so[0], so[1], pre + "visual_ln_object.weight", pre + "visual_ln_object.bias", dfinal, None, st_v.M, nodrop), None, None))
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_SCATTER_ADD, p=(dflast, last_rows, cntB, dfinal), n=(H, B)), None, None))
        dpre = self.tmp("dctx1", (st_v.M, H))
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_RELU_BWD, p=(dfinal, final, dpre), n=(st_v.M * H,)), None, None))
        self.gemm(b, L.TN, L.EPI_F32, [self.prob(dpre, x4, self.G(wds + ".weight"), H, W, st_v.M, H, W, W, bias_grad=self.G(wds + ".bias"))])
        dx4 = self.buf("vl_dx4096", (st_v.M, W))
        self.gemm(b, L.NN, L.EPI_BF16, [self.prob(dpre, self.W(wds + ".weight"), dx4, st_v.M, W, H, H, W, W)])
        part = self.tmp("vl_mask_partial", (L.lib.vk_rows32(st_v.M) * F_,), torch.float32)
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_VLBERT_MASKGRAD, p=(dx4, zflag, part, self.G(pre + "object_mask_visual_embedding.weight")),
                                                       n=(st_v.M, F_, W, 8 * dim), drop=dr0), None, None))
        return b

    # ---------------------------------------------------------------- encoder sub-layers
    def _names(self, n, typ):
        """Parameter names per modality for sub-layer n (shared sub-layers: the text names for both)."""
        cfg = self.cfg
        p = "bert.encoder.layer.%d." % n
        shared = n in cfg.shared_sublayers
        out = []
        for m in range(2):
            v = "v_" if (m == 1 and not shared) else ""
            if typ == "attn":
                out.append(dict(q=p + "attention_self.%squery" % v, k=p + "attention_self.%skey" % v, v=p + "attention_self.%svalue" % v,
                                o=p + "attention_output.%sdense" % v, ln=p + "attention_output.%sLayerNorm" % v))
            else:
                out.append(dict(up=p + "intermediate.%sdense" % v, down=p + "output.%sdense" % v, ln=p + "output.%sLayerNorm" % v))
        return out, shared

    def _attn_sublayer(self, n):
        cfg, B = self.cfg, self.B
        f = self.fwd.ops
        gate = [[int(n in cfg.tt_attn_sublayers), int(n in cfg.tv_attn_sublayers)], [int(n in cfg.vt_attn_sublayers), int(n in cfg.vv_attn_sublayers)]]
        act = [bool(gate[0][0] or gate[0][1]), bool(gate[1][0] or gate[1][1])]
        names, shared = self._names(n, "attn")
        ms = [m for m in range(2) if act[m]]
        tag = "L%d_" % n
        x_in = list(self.x)
        x8_in = list(self.x8)
        # widths: Hm = the stream's hidden size, Ha = the sub-layer's attention width for that stream, nhm heads of dh (encoders.py:164-206)
        Hm = [self.st[0].H, self.st[1].H]
        Ha = [cfg.sublayer2attn_hidden_size.get(str(n), cfg.hidden_size), cfg.sublayer2v_attn_hidden_size.get(str(n), cfg.v_hidden_size)]
        nhm = [cfg.sublayer2num_attention_heads.get(str(n), cfg.num_attention_heads), cfg.sublayer2v_num_attention_heads.get(str(n), cfg.v_num_attention_heads)]
        dh = [Ha[m] // nhm[m] for m in range(2)]
        for m in ms:
            if Ha[m] % nhm[m] or dh[m] not in (32, 64, 96, 128) or Ha[m] % 64:
                raise NotImplementedError("attention width %d with %d heads (head sizes 32, 64, 96, 128; widths multiples of 64)" % (Ha[m], nhm[m]))
        cross = gate[0][1] or gate[1][0]
        if shared and len(ms) == 2 and (Hm[0] != Hm[1] or Ha[0] != Ha[1]):
            raise ValueError("a shared attention sub-layer needs equal widths in both streams")
        if cross and (dh[0] != dh[1] or nhm[0] != nhm[1]):
            raise ValueError("cross-modal attention needs the same head count and size in both streams (sub-layer %d: %d x %d vs %d x %d)"
                             % (n, nhm[0], dh[0], nhm[1], dh[1]))
        qkv = {m: self.buf(tag + "qkv%d" % m, (self.st[m].M, 3 * Ha[m])) for m in ms}
        ctx = {m: self.buf(tag + "ctx%d" % m, (self.st[m].M, Ha[m])) for m in ms}
        lse = {m: self.buf(tag + "lse%d" % m, (B * nhm[m] * self.st[m].L,), torch.float32) for m in ms}
        d = {m: self.buf(tag + "z%d" % m, (self.st[m].M, Hm[m])) for m in ms}
        y = {m: self.buf(tag + "y%d" % m, (self.st[m].M, Hm[m])) for m in ms}
        mean = {m: self.buf(tag + "mean%d" % m, (self.st[m].M,), torch.float32) for m in ms}
        rstd = {m: self.buf(tag + "rstd%d" % m, (self.st[m].M,), torch.float32) for m in ms}

        def wqkv(m, which):
            nm = names[m]
            return self.arena.span([nm["q"] + ".weight", nm["k"] + ".weight", nm["v"] + ".weight"], which, (3 * Ha[m], Hm[m]))

        def bqkv(m, which):
            nm = names[m]
            return self.arena.span([nm["q"] + ".bias", nm["k"] + ".bias", nm["v"] + ".bias"], which, (3 * Ha[m],))

        if self.fp8:
            self.gemm_fp8(f, L.EPI_BF16, [(x8_in[m] or x_in[m], wqkv(m, "master"), qkv[m], bqkv(m, "master"), None) for m in ms])
        else:
            self.gemm(f, L.NT, L.EPI_BF16, [self.prob(x_in[m], wqkv(m, "shadow"), qkv[m], self.st[m].M, 3 * Ha[m], Hm[m], Hm[m], Hm[m], 3 * Ha[m], bias=bqkv(m, "master")) for m in ms])
        # dropout sites in the reference's call order: tt, tv, then vv, vt (encoders.py:294-295, 309-310)
        drops = {}
        if gate[0][0]:
            drops[(0, 0)] = self.drop(cfg.attention_probs_dropout_prob)
        if gate[0][1]:
            drops[(0, 1)] = self.drop(cfg.attention_probs_dropout_prob)
        if gate[1][1]:
            drops[(1, 1)] = self.drop(cfg.v_attention_probs_dropout_prob)
        if gate[1][0]:
            drops[(1, 0)] = self.drop(cfg.v_attention_probs_dropout_prob)
        # one launch covers every gate block when the active streams share head count and size (every ctrl_* config, and the co-attention
        # sub-layers of vilbert_base); two self-attentions with different heads (vilbert_base: 12 x 64 text, 8 x 128 vision) are two launches
        if len(ms) == 2 and not cross and (nhm[0], dh[0]) != (nhm[1], dh[1]):
            launches = [[[gate[0][0], 0], [0, 0]], [[0, 0], [0, gate[1][1]]]]
        else:
            launches = [gate]
        attn = []
        for gl in launches:
            mq = [m for m in range(2) if gl[m][0] or gl[m][1] or gl[0][m] or gl[1][m]]
            aa = L.AttnArgs()
            for m in mq:
                base = qkv[m].data_ptr()
                aa.q[m], aa.k[m], aa.v[m] = base, base + 2 * Ha[m], base + 4 * Ha[m]
                aa.ld[m], aa.L[m] = 3 * Ha[m], self.st[m].L
                aa.mask[m] = self.masks[m].data_ptr()
                aa.ctx[m], aa.ldo[m], aa.lse[m] = ctx[m].data_ptr(), Ha[m], lse[m].data_ptr()
            m0 = mq[0]
            aa.B, aa.nh, aa.scale, aa.dh = B, nhm[m0], 1.0 / math.sqrt(float(dh[m0])), dh[m0]
            for i in range(2):
                for j in range(2):
                    aa.gate[i][j] = gl[i][j]
                    aa.drop[i][j] = drops.get((i, j), L.dropout_cfg(None, 0, 0.0)) if gl[i][j] else L.dropout_cfg(None, 0, 0.0)
            if self.attn_maps:
                pb = {}
                for i in range(2):
                    for j in range(2):
                        if gl[i][j]:
                            pb[(i, j)] = self.buf(tag + "probs%d%d" % (i, j), (B, nhm[m0], self.st[i].L, self.st[j].L), torch.float32)
                            aa.probs[i][j] = pb[(i, j)].data_ptr()
                self.attn_map_info.append(dict(n=n, probs=pb, qkv={m: qkv[m] for m in mq}, Ha={m: Ha[m] for m in mq}, nh=nhm[m0], dh=dh[m0]))
            self.k(aa)
            # rows the generic kernels cannot hold in LDS fail HERE, when the plan is built, not at the first backward launch (their backward
            # keeps two row images of both modalities: ~491 keys at head size 64, ~258 at 128, less than the forward's 512)
            for bwd_pass in (0, 1):
                need = L.lib.vk_gated_attn_lds_bytes(C.byref(aa), bwd_pass)
                if need > 160 * 1024:
                    raise NotImplementedError("attention sub-layer %d: %d + %d rows at head size %d need %d bytes of LDS in the %s pass (160 KiB per workgroup)"
                                              % (n, self.st[0].L, self.st[1].L, dh[m0], need, "backward" if bwd_pass else "forward"))
            f.append((L.OP_ATTN_FWD, 0, 0, 0, aa, None, None))
            attn.append((aa, mq, gl))
        self.gemm(f, L.NT, L.EPI_BF16, [self.prob(ctx[m], self.W(names[m]["o"] + ".weight"), d[m], self.st[m].M, Hm[m], Ha[m], Ha[m], Ha[m], Hm[m], bias=self.Pm(names[m]["o"] + ".bias")) for m in ms])
        odrop, lnf = {}, []
        for m in ms:
            odrop[m] = self.drop(cfg.hidden_dropout_prob if m == 0 else cfg.v_hidden_dropout_prob)
            self.x8[m] = self.fp8_hidden(m) if self.fp8 else None
            lnf.append(self.ln_args(d[m], x_in[m], names[m]["ln"] + ".weight", names[m]["ln"] + ".bias", y[m], d[m], mean[m], rstd[m], self.st[m].M, odrop[m],
                                    fp8_out=self.x8[m], H=Hm[m]))
            self.x[m] = y[m]
        self._ln_pair(f, L.OP_LN_FWD, lnf)                     # both streams in one launch when their widths agree
        # ------------- backward
        b = []
        dz, dd, dctx, dqkv, dxn, dxi = {}, {}, {}, {}, {}, {}
        lnb = []
        for i, m in enumerate(ms):
            dxi[m], dxn[m] = self._dx_step(m)
            acc = 1 if (shared and i > 0) else 0
            par = self.sub_k % 2
            dz[m] = self.tmp("dz%d_%d" % (m, par), (self.st[m].M, Hm[m]))
            dd[m] = self.tmp("dd%d_%d" % (m, par), (self.st[m].M, Hm[m])) if self.train else dz[m]
            lnb.append(self.ln_bwd_args(dxi[m], d[m], mean[m], rstd[m], names[m]["ln"] + ".weight", names[m]["ln"] + ".bias", dz[m],
                                        dd[m] if self.train else None, self.st[m].M, odrop[m], accumulate=acc, defer=True, H=Hm[m]))
            wtag = "" if Ha[m] == Hm[m] == self.H else "_%d" % Ha[m]      # temporaries are shared by name: other widths get their own
            dctx[m] = self.tmp("dctx%d%s" % (m, wtag), (self.st[m].M, Ha[m]))
            dqkv[m] = self.tmp("dqkv%d_%d%s" % (m, par, wtag), (self.st[m].M, 3 * Ha[m]))
        self._ln_pair(b, L.OP_LN_BWD, lnb)
        self.gemm(b, L.NN, L.EPI_BF16, [self.prob(dd[m], self.W(names[m]["o"] + ".weight"), dctx[m], self.st[m].M, Ha[m], Hm[m], Hm[m], Ha[m], Ha[m]) for m in ms])
        for m in ms:
            if not (gate[0][m] or gate[1][m]):       # K/V of this modality unused: their gradient is zero
                b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_MEMSET, p=(dqkv[m],), n=(dqkv[m].numel() * 2, 0)), None, None))
        for aa, mq, gl in attn:
            ab = L.AttnBwdArgs()
            for m in mq:
                ab.dctx[m] = dctx[m].data_ptr()
                base = dqkv[m].data_ptr()
                ab.dq[m], ab.dk[m], ab.dv[m], ab.ldg[m] = base, base + 2 * Ha[m], base + 4 * Ha[m], 3 * Ha[m]
            self.k(ab)
            b.append((L.OP_ATTN_BWD, 0, 0, 0, aa, ab, None))
        # all weight gradients of the sub-layer in ONE grouped launch (more workgroups per CU), listed in front of the Q|K|V
        # dgrad: they need dqkv, not its product, and start beside that GEMM instead of beside the next sub-layer's LayerNorm backward
        self._wgrad(b, ms, shared, [lambda m: (dd[m], ctx[m], self.G(names[m]["o"] + ".weight"), self.G(names[m]["o"] + ".bias"), Hm[m], Ha[m], Hm[m], Ha[m]),
                                    lambda m: (dqkv[m], x_in[m], wqkv(m, "grad"), bqkv(m, "grad"), 3 * Ha[m], Hm[m], 3 * Ha[m], Hm[m])])
        self.gemm(b, L.NN, L.EPI_ADDR, self.retire_on([self.prob(dqkv[m], wqkv(m, "shadow"), dxn[m], self.st[m].M, Hm[m], 3 * Ha[m], 3 * Ha[m], Hm[m], Hm[m], R=dz[m], ldr=Hm[m]) for m in ms]))
        return b

    @staticmethod
    def _soft_pair_ok(shapes):
        """A producer [M, I, H] -> consumer [M, H, I] pair takes the soft boundary when every tile is whole (256-row blocks, 256-wide producer
        tiles, 192-wide consumer tiles) and both launches are large enough for the 256-row geometries the hand-off is built on."""
        if any(M % 256 or I % 256 or H % 192 or H % 64 for M, I, H in shapes):
            return False
        t_up = sum((M // 256) * (I // 256) for M, I, H in shapes)
        t_down = sum((M // 256) * -(-H // 256) for M, I, H in shapes)
        return t_up >= 160 and t_down >= 160

    def _ln_pair(self, ops, kind, jobs):
        """One LayerNorm launch for both streams when their widths agree (the kernels share the launch between two jobs of equal width),
        one launch per stream otherwise."""
        if len(jobs) == 2 and jobs[0].H == jobs[1].H:
            ops.append((kind, 0, 0, 0, jobs[0], jobs[1], None))
        else:
            for j in jobs:
                ops.append((kind, 0, 0, 0, j, None, None))

    def _ffn_sublayer(self, n):
        cfg = self.cfg
        f = self.fwd.ops
        # per-stream widths: hidden Hm, intermediate Im (config/vilbert_base.json: 768 / 3072 text, 1024 / 1024 vision; encoders.py:459-460,514-515)
        Hm = [self.st[0].H, self.st[1].H]
        Im = [cfg.sublayer2intermediate_size.get(str(n), cfg.intermediate_size), cfg.sublayer2v_intermediate_size.get(str(n), cfg.v_intermediate_size)]
        for m in range(2):
            if Im[m] % 64:
                raise NotImplementedError("intermediate sizes must be multiples of 64")
        act = [n in cfg.t_ff_sublayers, n in cfg.v_ff_sublayers]
        names, shared = self._names(n, "ff")
        ms = [m for m in range(2) if act[m]]
        tag = "L%d_" % n
        x_in = list(self.x)
        x8_in = list(self.x8)
        if shared and len(ms) == 2 and (Hm[0] != Hm[1] or Im[0] != Im[1]):
            raise ValueError("a shared feed-forward sub-layer needs equal widths in both streams")
        h = {m: self.buf(tag + "h%d" % m, (self.st[m].M, Im[m])) for m in ms}
        gp = {m: self.buf(tag + "gp%d" % m, (self.st[m].M, Im[m])) for m in ms}
        d = {m: self.buf(tag + "z%d" % m, (self.st[m].M, Hm[m])) for m in ms}
        y = {m: self.buf(tag + "y%d" % m, (self.st[m].M, Hm[m])) for m in ms}
        mean = {m: self.buf(tag + "mean%d" % m, (self.st[m].M,), torch.float32) for m in ms}
        rstd = {m: self.buf(tag + "rstd%d" % m, (self.st[m].M,), torch.float32) for m in ms}
        if self.fp8:
            # the GELU output also leaves the epilogue as e4m3 with one static scale (the FFN-down projection's A operand), de-quantised
            # there through scale_a = 1 / multiplier
            up, specs = [], []
            for m in ms:
                Mm = self.st[m].M
                h8 = self.tmp("fp8_h%d_%d" % (m, Mm), (Mm, Im[m]), torch.uint8)
                if "fp8_hscale_%d" % Mm not in self.bufs:
                    self.bufs["fp8_hscale_%d" % Mm] = torch.full((Mm,), 1.0 / self.H8_MUL, dtype=torch.float32, device=self.dev)
                up.append((x8_in[m] or x_in[m], self.Pm(names[m]["up"] + ".weight"), h[m], self.Pm(names[m]["up"] + ".bias"), gp[m], (h8, self.H8_MUL)))
                specs.append(((h8, self.bufs["fp8_hscale_%d" % Mm]), self.Pm(names[m]["down"] + ".weight"), d[m], self.Pm(names[m]["down"] + ".bias"), None))
            self.gemm_fp8(f, L.EPI_GELU, up)
            self.gemm_fp8(f, L.EPI_BF16, specs)
        else:
            up = [self.prob(x_in[m], self.W(names[m]["up"] + ".weight"), h[m], self.st[m].M, Im[m], Hm[m], Hm[m], Hm[m], Im[m], bias=self.Pm(names[m]["up"] + ".bias"), C2=gp[m]) for m in ms]
            down = [self.prob(h[m], self.W(names[m]["down"] + ".weight"), d[m], self.st[m].M, Hm[m], Im[m], Im[m], Im[m], Hm[m], bias=self.Pm(names[m]["down"] + ".bias")) for m in ms]
            g_up = g_down = 0
            pair_ok = self._soft_pair_ok([(self.st[m].M, Im[m], Hm[m]) for m in ms])
            if self.chain != "0" and pair_ok:
                for m, pu, pd in zip(ms, up, down):
                    pu.sig, pu.err = self.soft_counters(self.st[m].M // 256)
                    pd.dep, pd.err, pd.dep_need = pu.sig, pu.err, Im[m] // 256
                self.gemm_chain(f, L.NT, L.EPI_GELU, up, L.EPI_BF16, down)
                up = down = None
            elif self.soft and pair_ok:
                # FFN-down starts on the CUs FFN-up's last round leaves idle: row block r of h is handed over by counter (12 column tiles of 256)
                for m, pu, pd in zip(ms, up, down):
                    pu.sig, pu.err = self.soft_counters(self.st[m].M // 256)
                    pd.dep, pd.err, pd.dep_need = pu.sig, pu.err, Im[m] // 256
                g_up, g_down = 258, 259 | L.GEMM_SOFT_START
            if up is not None:
                self.gemm(f, L.NT, L.EPI_GELU, up, geometry=g_up)
                self.gemm(f, L.NT, L.EPI_BF16, down, geometry=g_down)
        odrop, lnf = {}, []
        for m in ms:
            odrop[m] = self.drop(cfg.hidden_dropout_prob if m == 0 else cfg.v_hidden_dropout_prob)
            self.x8[m] = self.fp8_hidden(m) if self.fp8 else None
            lnf.append(self.ln_args(d[m], x_in[m], names[m]["ln"] + ".weight", names[m]["ln"] + ".bias", y[m], d[m], mean[m], rstd[m], self.st[m].M, odrop[m],
                                    fp8_out=self.x8[m], H=Hm[m]))
            self.x[m] = y[m]
        self._ln_pair(f, L.OP_LN_FWD, lnf)                     # both streams in one launch when their widths agree
        b = []
        dz, dd, du, dxn, dxi = {}, {}, {}, {}, {}
        lnb = []
        for i, m in enumerate(ms):
            dxi[m], dxn[m] = self._dx_step(m)
            acc = 1 if (shared and i > 0) else 0
            par = self.sub_k % 2
            dz[m] = self.tmp("dz%d_%d" % (m, par), (self.st[m].M, Hm[m]))
            dd[m] = self.tmp("dd%d_%d" % (m, par), (self.st[m].M, Hm[m])) if self.train else dz[m]
            lnb.append(self.ln_bwd_args(dxi[m], d[m], mean[m], rstd[m], names[m]["ln"] + ".weight", names[m]["ln"] + ".bias", dz[m],
                                        dd[m] if self.train else None, self.st[m].M, odrop[m], accumulate=acc, defer=True, H=Hm[m]))
            wtag = "" if Im[m] == self.I else "_%d" % Im[m]               # temporaries are shared by name: other widths get their own
            du[m] = self.tmp("du%d_%d%s" % (m, par, wtag), (self.st[m].M, Im[m]))
        self._ln_pair(b, L.OP_LN_BWD, lnb)
        d1 = [self.prob(dd[m], self.W(names[m]["down"] + ".weight"), du[m], self.st[m].M, Im[m], Hm[m], Hm[m], Im[m], Im[m], R=gp[m], ldr=Im[m]) for m in ms]
        d2 = [self.prob(du[m], self.W(names[m]["up"] + ".weight"), dxn[m], self.st[m].M, Hm[m], Im[m], Im[m], Hm[m], Hm[m], R=dz[m], ldr=Hm[m]) for m in ms]
        chained = self.chain == "all" and not self.fp8 and self._soft_pair_ok([(self.st[m].M, Im[m], Hm[m]) for m in ms])
        if chained:
            # both dgrads in one launch; the weight gradients (which need du, the first one's output) start behind it
            for m, p1, p2 in zip(ms, d1, d2):
                p1.sig, p1.err = self.soft_counters(self.st[m].M // 256, backward=True)
                p2.dep, p2.err, p2.dep_need = p1.sig, p1.err, Im[m] // 256
            self.gemm_chain(b, L.NN, L.EPI_MULR, d1, L.EPI_ADDR, d2)
        else:
            self.gemm(b, L.NN, L.EPI_MULR, d1)
        # The weight gradients need dd, h, du and x_in: everything but the LAST dgrad's output.  Their side-stream block is listed in front of that
        # dgrad, so that it starts beside a GEMM (two MFMA-bound launches share the chip without loss) instead of beside the LayerNorm backward
        # that follows, which it used to keep waiting for CUs (profiles/r03_experiments.md: 17.00 / 16.94 -> 16.52 / 16.54 ms per step).
        gate_mode, self.side_gate = self.side_gate, self.side_gate and not chained      # behind a chain the block starts on the fork event
        self._wgrad(b, ms, shared, [lambda m: (dd[m], h[m], self.G(names[m]["down"] + ".weight"), self.G(names[m]["down"] + ".bias"), Hm[m], Im[m], Hm[m], Im[m]),
                                    lambda m: (du[m], x_in[m], self.G(names[m]["up"] + ".weight"), self.G(names[m]["up"] + ".bias"), Im[m], Hm[m], Im[m], Hm[m])])
        self.side_gate = gate_mode
        if not chained:
            self.gemm(b, L.NN, L.EPI_ADDR, self.retire_on(d2))
        return b

    def _wgrad(self, b, ms, shared, specs):
        """dW[Mo, No] = dY^T X (+ bias grad) for every spec and modality in ONE grouped TN launch.  The outputs are
        small (<= 3072 x 768) and the contraction (B*L rows) long, so every problem is split along K into chunks of
        ~5120 rows: each chunk is an ordinary problem writing its own fp32 slab (weights | bias), and one reduction
        pass sums the slabs into the gradient arena (plain stores: float atomics would cost 4x the slab traffic).
        This turns 36-144 long tiles into >= 256 balanced ones that take the 256x256 geometry.  Weights shared by
        both modalities simply sum the slabs of both."""
        jobs = []            # (dst weight grad, dst bias grad, Mo, No, [(dY, X, rows, lda, ldb), ...])
        for spec in specs:
            per_w = {}
            for m in ms:
                dY, X, gW, gB, Mo, No, lda, ldb = spec(m)
                key = gW.data_ptr()
                per_w.setdefault(key, (gW, gB, Mo, No, []))[4].append((dY, X, self.st[m].M, lda, ldb))
            jobs += list(per_w.values())
        # K-chunk length: ~5120 rows, halved (down to ~1280) while the whole group still yields fewer than 130 tiles of 256 x 256 (half the
        # CUs) -- the attention sub-layers' group ([768, 768] and [2304, 768] outputs) is 108 tiles of 150 K-steps at 5120, i.e. 42 % of the
        # CUs busy for 127 us; more, shorter chunks fill the chip, but every halving doubles the slab traffic: past half the chip it costs more
        # than the idle CUs (a threshold of 200 cut the text-only groups into 288 tiles of 40 K-steps: +0.08 ms per step)
        chunk_rows = 5120.0
        tiles = lambda cr: sum(-(-Mo // 256) * -(-No // 256) * sum(max(1, int(round(rows / cr))) for _, _, rows, _, _ in srcs) for _, _, Mo, No, srcs in jobs)
        while chunk_rows > 1280 and tiles(chunk_rows) < 130:     # (round 3 sweep, profiles/r03_experiments.md: 130 = half the CUs; 200 cut the text-only groups once more, +0.08 ms)
            chunk_rows /= 2
        probs, reduces = [], []
        for gW, gB, Mo, No, srcs in jobs:
            chunks = []
            for dY, X, rows, lda, ldb in srcs:
                ns = max(1, int(round(rows / chunk_rows)))
                step = -(-rows // ns)
                step = -(-step // 64) * 64
                r0 = 0
                while r0 < rows:
                    chunks.append((dY[r0:], X[r0:], min(step, rows - r0), lda, ldb))
                    r0 += step
            if len(chunks) == 1:
                dY, X, rows, lda, ldb = chunks[0]
                probs.append(self.prob(dY, X, gW, Mo, No, rows, lda, ldb, No, bias_grad=gB))
                continue
            stride = _round_up(Mo * No + Mo, 4)
            slab = self._slab(len(chunks) * stride)
            for i, (dY, X, rows, lda, ldb) in enumerate(chunks):
                base = slab[i * stride:]
                probs.append(self.prob(dY, X, base, Mo, No, rows, lda, ldb, No, bias_grad=base[Mo * No:]))
            reduces.append((gW, slab, stride, len(chunks), Mo * No))
            reduces.append((gB, slab[Mo * No:], stride, len(chunks), Mo))
        assert len(probs) <= 32, "too many wgrad problems in one group"
        gate = self._gate_words() if self.side_gate else None
        b.append((L.OP_SIDE_BEGIN, 1 if gate is not None else 0, 0, 0, None, None, None))
        if gate is not None:
            self._n_gate = getattr(self, "_n_gate", 0) + 1
            assert self._n_gate < gate.numel()
            flag = gate.data_ptr() + 8 * self._n_gate
            b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_GATE, p=(flag, gate.data_ptr(), self.bufs["gate_err"]), n=(1000000,)), None, None))
            self._gate_flag = (flag, gate.data_ptr())          # the caller hangs it on the dgrad it lists behind this block (retire_on)
        if self.side_delay_us > 0:
            # a pure delay (one wave, no LDS) at the head of the block: the dgrad listed behind it on the compute stream has claimed its CUs
            # before the weight gradients' workgroups arrive, whatever the latency of the cross-queue wake-up (profiles/r04_experiments.md)
            b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_HOLD, n=(1, self.side_delay_us, 0)), None, None))
        self.gemm(b, L.TN, L.EPI_F32, probs)
        # every slab sum and every deferred LayerNorm dgamma / dbeta reduction of the sub-layer in ONE launch (vk_side_tail)
        jobs = [L.TailJob(_addr(dst), None, _addr(src), None, stride, n, 0, ns, 0, 0) for dst, src, stride, ns, n in reduces]
        by_dst = {}
        for a in getattr(self, "_deferred_ln", []):          # LayerNorm parameter gradients of this sub-layer; a LayerNorm shared by both
            by_dst.setdefault(a.dgamma, []).append(a)         # modalities has two sets of partial records: ONE job sums both (no ordering between jobs)
        for group in by_dst.values():
            assert len(group) <= 2 and not (group[0].accumulate & 1) and all(g.accumulate & 1 for g in group[1:]), "unexpected LayerNorm sharing"
            a, b2 = group[0], (group[1] if len(group) > 1 else None)
            jobs.append(L.TailJob(a.dgamma, a.dbeta, a.partial, b2.partial if b2 else None, L.lib.vk_ln_bwd_partial_rows(b2.M) if b2 else 0, a.H, 1,
                                  L.lib.vk_ln_bwd_partial_rows(a.M), 0, 0))
        self._deferred_ln = []
        for i in range(0, len(jobs), L.TAIL_MAX_JOBS):
            chunk = jobs[i:i + L.TAIL_MAX_JOBS]
            arr = self.k((L.TailJob * len(chunk))(*chunk))
            b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_SIDE_TAIL, p=(arr,), n=(len(chunk),)), None, None))
        b.append((L.OP_SIDE_END, self.sub_k % 8, 0, 0, None, None, None))
        self._slab_cursor = 0

    def _gate_words(self):
        """uint64 words of the weight-gradient gates: word 0 = the epoch, bumped by the first launch of every backward pass (the side stream
        is ordered behind that launch with the one event of the pass); word k = the flag the k-th block's dgrad stores the epoch to."""
        if "gate_words" not in self.bufs:
            self.bufs["gate_words"] = torch.zeros(128, dtype=torch.int64, device=self.dev)
            self.bufs["gate_words"][1:] = -1
            self.bufs["gate_err"] = torch.zeros(1, dtype=torch.int32, device=self.dev)
            g = self.bufs["gate_words"]
            self.bwd_pro.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_BUMP, p=(g,)), None, None))
            self.bwd_pro.append((L.OP_SIDE_BEGIN, 0, 0, 0, None, None, None))       # an empty block: its fork event orders the side stream behind the bump
            self.bwd_pro.append((L.OP_SIDE_END, 13, 0, 0, None, None, None))
        return self.bufs["gate_words"]

    def retire_on(self, probs):
        """Hang the pending gate flag on a launch's problem 0: its workgroups release the weight-gradient block as they retire."""
        gf = getattr(self, "_gate_flag", None)
        if gf is not None:
            probs[0].retire_flag, probs[0].retire_stamp = gf
            self._gate_flag = None
        return probs

    def _slab(self, n):
        """fp32 workspace for split-K partials; one arena reused by every sub-layer (launches are stream-ordered)."""
        cap = 48 * 3072 * 768
        ws = self.tmp("wgrad_slabs", (cap,), torch.float32)
        cur = getattr(self, "_slab_cursor", 0)
        cur = _round_up(cur, 4)
        assert cur + n <= cap, "wgrad slab workspace too small"
        self._slab_cursor = cur + n
        return ws[cur:cur + n]

    # ---------------------------------------------------------------- heads + losses
    def _heads(self):
        cfg, B, H, T, Rv, R = self.cfg, self.B, self.H, self.T, self.Rv, self.R
        Hv = self.st[1].H                               # the vision stream's width (config/vilbert_base.json: 1024 against 768)
        f = self.fwd.ops
        st_t, st_v = self.st
        fm = cfg.fusion_method
        has_itm = fm in ("mul", "sum", "text")         # encoders.py:744-747: no ITM head for "none" / "vl-bert_vqa"
        P = cfg.pooler_size if has_itm else 0
        if has_itm and ((fm != "text" and P != cfg.v_pooler_size) or P % 64):
            raise NotImplementedError("pooler sizes must match and be multiples of 64")
        if fm == "vl-bert_vqa":                        # the VQA text pooler exists but feeds nothing in pre-training (its .grad stays None)
            self.unused_params |= {"bert.t_pooler.dense.weight", "bert.t_pooler.dense.bias"}
        V, Vp = cfg.vocab_size, _round_up(cfg.vocab_size, 64)
        targets = [(ix, float(w)) for ix, w in cfg.visual_target_weights.items() if w > 0]
        if not targets:
            raise NotImplementedError("no visual target with a positive weight")
        if {ix for ix, _ in targets} & {"1", "2", "5"} and cfg.add_global_imgfeat is not None:
            raise NotImplementedError("the feature-regression targets compare [B, R, 2048] predictions with the [B, R+1, 2048] input "
                                      "when a global feature is added (losses.py:28,41,108 fail on the shapes)")
        x_t, x_v = self.x
        self.sums = self.buf("loss_sums", (4,), torch.float32)
        self.losses = self.buf("losses", (3,), torch.float32)
        self.gout = self.buf("gout", (3,), torch.float32)
        f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_MEMSET, p=(self.sums,), n=(16, 0)), None, None))
        # ---- ITM: poolers, fusion, classifier
        pt = pv = None
        if has_itm:
            pt = self.buf("pooled_t", (B, P))
            pools = [self.prob(x_t, self.W("bert.t_pooler.dense.weight"), pt, B, P, H, T * H, H, P, bias=self.Pm("bert.t_pooler.dense.bias"))]
            if fm != "text":
                pv = self.buf("pooled_v", (B, P))
                pools.append(self.prob(x_v, self.W("bert.v_pooler.dense.weight"), pv, B, P, Hv, Rv * Hv, Hv, P, bias=self.Pm("bert.v_pooler.dense.bias")))
            self.gemm(f, L.NT, L.EPI_RELU, pools)
        elif fm == "vl-bert_vqa":
            pt = self._vqa_text_pooler(f, x_t)[0]               # BertModel's fourth output; nothing in the pre-training loss reads it
        # ---- masked LM on labelled rows
        n_t, n_v = self.buf("n_t", (1,), torch.int32), self.buf("n_v", (1,), torch.int32)
        rows_t, pos_t = self.buf("rows_t", (st_t.M,), torch.int32), self.buf("pos_t", (st_t.M,), torch.int32)
        g = self.generic(L.FN_SELECT, p=(None, rows_t, pos_t, n_t), n=(st_t.M, 0, T, T, 0))
        self.patch("masked_lm_labels", g, "p", 0)
        f.append((L.OP_GENERIC, 0, 0, 0, g, None, None))
        hx_t = self.buf("lm_hx", (st_t.M, H))
        f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_GATHER, p=(x_t, rows_t, n_t, hx_t), n=(H, st_t.M)), None, None))
        c = "cls.predictions."
        ht, gpt = self.buf("lm_ht", (st_t.M, H)), self.buf("lm_gp", (st_t.M, H))
        self.gemm(f, L.NT, L.EPI_GELU, [self.prob(hx_t, self.W(c + "transform.dense.weight"), ht, st_t.M, H, H, H, H, H, bias=self.Pm(c + "transform.dense.bias"), C2=gpt, dyn=n_t)])
        hn_t = self.buf("lm_hn", (st_t.M, H))
        lm_mean, lm_rstd = self.buf("lm_mean", (st_t.M,), torch.float32), self.buf("lm_rstd", (st_t.M,), torch.float32)
        nodrop = L.dropout_cfg(None, 0, 0.0)
        f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(ht, None, c + "transform.LayerNorm.weight", c + "transform.LayerNorm.bias", hn_t, None, lm_mean, lm_rstd, st_t.M, nodrop, dyn=n_t), None, None))
        logits_t = self.buf("lm_logits", (st_t.M, Vp), torch.float32)
        wword = "bert.embeddings.word_embeddings.weight"
        self.gemm(f, L.NT, L.EPI_F32, [self.prob(hn_t, self.W(wword), logits_t, st_t.M, V, H, H, H, Vp, bias=self.Pm(c + "bias"), dyn=n_t, n_store=Vp)])
        lse_t = self.buf("lm_lse", (st_t.M,), torch.float32)
        xa = self.k(L.XentArgs(_addr(logits_t), None, _addr(pos_t), _addr(n_t), _addr(lse_t), _addr(self.sums[0:1]), V, Vp, st_t.M))
        self.patch("masked_lm_labels", xa, "labels")
        f.append((L.OP_XENT_FWD, 0, 0, 0, xa, None, None))
        # ---- masked regions (kl_1601) on labelled rows: an independent chain of small launches -> side stream, joined before the loss finalisation
        f.append((L.OP_SIDE_BEGIN, 0, 0, 0, None, None, None))
        Mr = B * R
        rows_v, pos_v = self.buf("rows_v", (Mr,), torch.int32), self.buf("pos_v", (Mr,), torch.int32)
        off = 1 if cfg.add_global_imgfeat == "first" else 0
        g = self.generic(L.FN_SELECT, p=(None, rows_v, pos_v, n_v), n=(Mr, 1, R, Rv, off))
        self.patch("image_label", g, "p", 0)
        f.append((L.OP_GENERIC, 0, 0, 0, g, None, None))
        hx_v = self.buf("img_hx", (Mr, Hv))
        f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_GATHER, p=(x_v, rows_v, n_v, hx_v), n=(Hv, Mr)), None, None))
        ci = "cls.imagePredictions."
        hv, gpv = self.buf("img_ht", (Mr, Hv)), self.buf("img_gp", (Mr, Hv))
        self.gemm(f, L.NT, L.EPI_GELU, [self.prob(hx_v, self.W(ci + "transform.dense.weight"), hv, Mr, Hv, Hv, Hv, Hv, Hv, bias=self.Pm(ci + "transform.dense.bias"), C2=gpv, dyn=n_v)])
        if cfg.image_head_ln:
            hn_v = self.buf("img_hn", (Mr, Hv))
            im_mean, im_rstd = self.buf("img_mean", (Mr,), torch.float32), self.buf("img_rstd", (Mr,), torch.float32)
            f.append((L.OP_LN_FWD, 0, 0, 0, self.ln_args(hv, None, ci + "transform.LayerNorm.weight", ci + "transform.LayerNorm.bias", hn_v, None, im_mean, im_rstd, Mr, nodrop, dyn=n_v, H=Hv), None, None))
        else:
            hn_v = hv
        # one decoder + loss per configured visual target (encoders.py:718-737,1079-1087; losses.py); all of them add their WEIGHTED row
        # losses to sums[1], the image loss is sums[1] / max(#masked regions, 1)
        vis = []                                                # (ix, Cn, Cp, forward-args struct, is_kl)
        for ix, w in targets:
            Cn = VIS_TARGET_WIDTH[ix]
            Cp = _round_up(Cn, 64)
            tag = "img" if ix == "0" else "img%s" % ix
            logits_v = self.buf(tag + "_logits", (Mr, Cp), torch.float32)
            wdec = ci + "decoder_dict.%s." % ix
            self.gemm(f, L.NT, L.EPI_F32, [self.prob(hn_v, self.W(wdec + "weight"), logits_v, Mr, Cn, Hv, Hv, Hv, Cp, bias=self.Pm(wdec + "bias"), dyn=n_v, n_store=Cp)])
            lse_v = self.buf(tag + "_lse", (Mr,), torch.float32)
            if ix == "0":
                tsum_v = self.buf("img_tsum", (Mr,), torch.float32)
                la = self.k(L.KlArgs(_addr(logits_v), None, _addr(pos_v), _addr(n_v), _addr(lse_v), _addr(tsum_v), _addr(self.sums[1:2]), w, Cn, Cp, Mr))
                self.patch("image_cls", la, "target")
                f.append((L.OP_KL_FWD, 0, 0, 0, la, None, None))
            else:
                kind = {"1": L.VIS_MSE, "2": L.VIS_NCE, "3": L.VIS_XENT, "4": L.VIS_XENT, "5": L.VIS_HUBER, "6": L.VIS_XENT}[ix]
                la = L.VisLossArgs(_addr(logits_v), None, None, None, _addr(pos_v), _addr(n_v), None, _addr(lse_v), None, _addr(self.sums[1:2]), w, Cn, Cp, Mr, kind, 0)
                self.k(la)
                if kind in (L.VIS_MSE, L.VIS_HUBER, L.VIS_NCE):
                    self.patch("image_feat", la, "target")
                if kind == L.VIS_XENT:
                    self.patch("attr_labels" if ix == "4" else "obj_labels", la, "labels")
                    if ix in ("3", "4"):
                        self.patch("attr_confs" if ix == "4" else "obj_confs", la, "conf")
                if kind == L.VIS_NCE:
                    nneg = L.NCE_ACROSS + L.NCE_INSIDE
                    neg = self.buf("img_nce_neg", (B * R * nneg,), torch.int32)
                    aux = self.buf("img_nce_scores", (Mr, L.NCE_MAX_SAMPLES), torch.float32)
                    la.neg_index, la.aux, la.n_neg = _addr(neg), _addr(aux), nneg
                    self.nce_site = self.site                    # the negatives' counter-based stream (drawn in eval mode too, as the reference does)
                    self.site += 1
                    rng = L.rng_cfg(self.seed.data_ptr(), self.nce_site)
                    f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_NCE_NEG, p=(neg,), n=(B, R), drop=rng), None, None))
                f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_VIS_LOSS_FWD, p=(la,)), None, None))
            vis.append((ix, Cn, Cp, la, tag))
        f.append((L.OP_SIDE_END, 11, 0, 0, None, None, None))
        # ---- ITM head (its dropout site is the last one of the forward pass)
        fuse = {"mul": L.FUSE_MUL, "sum": L.FUSE_SUM, "text": L.FUSE_TEXT}.get(fm)
        pdrop = self.drop(0.1)                                   # the reference's nn.Dropout(0.1) sits in every variant of the heads
        if has_itm:
            pooled = self.buf("pooled", (B, P))
            f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_POOL_FWD, p=(pt, pv, pooled), n=(B, P, 0, fuse), drop=pdrop), None, None))
            itm = self.buf("itm_logits", (B, 64), torch.float32)
            self.gemm(f, L.NT, L.EPI_F32, [self.prob(pooled, self.W("cls.bi_seq_relationship.weight"), itm, B, 2, P, P, P, 64, bias=self.Pm("cls.bi_seq_relationship.bias"), n_store=64)])
            lse_i = self.buf("itm_lse", (B,), torch.float32)
            xi = self.k(L.XentArgs(_addr(itm), None, None, None, _addr(lse_i), _addr(self.sums[2:3]), 2, 64, B))
            self.patch("next_sentence_label", xi, "labels")
            f.append((L.OP_XENT_FWD, 0, 0, 0, xi, None, None))
        f.append((L.OP_WAIT_SIDE, 11, 0, 0, None, None, None))
        f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_LOSS_FINAL, p=(self.sums, n_t, n_v, self.losses), n=(B,), f=(1.0,)), None, None))
        self.taps.update(seq_t=x_t, seq_v=x_v, pooled_t=pt, pooled_v=pv)

        # ================= backward of the heads: produces dX[0], dX[1] (buffer 'a').  Three independent chains of small launches:
        # the masked-LM chain and the ITM chain stay on the caller's stream, the region chain (with its weight gradients) runs on the side
        # stream next to them (event 13, awaited before the encoder's backward reads dX[1]), and every weight gradient of the two main
        # chains follows in a second side block (event 12, awaited before the embedding backward adds to the word-embedding gradient
        # that the tied LM decoder's weight gradient initialises).  Each chain has its own temporaries.
        b, wg = [], []
        dxh = [self._dx(m, self.level[m] % 2) for m in range(2)]
        for m in range(2):
            b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_MEMSET, p=(dxh[m],), n=(dxh[m].numel() * 2, 0)), None, None))
        # ---- region chain (side stream)
        b.append((L.OP_SIDE_BEGIN, 0, 0, 0, None, None, None))
        dhn_v = self.tmp("head_v_d1", (Mr, Hv))
        for j, (ix, Cn, Cp, la, tag) in enumerate(vis):
            dlog_v = self.buf(tag + "_dlogits", (Mr, Cp))
            if ix == "0":
                b.append((L.OP_KL_BWD, Cp, 0, 0, la, dlog_v, self.gout[1:2]))
            else:
                b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_VIS_LOSS_BWD, p=(la, dlog_v, self.gout[1:2]), n=(Cp,)), None, None))
            wdec = ci + "decoder_dict.%s." % ix
            if j == 0:
                self.gemm(b, L.NN, L.EPI_BF16, [self.prob(dlog_v, self.W(wdec + "weight"), dhn_v, Mr, Hv, Cn, Cp, Hv, Hv, dyn=n_v)])
            else:                                               # the decoders share the transformed hidden state: their input gradients add up
                self.gemm(b, L.NN, L.EPI_ADDR, [self.prob(dlog_v, self.W(wdec + "weight"), dhn_v, Mr, Hv, Cn, Cp, Hv, Hv, dyn=n_v, R=dhn_v, ldr=Hv)])
            self.gemm(b, L.TN, L.EPI_F32, [self.prob(dlog_v, hn_v, self.G(wdec + "weight"), Cn, Hv, Mr, Cp, Hv, Hv, bias_grad=self.G(wdec + "bias"), dyn=n_v)])
        if cfg.image_head_ln:
            dhv = self.tmp("head_v_d2", (Mr, Hv))
            b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(dhn_v, hv, im_mean, im_rstd, ci + "transform.LayerNorm.weight", ci + "transform.LayerNorm.bias", dhv, None, Mr, nodrop, dyn=n_v, own_partial=True, H=Hv), None, None))
        else:
            dhv = dhn_v
        du_v = self.tmp("head_v_d3", (Mr, Hv))
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_MUL, p=(dhv, gpv, du_v, n_v), n=(Mr * Hv, Hv)), None, None))
        dhx_v = self.tmp("head_v_d2" if not cfg.image_head_ln else "head_v_d1", (Mr, Hv))
        self.gemm(b, L.NN, L.EPI_BF16, [self.prob(du_v, self.W(ci + "transform.dense.weight"), dhx_v, Mr, Hv, Hv, Hv, Hv, Hv, dyn=n_v)])
        self.gemm(b, L.TN, L.EPI_F32, [self.prob(du_v, hx_v, self.G(ci + "transform.dense.weight"), Hv, Hv, Mr, Hv, Hv, Hv, bias_grad=self.G(ci + "transform.dense.bias"), dyn=n_v)])
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_SCATTER_ADD, p=(dhx_v, rows_v, n_v, dxh[1]), n=(Hv, Mr)), None, None))
        b.append((L.OP_SIDE_END, 13, 0, 0, None, None, None))
        # ---- masked-LM chain
        dlog_t = self.buf("lm_dlogits", (st_t.M, Vp))
        b.append((L.OP_XENT_BWD, Vp, 0, 0, xa, dlog_t, self.gout[0:1]))
        dhn_t = self.tmp("head_d1", (st_t.M, H))
        # d(hidden) = dlogits[n_t, V] . E[V, H]: few rows, very long contraction -> split K over the vocabulary into
        # chunks written as fp32 slabs by one grouped launch, then summed (and rounded to bf16) in one pass
        nsplit = max(1, min(16, V // 1920))
        if nsplit == 1:
            self.gemm(b, L.NN, L.EPI_BF16, [self.prob(dlog_t, self.W(wword), dhn_t, st_t.M, H, V, Vp, H, H, dyn=n_t)])
        else:
            kc = _round_up(-(-V // nsplit), 64)
            stride = st_t.M * H
            slabs = self.buf("lm_dgrad_slabs", (nsplit * stride,), torch.float32)
            probs, k0, wv = [], 0, self.W(wword)
            while k0 < V:
                kk = min(kc, V - k0)
                i = len(probs)
                probs.append(self.prob(dlog_t[:, k0:], wv[k0:], slabs[i * stride:], st_t.M, H, kk, Vp, H, H, dyn=n_t))
                k0 += kc
            self.gemm(b, L.NN, L.EPI_F32, probs)
            b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_SUM_SLABS_BF16, p=(dhn_t, slabs, n_t), n=(stride, len(probs), stride, H)), None, None))
        self.gemm(wg, L.TN, L.EPI_F32, [self.prob(dlog_t, hn_t, self.G(wword), V, H, st_t.M, Vp, H, H, bias_grad=self.G(c + "bias"), dyn=n_t)])
        dht = self.tmp("head_d2", (st_t.M, H))
        b.append((L.OP_LN_BWD, 0, 0, 0, self.ln_bwd_args(dhn_t, ht, lm_mean, lm_rstd, c + "transform.LayerNorm.weight", c + "transform.LayerNorm.bias", dht, None, st_t.M, nodrop, dyn=n_t), None, None))
        du_t = self.tmp("head_d3", (st_t.M, H))
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_MUL, p=(dht, gpt, du_t, n_t), n=(st_t.M * H, H)), None, None))
        dhx_t = self.tmp("head_d1", (st_t.M, H))
        self.gemm(b, L.NN, L.EPI_BF16, [self.prob(du_t, self.W(c + "transform.dense.weight"), dhx_t, st_t.M, H, H, H, H, H, dyn=n_t)])
        self.gemm(wg, L.TN, L.EPI_F32, [self.prob(du_t, hx_t, self.G(c + "transform.dense.weight"), H, H, st_t.M, H, H, H, bias_grad=self.G(c + "transform.dense.bias"), dyn=n_t)])
        b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_SCATTER_ADD, p=(dhx_t, rows_t, n_t, dxh[0]), n=(H, st_t.M)), None, None))
        # ---- ITM chain
        poolers = []
        if has_itm:
            dlog_i = self.buf("itm_dlogits", (B, 64))
            b.append((L.OP_XENT_BWD, 64, 0, 0, xi, dlog_i, self.gout[2:3]))
            dpooled = self.buf("d_pooled", (B, P))
            wi = "cls.bi_seq_relationship."
            self.gemm(b, L.NN, L.EPI_BF16, [self.prob(dlog_i, self.W(wi + "weight"), dpooled, B, P, 2, 64, P, P)])
            self.gemm(wg, L.TN, L.EPI_F32, [self.prob(dlog_i, pooled, self.G(wi + "weight"), 2, P, B, 64, P, P, bias_grad=self.G(wi + "bias"))])
            dyt = self.buf("d_pool_t", (B, P))
            dyv = self.buf("d_pool_v", (B, P)) if pv is not None else None
            b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_POOL_BWD, p=(dpooled, pt, pv, dyt, dyv), n=(B, P, P, fuse), drop=pdrop), None, None))
            poolers = [(0, dyt, x_t, T, "bert.t_pooler.dense.")] + ([(1, dyv, x_v, Rv, "bert.v_pooler.dense.")] if pv is not None else [])
        for m, dy_, xm, Lm, pre in poolers:
            # rows b * L: the first token of every sample -- disjoint from the labelled rows the region chain scatters into
            Hm = self.st[m].H
            self.gemm(b, L.NN, L.EPI_ADDR, [self.prob(dy_, self.W(pre + "weight"), dxh[m], B, Hm, P, P, Hm, Lm * Hm, R=dxh[m], ldr=Lm * Hm)])
            self.gemm(wg, L.TN, L.EPI_F32, [self.prob(dy_, xm, self.G(pre + "weight"), P, Hm, B, P, Lm * Hm, Hm, bias_grad=self.G(pre + "bias"))])
        # ---- the main chains' weight gradients, off the critical path
        b.append((L.OP_SIDE_BEGIN, 0, 0, 0, None, None, None))
        b += wg
        b.append((L.OP_SIDE_END, 12, 0, 0, None, None, None))
        b.append((L.OP_WAIT_SIDE, 13, 0, 0, None, None, None))            # dX[1] is complete
        self._head_wgrad_event = 12
        return b

    def _vqa_text_pooler(self, f, x_t):
        """VLBertTextPooler (volta/encoders.py:610-623): ReLU(dense(hidden state of the token two places before the caption's end)).
        -> (pooled [B, P], gathered rows [B, H], row index [B], count)."""
        cfg, B, H, T = self.cfg, self.B, self.H, self.T
        P = cfg.pooler_size
        if P % 64:
            raise NotImplementedError("pooler size must be a multiple of 64")
        rows, cnt = self.buf("vqa_rows", (B,), torch.int32), self.buf("vqa_cnt", (1,), torch.int32)
        g = self.generic(L.FN_TEXT_END_ROWS, p=(None, rows, cnt), n=(B, T))
        self.patch("input_ids", g, "p", 0)
        f.append((L.OP_GENERIC, 0, 0, 0, g, None, None))
        xg = self.buf("vqa_x", (B, H))
        f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_GATHER, p=(x_t, rows, cnt, xg), n=(H, B)), None, None))
        pt = self.buf("pooled_t", (B, P))
        self.gemm(f, L.NT, L.EPI_RELU, [self.prob(xg, self.W("bert.t_pooler.dense.weight"), pt, B, P, H, H, H, P, bias=self.Pm("bert.t_pooler.dense.bias"))])
        return pt, xg, rows, cnt

    def _heads_tasks(self):
        """BertForVLTasks behind the encoder (volta/encoders.py:1117-1206): poolers (:1004-1011; none / text-only / VLBertTextPooler by fusion
        method, :936-947), the fusion + dropout of the pooled vectors (:1184-1195) and the task's classifier (:1128-1149: SimpleClassifier =
        Linear -> GELU -> LayerNorm -> Linear (:787-814), plain Linear heads, the one- and two-layer region-logit heads on dropout(seq_v)),
        forward and backward.  The prediction leaves the engine as fp32 logits (`self.pred`, leading dimension padded to 64); the backward is
        seeded by d(loss)/d(logits), which the host writes (bf16) into `self.d_pred`.  Without a task (BertModel.forward / encode()) only the
        poolers are built."""
        cfg, B, H, T, Rv = self.cfg, self.B, self.H, self.T, self.Rv
        f = self.fwd.ops
        fm = cfg.fusion_method
        P = cfg.pooler_size
        if fm != "none" and ((fm in ("mul", "sum") and P != cfg.v_pooler_size) or P % 64):
            raise NotImplementedError("pooler sizes must match and be multiples of 64")
        x_t, x_v = self.x
        st_v = self.st[1]
        Hv = st_v.H
        pt = pv = None
        vqa = None
        if fm == "vl-bert_vqa":
            pt, xg, vqa_rows, vqa_cnt = vqa = self._vqa_text_pooler(f, x_t)
        elif fm != "none":
            pt = self.buf("pooled_t", (B, P))
            pools = [self.prob(x_t, self.W("bert.t_pooler.dense.weight"), pt, B, P, H, T * H, H, P, bias=self.Pm("bert.t_pooler.dense.bias"))]
            if fm != "text":
                pv = self.buf("pooled_v", (B, P))
                pools.append(self.prob(x_v, self.W("bert.v_pooler.dense.weight"), pv, B, P, Hv, Rv * Hv, Hv, P, bias=self.Pm("bert.v_pooler.dense.bias")))
            self.gemm(f, L.NT, L.EPI_RELU, pools)
        self.taps.update(seq_t=x_t, seq_v=x_v, pooled_t=pt, pooled_v=pv)
        # The word-embedding gradient is accumulated with atomics by the embedding backward; in the pre-training model the LM
        # decoder's weight gradient (same tied tensor) is what initialises it, here nothing else writes it: zero it per step.
        for nm in self.arena.params:
            if nm.endswith("embeddings.word_embeddings.weight"):
                self._zero_grad(self.G(nm))
        b = []
        dxh = [self._dx(m, self.level[m] % 2) for m in range(2)]
        for m in range(2):                                 # the sequence outputs feed nothing but the task head
            b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_MEMSET, p=(dxh[m],), n=(dxh[m].numel() * 2, 0)), None, None))
        self.pred = self.d_pred = None
        heads_of_other_tasks = [nm for nm in self.arena.params if nm.startswith("clfs_dict.")]
        dpool = [None, None]                               # gradients at the poolers' pre-activation outputs
        if self.task is not None:
            task_id, tcfg = self.task
            typ = tcfg["type"]
            pre = "clfs_dict.%s." % task_id
            heads_of_other_tasks = [nm for nm in heads_of_other_tasks if not nm.startswith(pre)]
            nodrop = L.dropout_cfg(None, 0, 0.0)

            def linear_out(x, rows, K, wname, C):
                """logits = x W^T + b as fp32 [rows, 64 k]; returns (logits, dlogits bf16)."""
                Cp = _round_up(C, 64)
                out = self.buf("task_logits", (rows, Cp), torch.float32)
                self.gemm(f, L.NT, L.EPI_F32, [self.prob(x, self.W(wname + "weight"), out, rows, C, K, K, K, Cp, bias=self.Pm(wname + "bias"), n_store=Cp)])
                return out, self.buf("task_dlogits", (rows, Cp), zero=True), Cp

            def linear_bwd(dlog, Cp, x, rows, K, wname, C, dx):
                self.gemm(b, L.NN, L.EPI_BF16, [self.prob(dlog, self.W(wname + "weight"), dx, rows, K, C, Cp, K, K)])
                self.gemm(b, L.TN, L.EPI_F32, [self.prob(dlog, x, self.G(wname + "weight"), C, K, rows, Cp, K, K, bias_grad=self.G(wname + "bias"))])

            if typ.startswith("V-logit"):
                Mv = st_v.M
                d0 = self.drop(self.task_dropout)            # BertForVLTasks.dropout on the region states (:1198)
                xd = self.buf("task_xd", (Mv, Hv))
                f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_ADD_DROPOUT, p=(x_v, None, xd), n=(Mv, Hv, 0), f=(1.0,), drop=d0), None, None))
                if tcfg.get("num_clf_layers", 1) == 2:       # Linear -> GELU -> Dropout -> Linear (:1138-1144)
                    h, gp = self.buf("task_h", (Mv, Hv)), self.buf("task_gp", (Mv, Hv))
                    self.gemm(f, L.NT, L.EPI_GELU, [self.prob(xd, self.W(pre + "0.weight"), h, Mv, Hv, Hv, Hv, Hv, Hv, bias=self.Pm(pre + "0.bias"), C2=gp)])
                    d1 = self.drop(cfg.v_attention_probs_dropout_prob)
                    hd = self.buf("task_hd", (Mv, Hv))
                    f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_ADD_DROPOUT, p=(h, None, hd), n=(Mv, Hv, 0), f=(1.0,), drop=d1), None, None))
                    self.pred, self.d_pred, Cp = linear_out(hd, Mv, Hv, pre + "3.", 1)
                    dhd, dh, du = self.buf("task_dhd", (Mv, Hv)), self.buf("task_dh", (Mv, Hv)), self.buf("task_du", (Mv, Hv))
                    linear_bwd(self.d_pred, Cp, hd, Mv, Hv, pre + "3.", 1, dhd)
                    b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_ADD_DROPOUT, p=(dhd, None, dh), n=(Mv, Hv, 1), f=(1.0,), drop=d1), None, None))
                    b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_MUL, p=(dh, gp, du, None), n=(Mv * Hv, Hv)), None, None))
                    dxd = self.buf("task_dxd", (Mv, Hv))
                    linear_bwd(du, Hv, xd, Mv, Hv, pre + "0.", Hv, dxd)
                else:
                    self.pred, self.d_pred, Cp = linear_out(xd, Mv, Hv, pre, 1)
                    dxd = self.buf("task_dxd", (Mv, Hv))
                    linear_bwd(self.d_pred, Cp, xd, Mv, Hv, pre, 1, dxd)
                b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_ADD_DROPOUT, p=(dxd, None, dxh[1]), n=(Mv, Hv, 1), f=(1.0,), drop=d0), None, None))
                self.pred_shape = (B, Rv, 1)
            else:
                if fm == "none":
                    raise ValueError("task type %r needs a pooled output; fusion method 'none' has none (encoders.py:1192-1193)" % typ)
                fuse = {"mul": L.FUSE_MUL, "sum": L.FUSE_SUM, "text": L.FUSE_TEXT, "vl-bert_vqa": L.FUSE_TEXT}[fm]
                d0 = self.drop(self.task_dropout)            # BertForVLTasks.dropout on the fused pooled vector (:1184-1191)
                pooled = self.buf("pooled", (B, P))
                f.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_POOL_FWD, p=(pt, pv, pooled), n=(B, P, 0, fuse), drop=d0), None, None))
                rows, K0 = (B // 2, 2 * P) if typ == "VL-binary-classifier" else (B, P)      # NLVR2 pairs: view(-1, 2 P) (:1202)
                if typ == "VL-binary-classifier" and B % 2:
                    raise ValueError("VL-binary-classifier pools pairs of samples: the batch size must be even")
                dpooled = self.buf("d_pooled", (B, P))
                if typ in ("VL-classifier", "VL-classifier-GQA", "VL-binary-classifier"):
                    C = 2 if typ == "VL-binary-classifier" else int(tcfg["num_labels"])
                    Hc = cfg.clf_hidden_size
                    if Hc % 64 or Hc > 2048:
                        raise NotImplementedError("clf_hidden_size must be a multiple of 64, <= 2048")
                    hc, gpc = self.buf("task_h", (rows, Hc)), self.buf("task_gp", (rows, Hc))
                    w0, lnn, w3 = pre + "logit_fc.0.", pre + "logit_fc.2.", pre + "logit_fc.3."
                    self.gemm(f, L.NT, L.EPI_GELU, [self.prob(pooled, self.W(w0 + "weight"), hc, rows, Hc, K0, K0, K0, Hc, bias=self.Pm(w0 + "bias"), C2=gpc)])
                    hn = self.buf("task_hn", (rows, Hc))
                    mean, rstd = self.buf("task_mean", (rows,), torch.float32), self.buf("task_rstd", (rows,), torch.float32)
                    la = L.LnArgs(_addr(hc), None, None, _addr(self.Pm(lnn + "weight")), _addr(self.Pm(lnn + "bias")), _addr(hn), None, _addr(mean), _addr(rstd), None,
                                  rows, Hc, rows, 0, 1.0, nodrop, _mk_segs(nodrop, None))
                    f.append((L.OP_LN_FWD, 0, 0, 0, self.k(la), None, None))
                    self.pred, self.d_pred, Cp = linear_out(hn, rows, Hc, w3, C)
                    dhn, dhc, du = self.buf("task_dhn", (rows, Hc)), self.buf("task_dhc", (rows, Hc)), self.buf("task_du", (rows, Hc))
                    linear_bwd(self.d_pred, Cp, hn, rows, Hc, w3, C, dhn)
                    part = self.buf("task_ln_partial", (L.lib.vk_ln_bwd_partial_rows(rows) * 2 * Hc,), torch.float32)
                    lb = L.LnBwdArgs(_addr(dhn), _addr(hc), _addr(mean), _addr(rstd), _addr(self.Pm(lnn + "weight")), _addr(dhc), None, _addr(part),
                                     _addr(self.G(lnn + "weight")), _addr(self.G(lnn + "bias")), None, rows, Hc, rows, 0, 1.0, 0, nodrop, _mk_segs(nodrop, None))
                    b.append((L.OP_LN_BWD, 0, 0, 0, self.k(lb), None, None))
                    b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_MUL, p=(dhc, gpc, du, None), n=(rows * Hc, Hc)), None, None))
                    linear_bwd(du, Hc, pooled, rows, K0, w0, Hc, dpooled)
                else:                                         # VL-tri-classifier (3 classes), VL-logit (1 score): one Linear (:1134-1137)
                    C = 3 if typ == "VL-tri-classifier" else 1
                    self.pred, self.d_pred, Cp = linear_out(pooled, rows, K0, pre, C)
                    linear_bwd(self.d_pred, Cp, pooled, rows, K0, pre, C, dpooled)
                self.pred_shape = (rows, C)
                dpool[0] = self.buf("d_pool_t", (B, P))
                dpool[1] = self.buf("d_pool_v", (B, P)) if pv is not None else None
                b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_POOL_BWD, p=(dpooled, pt, pv, dpool[0], dpool[1]), n=(B, P, P, fuse), drop=d0), None, None))
        self.unused_params |= set(heads_of_other_tasks)      # the classifiers of the other tasks (and, for region-logit tasks, the poolers) get no gradient
        for m, (xm, Lm, pre_p, py) in enumerate(((x_t, T, "bert.t_pooler.dense.", pt), (x_v, Rv, "bert.v_pooler.dense.", pv))):
            if py is None:
                continue
            if dpool[m] is None:
                self.unused_params |= {pre_p + "weight", pre_p + "bias"}
                continue
            dy_ = dpool[m]
            if vqa is not None:                            # the pooled token differs per caption: gathered rows in, scatter-add out
                dxg = self.buf("vqa_dx", (B, H))
                self.gemm(b, L.NN, L.EPI_BF16, [self.prob(dy_, self.W(pre_p + "weight"), dxg, B, H, P, P, H, H)])
                b.append((L.OP_GENERIC, 0, 0, 0, self.generic(L.FN_SCATTER_ADD, p=(dxg, vqa_rows, vqa_cnt, dxh[0]), n=(H, B)), None, None))
                self.gemm(b, L.TN, L.EPI_F32, [self.prob(dy_, xg, self.G(pre_p + "weight"), P, H, B, P, H, H, bias_grad=self.G(pre_p + "bias"))])
            else:
                Hm = self.st[m].H
                self.gemm(b, L.NN, L.EPI_ADDR, [self.prob(dy_, self.W(pre_p + "weight"), dxh[m], B, Hm, P, P, Hm, Lm * Hm, R=dxh[m], ldr=Lm * Hm)])
                self.gemm(b, L.TN, L.EPI_F32, [self.prob(dy_, xm, self.G(pre_p + "weight"), P, Hm, B, P, Lm * Hm, Hm, bias_grad=self.G(pre_p + "bias"))])
        return b

    # ---------------------------------------------------------------- run
    def attention_maps(self):
        """(all_attention_mask_t, all_attention_mask_v) of BertEncoder.forward (volta/encoders.py:858-886) under config.visualization: per
        attention sub-layer and modality {"intra_attn", "inter_attn", "queries", "keys"} -- probabilities [B, heads, Lq, Lk] after dropout,
        query / key layers [B, heads, L, head size] (encoders.py:342-356); None where the modality takes no part.  Launches of a sub-layer
        whose streams differ in head geometry are listed one after the other."""
        out = ([], [])
        for info in self.attn_map_info:
            nh, dh = info["nh"], info["dh"]
            for m in range(2):
                if m not in info["qkv"]:
                    out[m].append({"intra_attn": None, "inter_attn": None, "queries": None, "keys": None})
                    continue
                qkv, Ha = info["qkv"][m], info["Ha"][m]
                Lm = self.st[m].L
                heads = lambda t: t.view(self.B, Lm, nh, dh).transpose(1, 2).float()
                p = info["probs"]
                out[m].append({"intra_attn": p[(m, m)].clone() if (m, m) in p else None,
                               "inter_attn": p[(m, 1 - m)].clone() if (m, 1 - m) in p else None,
                               "queries": heads(qkv[:, :Ha]), "keys": heads(qkv[:, Ha:2 * Ha])})
        return out

    def fwd_segments(self, bounds):
        """Cut the forward list for an optimizer that is still updating the arena in `bounds` = [chunk index where range r ends] (ranges in
        arena order = forward order): [(ranges that must be complete, op start, op end)].  The embeddings need what lies before the first
        encoder sub-layer, sub-layer n its own slots, the heads (poolers, cls.*, the tied decoder) everything."""
        arena = self.arena
        key = tuple(bounds)
        cache = self.__dict__.setdefault("_fwd_segments", {})
        if key in cache:
            return cache[key]

        def need(prefixes):          # number of leading ranges covering every parameter with one of these prefixes
            hi = max((arena.offset[nm] + int(torch.tensor(arena.shape[nm]).prod()) for nm in arena.params if nm.startswith(tuple(prefixes))), default=0)
            hi_chunk = -(-hi // CHUNK)
            return next((i + 1 for i, b in enumerate(bounds) if b >= hi_chunk), len(bounds))

        nemb = len(self.stage_prefix) - len(self.fwd_sub_start)
        cuts = [(need(self.stage_prefix[0] if nemb else ["bert.embeddings."]), 0)]
        for k, start in enumerate(self.fwd_sub_start):
            cuts.append((need(self.stage_prefix[nemb + k]), start))
        cuts.append((len(bounds), self.fwd_heads_start))
        segs, cur_need = [], 0
        for i, (nd, start) in enumerate(cuts):
            end = cuts[i + 1][1] if i + 1 < len(cuts) else len(self.fwd.ops)
            nd = max(nd, cur_need)                       # waits are cumulative
            if segs and nd == cur_need:
                segs[-1] = (nd, segs[-1][1], end)        # nothing new to wait for: extend the previous segment
            else:
                segs.append((nd, start, end))
            cur_need = nd
        cache[key] = segs
        return segs

    def run_forward(self):
        """The forward list; when a pipelined optimizer (AdamW(overlap_with_forward=True)) is still walking the arena on its own stream,
        every segment first waits for the ranges whose weights it reads."""
        pend = getattr(self.arena, "opt_pending", None)
        if not pend:
            self.fwd.run()
            return
        bounds, events = pend
        cur = torch.cuda.current_stream()
        waited = 0
        for nd, start, end in self.fwd_segments(bounds):
            for i in range(waited, nd):
                cur.wait_event(events[i])
            waited = max(waited, nd)
            self.fwd.run(start, end)
        self.arena.opt_pending = None

    def bind_inputs(self, tensors):
        """Patch the per-step input pointers into the few ops that read user tensors."""
        self._cur_inputs = tensors
        for name, sites in self.inputs.items():
            t = tensors[name]
            addr = t.data_ptr()
            for struct, field, index in sites:
                if index is None:
                    setattr(struct, field, addr)
                else:
                    getattr(struct, field)[index] = addr

    def prepare_step(self, seed):
        if self.train or getattr(self, "nce_site", None) is not None:      # nce_2048 draws its negatives in eval mode too
            check(L.lib.vk_set_seed(ptr(self.seed), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), L.stream_ptr()))


def _mk_segs(drop, segs):
    arr = (L.DropRows * 2)()
    if segs is None:
        segs = [(drop.site, 0, 0, 0), (drop.site + 1, 0, 0, 0)]
    for i, sg in enumerate(segs):
        arr[i] = L.DropRows(*sg)
    return arr
