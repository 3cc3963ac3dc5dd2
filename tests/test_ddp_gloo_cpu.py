"""world_size-2 gloo run of the data-parallel bucket reducer (volta_amd/parallel.py) on CPU tensors: after the
bucketed all-reduce every rank holds the mean gradient, whatever the bucket cut -- the closed-form check of apex's
ddp_race_condition_test.py:46-65 applied to the range-based reducer."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, cap, mode="allreduce", wire="fp32", pad_to=1, average=True, predivide=1.0):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from volta_amd.parallel import BucketReducer, plan_buckets
    spans = {"emb": (0, 1000), "l0.w": (1024, 4096), "l0.b": (5120, 64), "l1.w": (6144, 4096), "head": (10240, 300)}
    ready = {"head": 0, "l1.w": 1, "l0.w": 2, "l0.b": 2, "emb": 3}
    flat = torch.zeros(10240 + 1024)
    for it in range(3):
        for name, (o, n) in spans.items():
            flat[o:o + n] = (rank + 1) * (it + 1) * (1 + o % 7)        # closed form: mean = (world+1)/2 * ...
        red = BucketReducer(flat, mode=mode, wire=wire, gradient_average=average, gradient_predivide_factor=predivide)
        plan = plan_buckets(spans, ready, 4, cap, pad_to=pad_to, total=flat.numel())
        covered = sorted(r for _, rs in plan for r in rs)
        assert all(a[1] <= b[0] for a, b in zip(covered, covered[1:])), covered        # every element reduced at most once
        if pad_to > 1 and cap >= 1 << 30:
            assert len(covered) == 1, covered        # padded slots coalesce: the whole arena is ONE collective
        for stage, ranges in plan:
            red.reduce(ranges)
        red.finish()
        assert red.bytes_on_wire > 0
        for name, (o, n) in spans.items():
            want = (world + 1) / 2.0 * (it + 1) * (1 + o % 7) * (1.0 if average else world / predivide)      # apex: the sum of g / predivide when not averaging
            assert torch.allclose(flat[o:o + n], torch.full((n,), want)), (name, it, float(flat[o]), want)
    # parameters broadcast from rank 0 as one flat buffer
    params = torch.full((100,), float(rank))
    dist.broadcast(params, 0)
    assert float(params.sum()) == 0.0
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bucket_reducer_world2_gloo():
    for cap in (1, 4 * 5000, 1 << 30):
        mp.spawn(_worker, args=(2, _free_port(), cap), nprocs=2, join=True)


def test_bucket_reducer_modes_world2_gloo():
    """reduce-scatter + all-gather, the bf16 wire (fp32 sum) and slot-padded buckets give the same closed-form means."""
    for mode, wire in (("rs_ag", "fp32"), ("allreduce", "bf16"), ("rs_ag", "bf16")):
        for cap, pad_to in ((1, 1), (4 * 5000, 1024), (1 << 30, 1024)):
            mp.spawn(_worker, args=(2, _free_port(), cap, mode, wire, pad_to), nprocs=2, join=True)


def test_bucket_reducer_apex_scaling_options_world2_gloo():
    """gradient_average=False (sum of the pre-divided gradients) and gradient_predivide_factor (same mean, scaling split around
    the collective), apex/apex/parallel/distributed.py:445-454, under every mode."""
    for mode, wire in (("allreduce", "fp32"), ("rs_ag", "fp32"), ("allreduce", "bf16")):
        for average, predivide in ((False, 1.0), (True, 2.0), (False, 2.0)):
            mp.spawn(_worker, args=(2, _free_port(), 4 * 5000, mode, wire, 1024, average, predivide), nprocs=2, join=True)


def test_ddp_constructor_options():
    """The wrapper's constructor names every apex argument: unsupported ones raise instead of being swallowed."""
    import inspect
    import pytest
    from volta_amd.parallel import DistributedDataParallel
    sig = inspect.signature(DistributedDataParallel.__init__)
    for name in ("message_size", "delay_allreduce", "shared_param", "allreduce_trigger_params", "retain_allreduce_buffers", "allreduce_always_fp32",
                 "num_allreduce_streams", "allreduce_communicators", "gradient_average", "gradient_predivide_factor", "gradient_average_split_factor", "prof"):
        assert name in sig.parameters, name
    assert not any(p.kind == inspect.Parameter.VAR_KEYWORD for p in sig.parameters.values()), "no **kwargs: an unknown option must be a TypeError"
    with pytest.raises(ValueError):
        DistributedDataParallel(None, shared_param=True)
    for kw in (dict(allreduce_trigger_params=[]), dict(retain_allreduce_buffers=True), dict(num_allreduce_streams=2), dict(prof=True),
               dict(gradient_average_split_factor=2.0), dict(allreduce_communicators=([], []))):
        with pytest.raises(NotImplementedError):
            DistributedDataParallel(None, **kw)


def _worker_zero1(rank, world, port, cap):
    """mode "zero1": after the reduce-scatter every rank owns whole-slot shards that tile the arena exactly once (plus replicated
    remainders); a stand-in optimizer steps only what it owns, the gather makes the parameters whole and identical on every rank."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from volta_amd.parallel import BucketReducer, plan_buckets, SLOT
    # slot-aligned spans as the engine's arena has them: 7, 1, 4, 2 and 1 slots
    spans = {"emb": (0, 7 * SLOT - 10), "l0.w": (7 * SLOT, SLOT), "l0.b": (8 * SLOT, 4 * SLOT - 3), "l1.w": (12 * SLOT, 2 * SLOT), "head": (14 * SLOT, 300)}
    ready = {"head": 0, "l1.w": 1, "l0.w": 2, "l0.b": 2, "emb": 3}
    total = 15 * SLOT
    flat = torch.zeros(total)
    params = torch.arange(total, dtype=torch.float32) * 1e-3
    red = BucketReducer(flat, mode="zero1")
    for it in range(3):
        for name, (o, n) in spans.items():
            flat[o:o + n] = (rank + 1) * (it + 1) * (1 + o % 7)
        red.begin_step()
        for stage, ranges in plan_buckets(spans, ready, 4, cap, pad_to=SLOT, total=total):
            red.reduce(ranges)
        red.finish()
        owned = red.owned()
        assert all(lo % SLOT == 0 and hi % SLOT == 0 for lo, hi in owned), owned
        mask = torch.zeros(total)
        for lo, hi in owned:
            mask[lo:hi] += 1
        cover = mask.clone()
        dist.all_reduce(cover)
        rep = torch.zeros(total)
        for lo, hi in red.replicated:
            rep[lo:hi] = 1
        assert bool(((cover == 1) | ((cover == world) & (rep == 1))).all()), "shards tile the arena exactly once; remainders are replicated"
        for lo, hi in owned:                                  # what this rank owns is the mean
            for name, (o, n) in spans.items():
                a, b = max(lo, o), min(hi, o + n)
                if a < b:
                    want = (world + 1) / 2.0 * (it + 1) * (1 + o % 7)
                    assert torch.allclose(flat[a:b], torch.full((b - a,), want)), (name, it)
        for lo, hi in owned:                                  # the stand-in optimizer: only owned elements move
            params[lo:hi] -= 0.5 * flat[lo:hi]
        red.gather(params)
        ref = [torch.empty_like(params) for _ in range(world)]
        dist.all_gather(ref, params)
        assert all(torch.equal(r, ref[0]) for r in ref), "replicas identical after the gather"
    want = torch.arange(total, dtype=torch.float32) * 1e-3
    for name, (o, n) in spans.items():
        want[o:o + n] -= 0.5 * (world + 1) / 2.0 * (1 + 2 + 3) * (1 + o % 7)
    assert torch.allclose(params, want, atol=1e-4), float((params - want).abs().max())
    assert red.bytes_on_wire > 0
    dist.destroy_process_group()


def test_zero1_shards_world2_and_world3_gloo():
    for world in (2, 3):
        for cap in (1, 4 * 3000, 1 << 30):
            mp.spawn(_worker_zero1, args=(world, _free_port(), cap), nprocs=world, join=True)


class _StubArena:
    """What AdamW's checkpoint code touches of a ParamArena (no GPU): one parameter per slot range."""

    def __init__(self, params, slot):
        self.params = {"p%d" % i: p for i, p in enumerate(params)}
        self.offset = {"p%d" % i: i * slot for i in range(len(params))}
        self.shape = {"p%d" % i: tuple(p.shape) for i, p in enumerate(params)}
        self.opt_pending = None

    def sync_optimizer(self):
        pass


def _worker_zero1_checkpoint(rank, world, port):
    """The reference saves the optimizer on ONE rank (`if default_gpu:` around optimizer.state_dict(), volta/train_utils.py:295-316):
    state_dict() must not enter a collective.  Under "zero1" every rank calls consolidate_state_dict() first; without it state_dict()
    raises (on whichever rank calls it) instead of hanging or saving stale foreign shards."""
    import types
    import pytest
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from volta_amd.parallel import BucketReducer, plan_buckets, SLOT
    from volta_amd.optimization import AdamW
    nslots = 6
    total = nslots * SLOT
    params = [torch.nn.Parameter(torch.zeros(SLOT)) for _ in range(nslots)]
    opt = AdamW(params, lr=1e-3)
    flat = torch.ones(total)
    red = BucketReducer(flat, mode="zero1")
    red.begin_step()
    spans = {"p%d" % i: (i * SLOT, SLOT) for i in range(nslots)}
    for stage, ranges in plan_buckets(spans, {n: 0 for n in spans}, 1, 1 << 30, pad_to=SLOT, total=total):
        red.reduce(ranges)
    red.finish()
    model = types.SimpleNamespace(_ddp=types.SimpleNamespace(reducer=red))
    m, v = torch.zeros(total), torch.zeros(total)
    for lo, hi in red.owned():                      # the sharded step touched this rank's shards only
        m[lo:hi] = rank + 1.0
        v[lo:hi] = 10.0 * (rank + 1)
    opt._fused = dict(model=model, arena=_StubArena(params, SLOT), m=m, v=v, step=4, zero1_layout=tuple(red.sharded))
    with pytest.raises(RuntimeError, match="consolidate_state_dict"):
        opt.state_dict()                            # every rank, no collective entered: nobody hangs
    opt.consolidate_state_dict()                    # collective: every rank
    if rank == 0:                                   # the reference's `if default_gpu:` save
        sd = opt.state_dict()
        got = torch.cat([sd["state"][i]["exp_avg"].reshape(-1) for i in range(nslots)])
        assert bool((got > 0).all()), "every shard's moments reached the saving rank"
        owners = set(got.unique().tolist())
        assert owners == {float(r + 1) for r in range(world)}, owners
        assert all(int(sd["state"][i]["step"]) == 4 for i in range(nslots))
    dist.barrier()
    opt._fused["step"] = 5                          # another step invalidates the gathered copy
    with pytest.raises(RuntimeError, match="consolidate_state_dict"):
        opt.state_dict()
    dist.destroy_process_group()


def test_zero1_state_dict_is_collective_free_rank0_only_save_gloo():
    mp.spawn(_worker_zero1_checkpoint, args=(2, _free_port()), nprocs=2, join=True)
