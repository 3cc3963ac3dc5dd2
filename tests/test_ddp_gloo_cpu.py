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


def _worker(rank, world, port, cap, mode="allreduce", wire="fp32", pad_to=1):
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
        red = BucketReducer(flat, mode=mode, wire=wire)
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
            want = (world + 1) / 2.0 * (it + 1) * (1 + o % 7)
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
