"""Two ranks on two MI355X over RCCL / xGMI: the data-parallel wrapper's gradients must equal the mean of the two ranks' plain
single-rank backward passes (apex/apex/parallel/distributed.py:451-454), for every reduction mode of the bucket reducer.
Skipped on a one-GPU box (the driver's multi-GPU node runs it)."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, shared_gpu=False):
    """`shared_gpu`: every rank on cuda:0 with gloo collectives (RCCL refuses two ranks on one device) -- how tools/ddp_check.py runs the same
    assertions on a one-GPU box."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0 if shared_gpu else rank)
    if shared_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from test_engine_gpu import build
    from oracle import volta_ref as R
    from volta_amd.parallel import DistributedDataParallel
    try:
        model, rcfg, sd = build("vilbert")
        model.eval()

        def args_of(seed):
            cb = {k: v.cuda() for k, v in R.synthetic_batch(rcfg, 4, 20, 36, seed=seed).items()}
            return (cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"],
                    cb["lm_label_ids"], cb["image_label"], cb["image_cls"], None, None, None, None, None, cb["is_match"])

        plain = []
        for r in range(world):               # every rank computes both ranks' local gradients without the wrapper
            for p in model.parameters():
                p.grad = None
            sum(model(*args_of(100 + r))).sum().backward()
            torch.cuda.synchronize()
            plain.append(model._arena.grad.clone())
        want = sum(plain) / world
        for mode, wire in (("allreduce", "fp32"), ("rs_ag", "fp32"), ("rs_ag", "bf16")):
            for p in model.parameters():
                p.grad = None
            ddp = DistributedDataParallel(model, message_size=2000000, mode=mode, wire=wire)
            sum(ddp(*args_of(100 + rank))).sum().backward()
            torch.cuda.synchronize()
            got = model._arena.grad
            err = float((got - want).norm() / want.norm())
            assert err <= (1e-2 if wire == "bf16" else 1e-5), (mode, wire, err)
            assert ddp.reducer.bytes_on_wire > 0
            model.__dict__["_ddp"] = None
        # replicas stay in lock-step: identical averaged gradients on both ranks (amp_master_params/compare.py:12-26)
        mine = model._arena.grad.clone()
        other = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        assert all(torch.equal(o, other[0]) for o in other)
        # ---- sharded optimizer (mode "zero1"): three training steps (clip + AdamW) against the unsharded wrapper from the same weights:
        # master weights and bf16 copies bit-identical between the two paths and between the ranks, moments equal once gathered
        from volta_amd.optimization import AdamW, clip_grad_norm_
        runs = {}
        for mode in ("allreduce", "zero1"):
            m2, _, _ = build("vilbert")
            m2.train()
            m2.set_dropout_seed(1234)
            ddp = DistributedDataParallel(m2, message_size=2000000, mode=mode)
            opt = AdamW(m2.parameters(), lr=1e-3, weight_decay=0.01)
            for step in range(3):
                sum(ddp(*args_of(200 + 10 * step + rank))).sum().backward()
                clip_grad_norm_(m2.parameters(), 0.5)      # small enough to be active; the plain call of train_concap.py:307
                opt.step()
                opt.zero_grad()
            torch.cuda.synchronize()
            opt.consolidate_state_dict()           # collective (a no-op for the unsharded wrapper); state_dict() itself never is
            sd = opt.state_dict()["state"]
            runs[mode] = (m2._arena.master.clone(), m2._arena.shadow.clone(), torch.cat([sd[i]["exp_avg"].reshape(-1) for i in sorted(sd)]),
                          torch.cat([sd[i]["exp_avg_sq"].reshape(-1) for i in sorted(sd)]), ddp.reducer.bytes_on_wire)
            if mode == "zero1":
                assert ddp.reducer.sharded, "no bucket was sharded"
            m2.__dict__["_ddp"] = None
        # (between the two RUNS a last-bit tolerance: the backward's embedding and loss row sums are accumulated with float atomics, so
        # two backward passes over the same batch already differ in the last bit (tests/test_fullsize_gpu.py), and with more than two
        # ranks the collectives add a slot's contributions in different ring orders.  Within a run the replicas are bit-identical.)
        for a, b2, what in zip(runs["allreduce"][:4], runs["zero1"][:4], ("master weights", "bf16 copies", "exp_avg", "exp_avg_sq")):
            err = float((a.float() - b2.float()).norm() / b2.float().norm())
            assert err <= 1e-6, "zero1 vs unsharded: %s differ by %g" % (what, err)
        mine = runs["zero1"][0]
        other = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        assert all(torch.equal(o, other[0]) for o in other), "zero1: replicas differ"
    finally:
        dist.destroy_process_group()


def test_ddp_world2_rccl_matches_mean_of_plain_backwards():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one-GPU box)")
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port), nprocs=2, join=True)
