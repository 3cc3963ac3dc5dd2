"""The data-parallel wrapper on the real RCCL backend with a single-rank group (the 1-GPU box cannot host two
ranks): flat parameter broadcast, bucketed all-reduce on the side stream, event hand-offs and the epilogue wait
must leave exactly the gradients of a plain backward.  GPU only."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ddp_world1_matches_plain_backward():
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_engine_gpu import build
    from oracle import volta_ref as R
    from volta_amd.parallel import DistributedDataParallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29561")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        model, rcfg, sd = build("vilbert")
        batch = R.synthetic_batch(rcfg, 4, 20, 36, seed=7)
        cb = {k: v.cuda() for k, v in batch.items()}
        args = (cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"],
                cb["lm_label_ids"], cb["image_label"], cb["image_cls"], None, None, None, None, None, cb["is_match"])
        model.train()
        model.set_dropout_seed(5)
        sum(model(*args)).sum().backward()
        torch.cuda.synchronize()
        plain = {k: p.grad.clone() for k, p in model.named_parameters()}
        for p in model.parameters():
            p.grad = None
        ddp = DistributedDataParallel(model, message_size=2000000)     # several buckets
        model.set_dropout_seed(5)
        sum(ddp(*args)).sum().backward()
        torch.cuda.synchronize()
        eng = model._last[0]
        plan = ddp._plan(eng)
        assert len(plan) >= 3, "expected several buckets"
        covered = sorted(r for _, rs in plan for r in rs)
        assert covered[0][0] == 0 and sum(hi - lo for lo, hi in covered) >= sum(p.numel() for p in model.parameters())
        for k, p in model.named_parameters():
            if "embeddings.word_embeddings" in k or "token_type_embeddings" in k:
                # scatter-added with fp32 atomics: summation order (last bits) varies from run to run
                assert torch.allclose(p.grad, plain[k], rtol=1e-4, atol=1e-7), k
            else:
                assert torch.equal(p.grad, plain[k]), k
    finally:
        dist.destroy_process_group()


def test_gate_ordered_reduction_and_gate_started_weight_gradients_match_plain_backward(monkeypatch):
    """The switches of round 4 (profiles/r04_experiments.md section 5): the reducer ordered by flags and one-wave gates instead of stream events
    (VK_DDP_ORDER=gate) and the weight-gradient blocks released by the retiring workgroups of the dgrad listed behind them
    (VK_SIDE_START=gate) leave exactly the gradients of the plain backward, and no gate gives up."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_engine_gpu import build
    from oracle import volta_ref as R
    from volta_amd.parallel import DistributedDataParallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29562")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        model, rcfg, sd = build("vilbert")
        batch = R.synthetic_batch(rcfg, 4, 20, 36, seed=7)
        cb = {k: v.cuda() for k, v in batch.items()}
        args = (cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"],
                cb["lm_label_ids"], cb["image_label"], cb["image_cls"], None, None, None, None, None, cb["is_match"])
        model.train()
        model.set_dropout_seed(5)
        sum(model(*args)).sum().backward()
        torch.cuda.synchronize()
        plain = {k: p.grad.clone() for k, p in model.named_parameters()}
        monkeypatch.setenv("VK_DDP_ORDER", "gate")
        monkeypatch.setenv("VK_SIDE_START", "gate")
        model.__dict__["_engines"] = {}                    # the plan is compiled under the switches
        for rep in range(3):
            for p in model.parameters():
                p.grad = None
            ddp = DistributedDataParallel(model, message_size=2000000)
            assert ddp.reducer.gated
            ddp.reducer._reduce = lambda ranges: None       # one rank: the mean is the gradient itself; the ORDERING is what runs
            model.set_dropout_seed(5)
            sum(ddp(*args)).sum().backward()
            torch.cuda.synchronize()
            eng = model._last[0]
            assert eng.side_gate, "the side stream has a hardware queue of its own on this box"
            assert eng.soft_error() == 0 and int(ddp.reducer._err) == 0, "a gate gave up waiting"
            assert ddp.reducer._bucket >= 3
            for k, p in model.named_parameters():
                if "embeddings.word_embeddings" in k or "token_type_embeddings" in k:
                    assert torch.allclose(p.grad, plain[k], rtol=1e-4, atol=1e-7), k
                else:
                    assert torch.equal(p.grad, plain[k]), k
            model.__dict__["_ddp"] = None
    finally:
        dist.destroy_process_group()


def test_zero1_in_the_reference_construction_order():
    """mode="zero1" on the real engine with the reference's order of construction (model -> DistributedDataParallel -> optimizer -> first
    backward, train_concap.py:227-253): the optimizer is known to the arena from its construction on, so the wrapper's guard -- it refuses
    a backward whose sharded gradients no volta_amd optimizer would pick up -- lets the first backward through; without an optimizer it
    raises.  Two clip + AdamW steps leave the master weights of the unsharded wrapper (one-rank group: the shard is the whole bucket)."""
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_engine_gpu import build
    from oracle import volta_ref as R
    from volta_amd.parallel import DistributedDataParallel
    from volta_amd.optimization import AdamW, clip_grad_norm_
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29563")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        finals = {}
        for mode in ("allreduce", "zero1"):
            model, rcfg, sd = build("vilbert")
            batch = R.synthetic_batch(rcfg, 4, 20, 36, seed=7)
            cb = {k: v.cuda() for k, v in batch.items()}
            args = (cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"],
                    cb["lm_label_ids"], cb["image_label"], cb["image_cls"], None, None, None, None, None, cb["is_match"])
            model.train()
            model.set_dropout_seed(5)
            ddp = DistributedDataParallel(model, message_size=2000000, mode=mode)
            if mode == "zero1":
                with pytest.raises(RuntimeError, match="zero1"):
                    sum(ddp(*args)).sum().backward()              # nobody would step these shards
                model.set_dropout_seed(5)
            opt = AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
            for step in range(2):
                sum(ddp(*args)).sum().backward()                  # the FIRST backward comes before any optimizer.step()
                clip_grad_norm_(model.parameters(), 0.5)
                opt.step()
                opt.zero_grad()
            torch.cuda.synchronize()
            if mode == "zero1":
                assert ddp.reducer.sharded, "no bucket was sharded"
            finals[mode] = model._arena.master.clone()
            model.__dict__["_ddp"] = None
        err = float((finals["allreduce"] - finals["zero1"]).norm() / finals["allreduce"].norm())
        assert err <= 1e-6, err
    finally:
        dist.destroy_process_group()
