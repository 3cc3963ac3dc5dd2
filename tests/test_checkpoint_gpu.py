"""Save / resume in the reference's checkpoint format (volta/train_utils.py:295-340): an interrupted run resumed from
`pytorch_ckpt_latest.tar` continues bit-identically (embedding tables: to 1e-6, their gradients are accumulated with atomics); the optimizer state has pytorch_transformers.AdamW's layout
({"step", "exp_avg", "exp_avg_sq"} per parameter, indexed in param_groups order).  GPU only."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(seed=2):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_engine_gpu import build
    from oracle import volta_ref as R
    from volta_amd.optimization import AdamW, WarmupLinearSchedule
    model, rcfg, sd = build("gated", seed=seed)
    model.eval()                                  # dropout off: the resumed run must not depend on the step counter of the masks
    groups = [{"params": [p], "lr": 1e-3, "weight_decay": 0.01 if R.decays(k) else 0.0} for k, p in model.named_parameters()]
    opt = AdamW(groups, lr=1e-3, eps=1e-6, betas=(0.9, 0.999))
    sched = WarmupLinearSchedule(opt, warmup_steps=2, t_total=10)
    return model, rcfg, opt, sched


def _step(model, opt, sched, batch):
    from volta_amd.optimization import clip_grad_norm_
    lm, img, nsp = model(batch["input_ids"], batch["image_feat"], batch["image_loc"], batch["segment_ids"], batch["input_mask"], batch["image_mask"],
                         batch["lm_label_ids"], batch["image_label"], batch["image_cls"], None, None, None, None, None, batch["is_match"])
    (lm + img + nsp).backward()
    clip_grad_norm_(model.parameters(), 5.0)
    opt.step()
    sched.step()
    opt.zero_grad()


def test_save_resume_continues_bit_identically(tmp_path):
    from oracle import volta_ref as R
    model, rcfg, opt, sched = _setup()
    batches = [{k: v.cuda() for k, v in R.synthetic_batch(rcfg, 4, 20, 36, seed=20 + i, pad=True).items()} for i in range(3)]
    _step(model, opt, sched, batches[0])
    _step(model, opt, sched, batches[1])
    # the reference's save() payload
    ckpt = tmp_path / "pytorch_ckpt_latest.tar"
    torch.save({"model_state_dict": {"module." + k: v for k, v in model.state_dict().items()},      # as saved from a DDP wrapper
                "optimizer_state_dict": opt.state_dict(), "scheduler_state_dict": sched.state_dict(),
                "global_step": 2, "epoch_id": 0, "tb_logger": None, "score": None}, ckpt)
    osd = opt.state_dict()
    n_params = sum(1 for _ in model.parameters())
    assert sorted(osd["state"].keys()) == list(range(n_params))
    st0 = osd["state"][0]
    assert set(st0.keys()) == {"step", "exp_avg", "exp_avg_sq"} and st0["step"] == 2
    assert st0["exp_avg"].shape == next(model.parameters()).shape
    _step(model, opt, sched, batches[2])
    torch.cuda.synchronize()
    want = {k: v.detach().clone() for k, v in model.state_dict().items()}

    # resume() (train_utils.py:319-340) into a fresh model / optimizer / scheduler
    model2, _, opt2, sched2 = _setup(seed=9)
    ck = torch.load(ckpt, map_location="cpu", weights_only=False)
    model2.load_state_dict({k.replace("module.", "", 1): v for k, v in ck["model_state_dict"].items()})
    sched2.load_state_dict(ck["scheduler_state_dict"])
    opt2.load_state_dict(ck["optimizer_state_dict"])
    _step(model2, opt2, sched2, batches[2])
    torch.cuda.synchronize()
    for k, v in model2.state_dict().items():
        if "embeddings" in k or k == "cls.predictions.decoder.weight":      # tables (and the decoder tied to one): gradients accumulated with fp32 atomics, order not fixed
            assert float((v - want[k]).abs().max()) <= 1e-6, k
        else:
            assert torch.equal(v, want[k]), k
