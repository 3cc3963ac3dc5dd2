"""Fused AdamW / grad-norm kernels against the oracle's per-tensor restatement (pytorch-transformers semantics)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_adamw_clip_schedule_match_oracle():
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_engine_gpu import build
    from oracle import volta_ref as R
    from volta_amd.optimization import AdamW, WarmupLinearSchedule, clip_grad_norm_
    model, rcfg, sd = build("gated")
    model.materialize()
    named = list(model.named_parameters())
    groups = []                                    # one group per parameter, as train_concap.py:213-224
    for k, p in named:
        groups.append({"params": [p], "lr": 1e-3, "weight_decay": 0.01 if R.decays(k) else 0.0})
    opt = AdamW(groups, lr=1e-3, eps=1e-6, betas=(0.9, 0.999))
    sched = WarmupLinearSchedule(opt, warmup_steps=2, t_total=10)
    # oracle state
    op = {k: p.detach().float().cpu().clone() for k, p in named}
    om = {k: torch.zeros_like(v) for k, v in op.items()}
    ov = {k: torch.zeros_like(v) for k, v in op.items()}
    g = torch.Generator().manual_seed(0)
    for step in range(1, 5):
        grads = {k: torch.randn(v.shape, generator=g) * (3.0 if step == 2 else 0.01) for k, v in op.items()}
        for k, p in named:
            model._arena.view(k, "grad").copy_(grads[k].cuda())
            p.grad = model._arena.view(k, "grad")
        total = clip_grad_norm_(model.parameters(), 5.0, defer_to_optimizer=(step % 2 == 0))
        ref_total = R.clip_grad_norm(list(grads.values()), 5.0)
        assert abs(float(total) - float(ref_total)) <= 1e-4 * float(ref_total)
        lr = 1e-3 * R.warmup_linear(step - 1, 2, 10)
        opt.step()
        sched.step()
        for k in op:
            R.adamw_step(op[k], grads[k], om[k], ov[k], step, lr, 0.9, 0.999, 1e-6, 0.01 if R.decays(k) else 0.0, True)
        torch.cuda.synchronize()
        worst = max(float((p.detach().cpu() - op[k]).abs().max()) for k, p in named)
        assert worst < 2e-6, (step, worst)
        sh = model._arena.view(named[3][0], "shadow").float().cpu()
        assert torch.equal(sh, op[named[3][0]].bfloat16().float()) or float((sh - op[named[3][0]]).abs().max()) < 1e-2
    assert model._arena.shadow_version == model._arena.master._version
