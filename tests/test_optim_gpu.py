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
    assert model._arena.shadow_version == model._arena.param_version()


def _args(rcfg, seed=7, B=4):
    from oracle import volta_ref as R
    cb = {k: v.cuda() for k, v in R.synthetic_batch(rcfg, B, 20, 36, seed=seed).items()}
    return (cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"],
            cb["lm_label_ids"], cb["image_label"], cb["image_cls"], None, None, None, None, None, cb["is_match"])


def _same(a, b, tol=2e-6):
    return all(abs(x - y) <= tol * max(abs(y), 1.0) for x, y in zip(a, b))


def _build(name="gated", seed=2):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_engine_gpu import build
    return build(name, seed=seed)


def test_bf16_weight_copies_follow_load_state_dict_and_torch_optimizers():
    """The GEMMs read a bf16 shadow of the fp32 parameters.  Whatever writes the parameters through torch after the first
    forward -- load_state_dict, a stock torch optimizer, a manual re-initialisation -- must be seen by the next forward."""
    model, rcfg, sd = _build()
    model.eval()
    args = _args(rcfg)
    with torch.no_grad():
        first = [float(x) for x in model(*args)]
        other, _, sd2 = _build(seed=5)
        model.load_state_dict(sd2, strict=True)                   # after the arena exists and the shadow was built
        got = [float(x) for x in model(*args)]
        want = [float(x) for x in other.eval()(*args)]
    assert _same(got, want) and not _same(got, first, 1e-3), (first, got, want)
    # a stock torch optimizer on model.parameters(): the next forward must use the updated weights
    model.train()
    model.set_dropout_seed(3)
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    sum(model(*args)).sum().backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    model.eval()
    with torch.no_grad():
        after = [float(x) for x in model(*args)]
        fresh, _, _ = _build(seed=5)
        fresh.load_state_dict({k: v.detach().cpu() for k, v in model.state_dict().items()}, strict=True)
        want = [float(x) for x in fresh.cuda().eval()(*args)]
    assert _same(after, want) and not _same(after, got, 1e-3), (got, after, want)
    # in-place edits through .data bypass every version counter: the documented escape hatch
    with torch.no_grad():
        for p in model.parameters():
            p.data.mul_(0.5)
        model._arena.invalidate_shadow()
        halved = [float(x) for x in model(*args)]
    assert not _same(halved, after, 1e-3)


def test_frozen_and_gradient_less_parameters_are_skipped():
    """config.fixed_layers / freeze_layers (volta/train_utils.py:250-255): frozen parameters get no .grad, are left out of the
    clipping norm and are not touched by AdamW (no decay either) -- pytorch_transformers' `if p.grad is None: continue`."""
    from volta_amd.optimization import AdamW, clip_grad_norm_
    model, rcfg, sd = _build()
    model.train()
    model.set_dropout_seed(11)
    frozen = [n for n, _ in model.named_parameters() if n.startswith("bert.encoder.layer.0.") or n.startswith("bert.embeddings.")]
    assert frozen
    for n, p in model.named_parameters():
        p.requires_grad_(n not in frozen)
    model.materialize()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    trainable = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    opt = AdamW([{"params": [p], "weight_decay": 0.01} for _, p in trainable], lr=1e-3)
    args = _args(rcfg)
    sum(model(*args)).sum().backward()
    named = dict(model.named_parameters())
    assert all(named[n].grad is None for n in frozen) and all(p.grad is not None for _, p in trainable)
    want = torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in trainable))      # frozen slices of the arena hold gradients too: left out
    total = clip_grad_norm_(model.parameters(), 1e-3)            # tiny max-norm: the clip engages
    assert abs(float(total) - float(want)) <= 1e-4 * float(want)
    assert abs(float(torch.sqrt(sum((p.grad.double() ** 2).sum() for _, p in trainable))) - 1e-3) <= 1e-5     # clipped in place, frozen chunks untouched
    opt.step()
    torch.cuda.synchronize()
    for n, p in model.named_parameters():
        if n in frozen:
            assert torch.equal(p.detach(), before[n]), n
        else:
            assert not torch.equal(p.detach(), before[n]), n
    # a trainable parameter without a gradient this step is skipped too, and a later step picks it up again
    opt.zero_grad()
    sum(model(*args)).sum().backward()
    victim_n, victim = trainable[-1]
    snap, vgrad = victim.detach().clone(), victim.grad
    victim.grad = None
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(victim.detach(), snap)
    victim.grad = vgrad
    opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(victim.detach(), snap)
    # a foreign tensor in .grad is refused instead of silently stepping with the arena's stale gradient
    victim.grad = torch.zeros_like(victim)
    with pytest.raises(RuntimeError):
        opt.step()


def test_pipelined_adamw_under_the_next_forward_changes_nothing():
    """AdamW(overlap_with_forward=True): the update runs range by range on its own stream while the next forward waits only for the
    ranges it is about to read -- same losses, same parameters, same moments as the single-launch step."""
    from volta_amd.optimization import AdamW, clip_grad_norm_
    results = []
    for overlap in (False, True):
        model, rcfg, sd = _build("vilbert")
        model.train()
        model.set_dropout_seed(21)
        model.materialize()
        opt = AdamW([{"params": [p], "weight_decay": 0.01} for p in model.parameters()], lr=5e-3, overlap_with_forward=overlap, overlap_ranges=5)
        args = _args(rcfg)
        losses = []
        for _ in range(4):
            out = model(*args)
            sum(out).sum().backward()
            clip_grad_norm_(model.parameters(), 5.0, defer_to_optimizer=True)
            opt.step()
            opt.zero_grad()
            losses.append([x.detach().clone() for x in out])          # no host sync inside the loop: the overlap is real
        if overlap:
            assert model._arena.opt_pending is not None                # the last step is still owed a wait
            eng = model._last[0]
            segs = eng.fwd_segments(opt._fused["bounds"])
            assert len(segs) >= 3 and segs[0][0] >= 1 and segs[-1][0] == 5 and segs[-1][2] == len(eng.fwd.ops)
            assert all(a[2] == b[1] for a, b in zip(segs, segs[1:])) and segs[0][1] == 0
        opt.synchronize()
        torch.cuda.synchronize()
        results.append(([[float(x) for x in l] for l in losses], model._arena.master.clone(), opt._fused["m"].clone(), model._arena.shadow.clone()))
    (l0, p0, m0, s0), (l1, p1, m1, s1) = results
    assert all(_same(a, b, 1e-5) for a, b in zip(l0, l1)), (l0, l1)
    # word-embedding rows are scatter-added with float atomics (last-bit order effects); everything else is bit-identical
    assert float((p0 - p1).abs().max()) <= 1e-5 and float((m0 - m1).abs().max()) <= 1e-5
    assert float((s0.float() - s1.float()).abs().max()) <= 1e-2
    assert l0[0] != l0[-1]                                             # the model did move


@pytest.mark.parametrize("ncus", [1, 24, 256])
def test_adamw_step_on_gives_the_bits_of_adamw_step(ncus):
    """vk_adamw_step_on (a few resident workgroups striding over the arena: the form the pipelined step runs under the next forward) leaves
    exactly the parameters, moments and bf16 copies of vk_adamw_step -- chunk classes, skipped chunks, a clip coefficient and an arena that is
    not a multiple of the kernel's 4-chunk blocks included."""
    import ctypes as C
    from volta_amd import _lib as L
    nch = 4 * 24 * 4 * 3 + 7                         # three full trips of 24 workgroups and a ragged tail
    n = nch * 1024
    gen = torch.Generator(device="cuda").manual_seed(5)
    g = torch.randn(n, device="cuda", generator=gen) * 0.1
    p0 = torch.randn(n, device="cuda", generator=gen)
    m0 = torch.randn(n, device="cuda", generator=gen) * 0.01
    v0 = torch.rand(n, device="cuda", generator=gen) * 1e-3
    cls = torch.randint(0, 3, (nch,), device="cuda", generator=gen).to(torch.uint8)
    cls[torch.rand(nch, device="cuda", generator=gen) < 0.1] = 255          # VK_CHUNK_SKIP
    clip = torch.tensor([7.0, 0.71], device="cuda")
    outs = []
    for which in ("wide", "narrow"):
        p, m, v = p0.clone(), m0.clone(), v0.clone()
        sh = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
        a = L.AdamwArgs()
        a.p, a.g, a.m, a.v, a.shadow, a.chunk_class, a.clip, a.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr(), cls.data_ptr(), clip.data_ptr(), n
        for i, (lr, wd) in enumerate([(1.0, 0.01), (0.5, 0.0), (2.0, 0.1)]):
            a.cls_lr_mult[i], a.cls_wd[i] = lr, wd
        a.lr, a.beta1, a.beta2, a.eps, a.step_mult, a.grad_scale = 1e-3, 0.9, 0.999, 1e-6, 1.7, 0.5
        for _ in range(2):
            if which == "wide":
                L.check(L.lib.vk_adamw_step(C.byref(a), L.stream_ptr()))
            else:
                L.check(L.lib.vk_adamw_step_on(C.byref(a), ncus, L.stream_ptr()))
        torch.cuda.synchronize()
        outs.append((p, m, v, sh))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    skipped = (cls == 255).repeat_interleave(1024)
    assert torch.equal(outs[1][0][skipped], p0[skipped]) and not torch.equal(outs[1][0][~skipped], p0[~skipped])
