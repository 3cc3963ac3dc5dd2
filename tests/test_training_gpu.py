"""Several optimizer steps in a row: the HIP step (forward + backward + clipping + AdamW + schedule, dropout on) against the oracle run the same
way on the same batch, with the engine's Philox masks replayed step by step.  A single-step test cannot see an error that only builds up through the
weights (a stale bf16 weight copy, a moment buffer one step behind, a schedule off by one); this one can."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["gated", "uniter"])
def test_loss_trajectory_follows_the_oracle(name):
    from test_engine_gpu import build
    from oracle import volta_ref as R
    from volta_amd.optimization import AdamW, WarmupLinearSchedule, clip_grad_norm_
    model, rcfg, sd = build(name)
    model.train()
    steps, lr0, warm, total_steps, max_norm, seed = 12, 5e-5, 3, 40, 1.0, 0x5EED1234
    batch = R.synthetic_batch(rcfg, 8, 20, 36, seed=11, pad=True)
    cb = {k: v.cuda() for k, v in batch.items()}
    args = (cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"],
            cb["lm_label_ids"], cb["image_label"], cb["image_cls"], None, None, None, None, None, cb["is_match"])
    model.materialize()
    groups = [{"params": [p], "lr": lr0, "weight_decay": 0.01 if R.decays(k) else 0.0} for k, p in model.named_parameters()]
    opt = AdamW(groups, lr=lr0, eps=1e-6, betas=(0.9, 0.999))
    sched = WarmupLinearSchedule(opt, warmup_steps=warm, t_total=total_steps)
    # oracle state: fp32 leaves (tied weights share one leaf), moments
    aliases = R.param_aliases(rcfg)
    leaves = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if k not in aliases}
    m = {k: torch.zeros_like(v) for k, v in leaves.items()}
    v2 = {k: torch.zeros_like(v) for k, v in leaves.items()}
    got, want, norms = [], [], []
    for step in range(1, steps + 1):
        model.set_dropout_seed(seed + step)
        lm, img, nsp = model(*args)
        (lm + img + nsp).backward()
        gn = clip_grad_norm_(model.parameters(), max_norm, defer_to_optimizer=True)
        opt.step()
        sched.step()
        opt.zero_grad()
        got.append(float((lm + img + nsp).detach()))
        # the oracle's step
        full = dict(leaves)
        for a, t in aliases.items():
            full[a] = leaves[t]
        for leaf in leaves.values():
            leaf.grad = None
        olm, oimg, onsp = R.forward_from_batch(full, rcfg, batch, train=True, philox_seed=seed + step)
        (olm + oimg + onsp).backward()
        grads = {k: (leaf.grad if leaf.grad is not None else torch.zeros_like(leaf)) for k, leaf in leaves.items()}
        ref_norm = R.clip_grad_norm(list(grads.values()), max_norm)
        lr = lr0 * R.warmup_linear(step - 1, warm, total_steps)
        with torch.no_grad():
            for k, leaf in leaves.items():
                R.adamw_step(leaf, grads[k], m[k], v2[k], step, lr, 0.9, 0.999, 1e-6, 0.01 if R.decays(k) else 0.0, True)
        want.append(float((olm + oimg + onsp).detach()))
        norms.append((float(gn), float(ref_norm)))
    torch.cuda.synchronize()
    # (at a learning rate ten times higher the fixed batch is memorised within six steps, 9.75 -> 1.8, and the two runs, 4e-4 apart at step 2,
    # are 2 % apart at step 5: so fast a descent amplifies any difference -- the problem's conditioning, not a property of the kernels)
    # the loss falls (a fixed batch, 12 steps), and the two trajectories stay together: 1e-3 at the first step (the single-step contract), within
    # 1 % after twelve updates through bf16 activations (observed: 3.4e-3 while the loss falls from 9.75 to 2.85)
    print("losses", got, want)
    assert want[-1] < 0.99 * want[0] and got[-1] < 0.99 * got[0], (got, want)
    assert abs(got[0] - want[0]) <= 2e-3 * abs(want[0]), (got[0], want[0])
    worst = max(abs(g - w) / abs(w) for g, w in zip(got, want))
    assert worst <= 1e-2, (worst, got, want)
    assert all(abs(a - b) <= 5e-2 * b for a, b in norms), norms        # the clipping norm (always active here: norm > 1)
    assert min(b for _, b in norms) > max_norm
    # the weights after 12 steps.  Adam divides by sqrt(v): an element whose gradient is smaller than the bf16 noise moves by +-lr per step in a
    # direction the noise decides, so the comparison is on the MOVEMENT of all weights together (direction and size), and on each tensor's distance
    # from the oracle's relative to its own norm
    named = dict(model.named_parameters())
    dot = n_got = n_ref = 0.0
    worst_rel = (0.0, None)
    for k, leaf in leaves.items():
        w0, w_ref, w_got = sd[k].float(), leaf.detach(), named[k].detach().float().cpu()
        dg, dr = (w_got - w0).double(), (w_ref - w0).double()
        dot, n_got, n_ref = dot + float((dg * dr).sum()), n_got + float((dg * dg).sum()), n_ref + float((dr * dr).sum())
        rel = float((w_got - w_ref).norm() / (w_ref.norm() + 1e-12))
        if rel > worst_rel[0] and float(w_ref.norm()) > 1e-3:
            worst_rel = (rel, k)
    cos, ratio = dot / (n_got ** 0.5 * n_ref ** 0.5), (n_got / n_ref) ** 0.5
    print("trajectory", name, "loss", got[0], "->", got[-1], "oracle", want[0], "->", want[-1], "worst loss gap", worst, "movement cosine", cos, "size ratio", ratio,
          "worst tensor", worst_rel)
    assert cos >= 0.995 and 0.99 <= ratio <= 1.01, (cos, ratio)        # observed 0.9995 / 1.0003
    assert worst_rel[0] <= 5e-3, worst_rel                             # observed 4.7e-4


@pytest.mark.parametrize("task", ["TASK1", "TASK9"])
def test_finetuning_trajectory_follows_the_oracle(task):
    """The task model over several AdamW steps (eval-mode forward: no dropout, so no masks to replay): VQA-style soft-target BCE on the pooled
    classifier (TASK1, volta/task_utils.py's loss for VL-classifier: mean BCE x number of labels) and region cross-entropy on the V-logit head (TASK9)."""
    import torch.nn.functional as F
    from test_engine_gpu import CONFIGS
    from test_tasks_gpu import TASK_CFG
    from oracle import volta_ref as R
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLTasks
    from volta_amd.optimization import AdamW, clip_grad_norm_
    cd = dict(CONFIGS["vilbert"], clf_hidden_size=1536)
    rcfg = R.RefConfig(cd)
    sd = R.make_task_weights(rcfg, TASK_CFG, [task], seed=4, std=0.04)
    model = BertForVLTasks(BertConfig.from_dict(cd), TASK_CFG, [task])
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    B, T, Rn, steps, lr = 8, 20, 36, 6, 2e-5
    batch = R.synthetic_batch(rcfg, B, T, Rn, seed=9, pad=True)
    cb = {k: v.cuda() for k, v in batch.items()}
    g = torch.Generator().manual_seed(3)
    if task == "TASK1":
        target = (torch.rand(B, 3129, generator=g) < 0.002).float() * torch.rand(B, 3129, generator=g)
        loss_fn = lambda pred, tgt: F.binary_cross_entropy_with_logits(pred, tgt, reduction="mean") * tgt.size(1)
    else:
        live = batch["image_mask"].sum(1)                                       # padded regions carry the -10000 mask: the labelled region is a live one
        target = (torch.rand(B, generator=g) * live).long().clamp(max=Rn)
        loss_fn = lambda pred, tgt: F.cross_entropy(pred.squeeze(2), tgt)
    model.materialize()
    groups = [{"params": [p], "lr": lr, "weight_decay": 0.01 if R.decays(k) else 0.0} for k, p in model.named_parameters()]
    opt = AdamW(groups, lr=lr, eps=1e-6, betas=(0.9, 0.999))
    aliases = R.param_aliases(rcfg)
    leaves = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if k not in aliases}
    m = {k: torch.zeros_like(v) for k, v in leaves.items()}
    v2 = {k: torch.zeros_like(v) for k, v in leaves.items()}
    got, want = [], []
    for step in range(1, steps + 1):
        pred = model(cb["input_ids"], cb["image_feat"], cb["image_loc"], task, cb["segment_ids"], cb["input_mask"], cb["image_mask"])[0]
        loss = loss_fn(pred, target.cuda())
        loss.backward()
        clip_grad_norm_(model.parameters(), 1.0, defer_to_optimizer=True)
        opt.step()
        opt.zero_grad()
        got.append(float(loss.detach()))
        full = dict(leaves)
        for a, t in aliases.items():
            full[a] = leaves[t]
        for leaf in leaves.values():
            leaf.grad = None
        o = loss_fn(R.tasks_forward(full, rcfg, TASK_CFG, task, batch["input_ids"], batch["image_feat"].clone(), batch["image_loc"], batch["segment_ids"],
                                    batch["input_mask"], batch["image_mask"]), target)
        o.backward()
        used = {k: leaf for k, leaf in leaves.items() if leaf.grad is not None}       # the pre-training-only leaves get no gradient and no update
        R.clip_grad_norm([leaf.grad for leaf in used.values()], 1.0)
        with torch.no_grad():
            for k, leaf in used.items():
                R.adamw_step(leaf, leaf.grad, m[k], v2[k], step, lr, 0.9, 0.999, 1e-6, 0.01 if R.decays(k) else 0.0, True)
        want.append(float(o.detach()))
    worst = max(abs(a - b) / abs(b) for a, b in zip(got, want))
    print("finetune", task, got, want, worst)
    assert want[-1] < want[0] and got[-1] < got[0], (got, want)
    assert worst <= 5e-3, (worst, got, want)            # observed 4e-4 (TASK1)
