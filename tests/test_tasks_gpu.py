"""BertForVLTasks on the HIP engine (encoder forward + backward through the engine, task heads in torch) against the CPU oracle:
predictions of every head type and the gradients of encoder, pooler and head parameters for loss = sum(prediction * probe).
True layer widths, reduced depth, eval mode (the oracle is pinned by the real reference in tests/test_tasks_cpu.py).  GPU only."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

TASK_CFG = {"TASK1": {"type": "VL-classifier", "num_labels": 3129}, "TASK9": {"type": "V-logit"}, "TASK10": {"type": "V-logit", "num_clf_layers": 2},
            "TASK12": {"type": "VL-binary-classifier"}, "TASK13": {"type": "VL-tri-classifier"}, "TASK8": {"type": "VL-logit"}}


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.mark.parametrize("cfg_name", ["vilbert", "uniter", "vilbert:vl-bert_vqa", "vilbert:sum", "vilbert:text"])
def test_task_heads_forward_backward_parity(cfg_name):
    """`name:fusion` runs the other fusion methods of the task model (encoders.py:1184-1195): sum, text-only, and vl-bert_vqa, whose
    text pooler (VLBertTextPooler, :610-623) takes the token two places before each caption's end -- gathered and pooled by the engine."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_engine_gpu import CONFIGS
    from oracle import volta_ref as R
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLTasks
    cfg_name, _, fusion = cfg_name.partition(":")
    cd = dict(CONFIGS[cfg_name], clf_hidden_size=1536)
    ids = list(TASK_CFG)
    if fusion:
        cd["fusion_method"] = fusion
        ids = ["TASK1", "TASK9", "TASK12"]
    rcfg = R.RefConfig(cd)
    sd = R.make_task_weights(rcfg, TASK_CFG, ids, seed=4, std=0.04)
    model = BertForVLTasks(BertConfig.from_dict(cd), TASK_CFG, ids)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    B, T, Rn = 4, 20, 36
    batch = R.synthetic_batch(rcfg, B, T, Rn, seed=9, pad=True)
    cb = {k: v.cuda() for k, v in batch.items()}
    aliases = R.param_aliases(rcfg)
    for t in ids:
        for p in model.parameters():
            p.grad = None
        pred = model(cb["input_ids"], cb["image_feat"], cb["image_loc"], t, cb["segment_ids"], cb["input_mask"], cb["image_mask"])[0]
        probe = torch.randn(pred.shape, generator=torch.Generator().manual_seed(sum(map(ord, t))))
        (pred * probe.cuda()).sum().backward()
        torch.cuda.synchronize()
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k not in aliases}
        full = dict(leaves)
        for a, tgt in aliases.items():
            full[a] = leaves[tgt]
        want = R.tasks_forward(full, rcfg, TASK_CFG, t, batch["input_ids"], batch["image_feat"].clone(), batch["image_loc"], batch["segment_ids"],
                               batch["input_mask"], batch["image_mask"])
        (want * probe).sum().backward()
        got = pred.detach().float().cpu()
        if TASK_CFG[t]["type"].startswith("V-logit"):          # padded regions carry the -10000 mask: compare exactly there, relatively elsewhere
            live = batch["image_mask"].bool().unsqueeze(2)
            assert torch.equal(got[~live] < -5000, torch.ones_like(got[~live], dtype=torch.bool))
            e = rel(got[live], want.detach()[live])
        else:
            e = rel(got, want.detach())
        assert e <= 3e-2, (t, e)
        named = dict(model.named_parameters())
        checked, worst, bad = 0, 0.0, []
        pooled_only = not TASK_CFG[t]["type"].startswith("V-logit")
        for k, leaf in leaves.items():
            if leaf.grad is None or k not in named or k.endswith("key.bias"):      # d/d(key bias) is identically zero (softmax shift invariance)
                continue
            g = named[k].grad
            if leaf.grad.norm() < 1e-9:
                assert g is None or float(g.norm()) <= 1e-4, k
                continue
            assert g is not None, (t, k)
            assert g.data_ptr() == model._arena.view(k, "grad").data_ptr(), k         # every gradient lives in the arena
            # pooled-output tasks back-propagate through ONE token per sample (B = 4): the same bf16 forward noise as the ITM head
            # of the pre-training model (tests/test_engine_gpu.py); region-logit tasks use every region
            tol = (0.2 if ("query." in k or "key." in k) else 0.15) if pooled_only else (0.12 if ("query." in k or "key." in k) else 6e-2)
            if leaf.grad.dim() == 1:      # bias / LayerNorm vectors are column sums over few rows: bf16 rounding of the summands shows (observed cos 0.9925)
                tol = max(tol, 0.25 if pooled_only else 0.15)
            e_k = rel(g.float().cpu(), leaf.grad)
            worst = max(worst, e_k)
            if e_k > tol:
                bad.append((k, round(e_k, 4), tol))
            checked += 1
        assert not bad, (t, bad[:10], len(bad), checked)
        assert checked > 20, checked
        print(cfg_name, t, "prediction %.2e  worst gradient %.2e over %d parameters" % (e, worst, checked))


def test_task_model_trains_with_fused_optimizer():
    """One fine-tuning step end to end: encoder and torch-side head parameters are both updated by the fused AdamW."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_engine_gpu import CONFIGS
    from oracle import volta_ref as R
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLTasks
    from volta_amd.optimization import AdamW, clip_grad_norm_
    cd = dict(CONFIGS["gated"], clf_hidden_size=1536)
    rcfg = R.RefConfig(cd)
    tc = {"TASK1": TASK_CFG["TASK1"]}
    model = BertForVLTasks(BertConfig.from_dict(cd), tc, ["TASK1"]).cuda().train()
    model.set_dropout_seed(5)
    opt = AdamW(model.parameters(), lr=1e-3)
    batch = {k: v.cuda() for k, v in R.synthetic_batch(rcfg, 4, 20, 36, seed=2, pad=True).items()}
    target = torch.randint(0, 3129, (4,), device="cuda")
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    losses = []
    for _ in range(3):
        pred = model(batch["input_ids"], batch["image_feat"], batch["image_loc"], "TASK1", batch["segment_ids"], batch["input_mask"], batch["image_mask"])[0]
        loss = torch.nn.functional.cross_entropy(pred, target)
        loss.backward()
        clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        opt.zero_grad()
        losses.append(float(loss.detach()))
    after = model.state_dict()
    assert losses[-1] < losses[0], losses
    for k in ("clfs_dict.TASK1.logit_fc.3.weight", "bert.encoder.layer.0.attention_self.query.weight", "bert.t_pooler.dense.weight"):
        assert not torch.equal(after[k], before[k]), k


def test_task_model_dropout_prob_argument():
    """BertForVLTasks(dropout_prob=...) (encoders.py:1118-1122) sets the dropout on the fused pooled vector: with every other dropout of
    the configuration at 0, training-mode predictions equal the evaluation ones for dropout_prob = 0, differ for 0.5, and the share of
    zeroed pooled elements -- read off a linear head on a constant-weight probe -- follows the argument."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_engine_gpu import CONFIGS
    from oracle import volta_ref as R
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLTasks
    cd = dict(CONFIGS["vilbert"], clf_hidden_size=1536, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
              v_hidden_dropout_prob=0.0, v_attention_probs_dropout_prob=0.0)
    rcfg = R.RefConfig(cd)
    sd = R.make_task_weights(rcfg, TASK_CFG, ["TASK8"], seed=4, std=0.04)
    batch = R.synthetic_batch(rcfg, 32, 20, 36, seed=9, pad=True)
    cb = {k: v.cuda() for k, v in batch.items()}
    args = (cb["input_ids"], cb["image_feat"], cb["image_loc"], "TASK8", cb["segment_ids"], cb["input_mask"], cb["image_mask"])
    preds = {}
    for p_drop in (0.0, 0.1, 0.5):
        model = BertForVLTasks(BertConfig.from_dict(cd), TASK_CFG, ["TASK8"], dropout_prob=p_drop)
        model.load_state_dict(sd, strict=True)
        model = model.cuda()
        model.eval()
        with torch.no_grad():
            ev = model(*args)[0].float().cpu()
        model.train()
        model.set_dropout_seed(77)
        with torch.no_grad():
            tr = model(*args)[0].float().cpu()
        preds[p_drop] = (ev, tr)
    assert torch.allclose(preds[0.0][0], preds[0.0][1], atol=1e-6), "dropout_prob = 0: training mode equals evaluation mode"
    assert torch.allclose(preds[0.0][0], preds[0.5][0]), "evaluation mode does not depend on dropout_prob"
    d1 = float((preds[0.1][1] - preds[0.1][0]).abs().mean()), float((preds[0.5][1] - preds[0.5][0]).abs().mean())
    assert d1[0] > 0 and d1[1] > 1.5 * d1[0], d1            # the perturbation grows with the probability (std ~ sqrt(p / (1 - p)): 3 x from 0.1 to 0.5)
    with pytest.raises(ValueError):
        BertForVLTasks(BertConfig.from_dict(cd), TASK_CFG, ["TASK8"], dropout_prob=1.5)
