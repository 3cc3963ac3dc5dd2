"""SURVEY.md 8f-4 on the HIP engine: the other fusion methods (sum / text / vl-bert_vqa / none), global feature last or absent, the
mse / nce / xent_1600 / xent_400 / huber / xent_1601 visual targets (volta/losses.py:25-126) and VL-BERT's masked-region word -- at
true width against the oracle (which tests/test_oracle_golden.py pins to the real model on the var_* fixtures), and the two real
non-ctrl configs that keep the 768-wide geometry (config/lxmert.json, config/vl-bert_base.json) against fixtures of the real model."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import volta_ref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BASE = dict(vocab_size=3000, hidden_size=768, num_attention_heads=12, intermediate_size=3072, pooler_size=1024,
            max_position_embeddings=64, type_vocab_size=2, num_locs=5, add_global_imgfeat="first", v_feature_size=256,
            v_hidden_size=768, v_num_attention_heads=12, v_intermediate_size=3072, v_pooler_size=1024,
            visual_target_weights={"0": 1.0}, fusion_method="mul", v_initializer_range=0.02)
ENC = dict(tt_attn_sublayers=[0, 4], t_ff_sublayers=[1, 3, 5], tv_attn_sublayers=[2], vt_attn_sublayers=[2], vv_attn_sublayers=[4], v_ff_sublayers=[3, 5])
SINGLE = dict(tt_attn_sublayers=[0, 2], tv_attn_sublayers=[0, 2], vt_attn_sublayers=[0, 2], vv_attn_sublayers=[0, 2],
              t_ff_sublayers=[1, 3], v_ff_sublayers=[1, 3], shared_sublayers=[0, 1, 2, 3], single_ln_sublayers=[0, 1, 2, 3])
BIG = dict(BASE, v_feature_size=2048)
VARIANTS = {
    "lxmert_text": dict(BIG, image_embeddings="lxmert", fusion_method="text", add_global_imgfeat=None, num_locs=4, pooler_size=768,
                        visual_target_weights={"3": 6.667, "4": 6.667, "5": 6.667},
                        tt_attn_sublayers=[0, 2, 5], vv_attn_sublayers=[0, 5], tv_attn_sublayers=[4], vt_attn_sublayers=[4],
                        shared_sublayers=[4], t_ff_sublayers=[1, 3, 6], v_ff_sublayers=[1, 6]),
    "vlbert_none": dict(BASE, image_embeddings="vl-bert", type_vocab_size=3, image_head_ln=False, v_coordinate_embeddings_dim=32,
                        fusion_method="none", add_global_imgfeat="last", num_locs=4, pooler_size=None, v_pooler_size=None,
                        visual_target_weights={"6": 1.0}, **SINGLE),
    "sum_mse_kl": dict(BIG, image_embeddings="vilbert", fusion_method="sum", add_global_imgfeat=None,
                       visual_target_weights={"1": 2.0, "0": 0.5}, **ENC),
    "vqa_nce": dict(BIG, image_embeddings="vilbert", fusion_method="vl-bert_vqa", add_global_imgfeat=None,
                    visual_target_weights={"2": 1.5}, **ENC),
    "mul_last_two": dict(BASE, image_embeddings="uniter", add_global_imgfeat="last", visual_target_weights={"6": 0.7, "0": 1.0}, **SINGLE),
    # config/vilbert_base.json's geometry at reduced depth: 768 / 12 x 64 text, 1024 / 8 x 128 vision with a 1024-wide feed-forward, a
    # co-attention sub-layer that projects both streams to 1024 = 8 x 128, a sub-layer with text AND vision self-attention of different heads
    "wide_vilbert": dict(BASE, image_embeddings="vilbert", v_hidden_size=1024, v_num_attention_heads=8, v_intermediate_size=1024,
                         sublayer2attn_hidden_size={"2": 1024}, sublayer2num_attention_heads={"2": 8}, **ENC),
}


def rel_err(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


def build(cd, seed=2, std=0.04):
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLPreTraining
    rcfg = R.RefConfig(cd)
    sd = R.make_weights(rcfg, seed=seed, std=std)
    model = BertForVLPreTraining(BertConfig.from_dict(cd))
    assert set(model.state_dict().keys()) == set(sd.keys())
    model.load_state_dict(sd, strict=True)
    return model.cuda(), rcfg, sd


def run_model(model, cb):
    return model(cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"], cb["lm_label_ids"],
                 cb["image_label"], cb["image_cls"], cb.get("obj_labels"), cb.get("obj_confs"), cb.get("attr_labels"), cb.get("attr_confs"), None, cb["is_match"])


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant_forward_backward_parity(name, train):
    cd = VARIANTS[name]
    model, rcfg, sd = build(cd)
    B, T, Rn = 4, 20, 36
    bseed = 7
    while True:                                      # a batch that keeps a few masked regions AND a few masked tokens after the objective-1 relabel
        batch = R.synthetic_batch(rcfg, B, T, Rn, seed=bseed, pad=True)       # (an MLM loss over 3-4 tokens moves by 1-3e-3 with the bf16 noise alone)
        if int((batch["image_label"] == 1).sum()) >= 6 and int((batch["lm_label_ids"] != -1).sum()) >= 8:
            break
        bseed += 1
    seed = 0xABCDEF12345
    model.train(train)
    model.set_dropout_seed(seed)
    cb = {k: v.cuda() for k, v in batch.items()}
    lm, img, nsp = run_model(model, cb)
    torch.cuda.synchronize()
    eng = model._last[0]
    aliases = R.param_aliases(rcfg)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k not in aliases}
    full = dict(leaves)
    for a, t in aliases.items():
        full[a] = leaves[t]
    kw = {}
    if "2" in cd["visual_target_weights"]:
        # the device's negatives: same counter-based stream as the oracle's generator, bit for bit
        draws = R.nce_draws(seed, eng.nce_site, B, Rn)
        want_idx = R.nce_negative_index(draws, B, Rn)
        got_idx = eng.bufs["img_nce_neg"].cpu().view(B, Rn, -1).long()
        assert torch.equal(got_idx, want_idx)
        kw["nce_index"] = want_idx
    taps = {}
    olm, oimg, onsp = R.forward_from_batch(full, rcfg, batch, train=train, philox_seed=seed if train else None, taps=taps, **kw)
    for key, ref in taps.items():
        if key not in eng.taps or ref is None or not torch.is_tensor(ref) or ref.dim() < 2:
            continue
        if eng.taps[key] is None:
            continue
        e = rel_err(eng.taps[key].float().cpu().view(ref.shape), ref.detach())
        assert e < 2e-2, (name, key, e)
    for got, ref, nm in ((lm, olm, "lm"), (img, oimg, "img"), (nsp, onsp, "nsp")):
        g, r = float(got.detach()), float(ref.detach())
        # the regression / nce targets square or exponentiate bf16-rounded 2048-wide predictions: 3e-3; hard-label targets as the MLM loss
        # MLM at B = 4 (9 masked tokens): 1.1e-3 observed for one variant in training mode; ITM at B = 4 (four 2-way logits of bf16-noisy
        # pooled vectors): 1.2-1.4e-2 observed for two variants (the ctrl fixtures at B = 2 gate it at 3e-2, the B = 256 fixture at 1.5e-3)
        tol = 2e-2 if nm == "nsp" else (3e-3 if nm == "img" else 1.5e-3)
        assert abs(g - r) <= tol * max(abs(r), 1e-3) + 1e-4, (name, nm, g, r)
    assert float(oimg.detach()) > 0
    named = dict(model.named_parameters())
    # feature-regression / nce targets: d loss / d prediction = (prediction - feature) or a softmax over <feature, prediction>, both differences
    # of quantities that carry the bf16 noise of the 2048-wide prediction -- observed 4.5-5.6e-2 / cosine 0.9984 where the
    # hard-label targets give 1e-2 / 0.9999
    regress = bool(set(cd["visual_target_weights"]) & {"1", "2", "5"})
    # gradients once for the MLM + region losses, once for the ITM loss (B x 2 logits: the forward bf16 noise shows up as a common scale
    # error, rel. error 0.2-0.4 at B = 4; its tight gate is the B = 32 reference fixture of tests/test_fullsize_golden_gpu.py)
    passes = [("lm+img", (8e-2, 0.997) if regress else (7e-2, 0.9975))]        # worst observed without regression targets: 6.5e-2 / 0.9979 (a [1, F] vector)
    if float(onsp.detach()) != 0.0:
        passes.append(("nsp", (0.5, 0.9)))
    for which, (tol, min_cos) in passes:
        for p in named.values():
            p.grad = None
        for leaf in leaves.values():
            leaf.grad = None
        if which == "nsp":
            nsp.sum().backward()
            onsp.sum().backward()
        else:
            (lm + img).sum().backward(retain_graph=True)
            (olm + oimg).sum().backward(retain_graph=True)
        torch.cuda.synchronize()
        bad = []
        for k, leaf in leaves.items():
            if leaf.grad is None:                      # the reference's autograd leaves these parameters alone: so must the engine
                if which == "lm+img" and not k.startswith(("bert.t_pooler", "bert.v_pooler", "cls.bi_seq")):
                    assert named[k].grad is None or float(named[k].grad.float().norm()) == 0.0, (name, k)
                continue
            if named[k].grad is None:
                bad.append((k, "no gradient"))
                continue
            g_ref, g_got = leaf.grad, named[k].grad.float().cpu()
            assert torch.isfinite(g_got).all(), (name, k)
            # key biases: identically zero by the softmax's shift invariance (the reference leaves 1e-6-sized rounding residue); the engine's residue
            # is the column sum of bf16-rounded dK rows -- bounded against the query bias of the same sub-layer, whose gradient is the same kind of sum
            scale = float(leaves[k.replace("key.", "query.")].grad.norm()) if k.endswith("key.bias") and leaves[k.replace("key.", "query.")].grad is not None else 0.0
            if float(g_ref.norm()) < 1e-6 + 1e-4 * scale:
                if float(g_got.norm()) > 5e-3 + 0.3 * scale:
                    bad.append((k, "expected ~0", float(g_got.norm()), scale))
                continue
            e = rel_err(g_got, g_ref)
            cos = float((g_got * g_ref).sum() / (g_got.norm() * g_ref.norm()))
            qk = "query." in k or "key." in k
            if e > (max(0.1, tol) if qk else tol) or cos < (min(0.995, min_cos) if qk else min_cos):
                bad.append((k, e, cos, float(g_ref.norm())))
        assert not bad, (name, train, which, bad[:12], len(bad))


def test_variant_training_step_skips_unused_parameters():
    """vl-bert_vqa in pre-training: the text pooler has no gradient (reference fixture: grad None); clip + AdamW leave it and its moments alone."""
    from volta_amd.optimization import AdamW, clip_grad_norm_
    model, rcfg, sd = build(VARIANTS["vqa_nce"])
    model.train()
    model.set_dropout_seed(5)
    opt = AdamW([{"params": [p], "weight_decay": 0.01} for p in model.parameters()], lr=1e-3)
    cb = {k: v.cuda() for k, v in R.synthetic_batch(rcfg, 4, 20, 36, seed=7).items()}
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    first = None
    for _ in range(3):
        out = run_model(model, cb)
        sum(out).sum().backward()
        clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        opt.zero_grad()
        first = first if first is not None else [float(x) for x in out]
    last = [float(x) for x in out]
    torch.cuda.synchronize()
    after = dict(model.named_parameters())
    assert torch.equal(after["bert.t_pooler.dense.weight"].detach(), before["bert.t_pooler.dense.weight"])
    assert not torch.equal(after["cls.imagePredictions.decoder_dict.2.weight"].detach(), before["cls.imagePredictions.decoder_dict.2.weight"])
    assert last[0] + last[1] < first[0] + first[1] and last[2] == 0.0


@pytest.mark.parametrize("name", ["lxmert", "vl-bert_base", "vilbert_base"])
def test_non_ctrl_config_matches_reference_fixture(golden_dir, name):
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLPreTraining
    z = np.load(os.path.join(golden_dir, "full_" + name + ".npz"))
    path = os.path.join(ROOT, "config", name + ".json")
    rcfg = R.RefConfig(json.load(open(path)))
    sd = R.make_weights(rcfg, seed=3, std=0.03)
    model = BertForVLPreTraining(BertConfig.from_json_file(path))
    assert list(model.state_dict().keys()) == [str(k) for k in z["ref_keys"]]
    assert sum(p.numel() for p in model.parameters()) == int(z["n_params"][0])
    model.load_state_dict(sd, strict=True)
    model.cuda().eval()
    cb = {k: v.cuda() for k, v in R.synthetic_batch(rcfg, B=2, T=20, R=36, seed=7).items()}
    lm, img, nsp = run_model(model, cb)
    (lm + img + nsp).sum().backward()
    torch.cuda.synchronize()
    report = {}
    for got, key in ((lm, "loss_lm"), (img, "loss_img"), (nsp, "loss_nsp")):
        want = float(z["out::" + key][0])
        report[key] = abs(float(got) - want) / max(abs(want), 1e-6) if want else abs(float(got))
    eng = model._last[0]
    H, Hv = rcfg.hidden_size, rcfg.v_hidden_size
    report["seq_t"] = rel_err(eng.taps["seq_t"].float().cpu().view(2, 20, H)[:, :, :64], torch.from_numpy(z["out::seq_t_slice"]))
    Rv = eng.Rv
    report["seq_v"] = rel_err(eng.taps["seq_v"].float().cpu().view(2, Rv, Hv)[:, :8, :64], torch.from_numpy(z["out::seq_v_slice"]))
    # B = 2: MLM / region losses 1.5e-3 (as the ctrl fixtures at B = 2), ITM 3e-2 (two samples of bf16-noisy logits)
    assert report["loss_lm"] <= 1.5e-3 and report["loss_img"] <= 3e-3 and report["loss_nsp"] <= 3e-2, report
    assert report["seq_t"] <= 2e-2 and report["seq_v"] <= 2e-2, report
    named = dict(model.named_parameters())
    none = set(str(k) for k in z["out::grad_none"])
    assert none == {k for k, p in named.items() if p.grad is None}
    for k in z.files:
        if not k.startswith("out::gradslice::"):
            continue
        pname = k[len("out::gradslice::"):]
        g = named[pname].grad.float().cpu().numpy()
        gn = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
        want_n = float(z["out::gradnorm::" + pname][0])
        gs = g.reshape(g.shape[0], -1)[:16, :64] if g.ndim > 1 else g[:64]
        cos = float((gs.ravel() @ z[k].ravel()) / (np.linalg.norm(gs) * np.linalg.norm(z[k]) + 1e-30))
        itm = pname in ("bert.t_pooler.dense.weight", "cls.bi_seq_relationship.weight")
        assert abs(gn - want_n) / want_n <= (0.1 if itm else 3e-2) and cos >= (0.98 if itm else 0.995), (pname, gn, want_n, cos)
    total = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    assert abs(total - float(z["out::grad_norm"][0])) / float(z["out::grad_norm"][0]) <= 1e-2, report
    print(name, {k: float("%.2e" % v) for k, v in report.items()})


@pytest.mark.parametrize("name", ["lxmert_text", "sum_mse_kl", "vlbert_none"])
def test_score_returning_branch(name):
    """BertForVLPreTraining.forward without labels (volta/encoders.py:1113-1114): the heads' scores at every position against the oracle's
    taps -- text scores, every visual target's scores, the ITM score where the fusion method has one, the fused pooled vector."""
    cd = VARIANTS[name]
    model, rcfg, sd = build(cd)
    model.eval()
    batch = R.synthetic_batch(rcfg, 3, 20, 36, seed=5, pad=True)
    cb = {k: v.cuda() for k, v in batch.items()}
    st, sv, itm, maps, pooled = model(cb["input_ids"], cb["image_feat"], cb["image_loc"], cb["segment_ids"], cb["input_mask"], cb["image_mask"])
    torch.cuda.synchronize()
    taps = {}
    with torch.no_grad():
        R.forward_from_batch(sd, rcfg, batch, taps=taps)
    assert rel_err(st.float().cpu(), taps["scores_t"]) <= 3e-2
    assert set(sv) == set(taps["scores_v_dict"])
    for ix, t in sv.items():
        assert rel_err(t.float().cpu(), taps["scores_v_dict"][ix]) <= 3e-2, ix
    if taps["itm"] is None:
        assert itm is None
    else:
        assert rel_err(itm.float().cpu(), taps["itm"]) <= 5e-2
    assert maps == ([], [])
    assert (pooled is None) == (cd["fusion_method"] == "none")
