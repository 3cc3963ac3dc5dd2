"""The CPU oracle (oracle/volta_ref.py) against fixtures produced by the imported reference
(oracle/make_golden.py).  CPU only; this is what pins the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import volta_ref as R

TINY = ["tiny_vilbert", "tiny_lxmert", "tiny_uniter", "tiny_visualbert", "tiny_vlbert", "tiny_gated"]
CTRL = ["ctrl_vilbert_base", "ctrl_lxmert", "ctrl_uniter_base", "ctrl_visualbert_base", "ctrl_vl-bert_base"]


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("name", TINY)
def test_tiny_forward_backward_matches_reference(golden_dir, name):
    z = load(golden_dir, name)
    cfg = R.RefConfig(json.loads(str(z["cfg_json"])))
    sd = {k[3:]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in z.items() if k.startswith("w::")}
    for alias, target in R.param_aliases(cfg).items():
        sd[alias] = sd[target]
    batch = {k[4:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("in::")}
    taps = {}
    lm, img, nsp = R.forward_from_batch(sd, cfg, batch, taps=taps)
    for got, key in ((lm, "loss_lm"), (img, "loss_img"), (nsp, "loss_nsp")):
        np.testing.assert_allclose(got.detach().numpy(), z["out::" + key], rtol=2e-6, atol=2e-6)
    for key in ("seq_t", "seq_v", "pooled_t", "pooled_v"):
        np.testing.assert_allclose(taps[key].detach().numpy(), z["out::" + key], rtol=0, atol=5e-6)
    (lm + img + nsp).sum().backward()
    for k, v in z.items():
        if k.startswith("out::grad::"):
            g = sd[k[len("out::grad::"):]].grad.numpy()
            np.testing.assert_allclose(g, v, rtol=0, atol=2e-6 + 1e-5 * np.abs(v).max())
    uniq = {id(t): t for t in sd.values()}
    total = np.sqrt(sum(float((t.grad.double() ** 2).sum()) for t in uniq.values() if t.grad is not None))
    np.testing.assert_allclose(total, z["out::grad_norm"][0], rtol=1e-5)


@pytest.mark.parametrize("name", TINY)
def test_param_inventory_matches_reference_state_dict(golden_dir, name):
    z = load(golden_dir, name)
    cfg = R.RefConfig(json.loads(str(z["cfg_json"])))
    ours = set(R.param_shapes(cfg)) | set(R.param_aliases(cfg))
    assert ours == set(str(k) for k in z["ref_keys"])
    for k, shape in R.param_shapes(cfg).items():
        assert tuple(z["w::" + k].shape) == tuple(shape)


@pytest.mark.parametrize("name", CTRL)
def test_ctrl_config_inventory(golden_dir, name):
    z = load(golden_dir, name)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = R.RefConfig.from_json_file(os.path.join(root, "config", name + ".json"))
    shapes = R.param_shapes(cfg)
    assert set(shapes) | set(R.param_aliases(cfg)) == set(str(k) for k in z["ref_keys"])
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(z["n_params"][0])


@pytest.mark.parametrize("name", ["ctrl_vilbert_base", "ctrl_visualbert_base"])
def test_ctrl_full_width_forward(golden_dir, name):
    """Real-width model at B=2 (config[0] of BASELINE.json is ctrl_visualbert_base B=2 on CPU)."""
    z = load(golden_dir, name)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = R.RefConfig.from_json_file(os.path.join(root, "config", name + ".json"))
    sd = R.make_weights(cfg, seed=3, std=0.03)
    batch = R.synthetic_batch(cfg, B=2, T=20, R=36, seed=7)
    taps = {}
    with torch.no_grad():
        lm, img, nsp = R.forward_from_batch(sd, cfg, batch, taps=taps)
    for got, key in ((lm, "loss_lm"), (img, "loss_img"), (nsp, "loss_nsp")):
        np.testing.assert_allclose(got.numpy(), z["out::" + key], rtol=1e-5)
    np.testing.assert_allclose(taps["seq_t"].numpy()[:, :, :64], z["out::seq_t_slice"], atol=2e-4)
    np.testing.assert_allclose(taps["seq_v"].numpy()[:, :8, :64], z["out::seq_v_slice"], atol=2e-4)
    np.testing.assert_allclose(taps["pooled_t"].numpy(), z["out::pooled_t"], atol=2e-4)


def test_ctrl_vilbert_b32_forward_matches_reference(golden_dir):
    """The oracle at a real batch (B=32, the cpu_baseline's own sample size) against the reference's fixture."""
    z = load(golden_dir, "ctrl_vilbert_base_b32")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = R.RefConfig.from_json_file(os.path.join(root, "config", "ctrl_vilbert_base.json"))
    sd = R.make_weights(cfg, seed=3, std=0.03)
    batch = R.synthetic_batch(cfg, B=32, T=20, R=36, seed=7)
    taps = {}
    with torch.no_grad():
        lm, img, nsp = R.forward_from_batch(sd, cfg, batch, taps=taps)
    for got, key in ((lm, "loss_lm"), (img, "loss_img"), (nsp, "loss_nsp")):
        np.testing.assert_allclose(got.numpy(), z["out::" + key], rtol=1e-5)
    np.testing.assert_allclose(taps["seq_t"].numpy()[::16, :, :64], z["out::seq_t_slice"], atol=2e-4)
    np.testing.assert_allclose(taps["pooled_v"].numpy()[:, :64], z["out::pooled_v_slice"], atol=2e-4)


def test_single_stream_equals_vanilla_bert_on_concatenated_sequence():
    """SURVEY 8a': shared + single_ln sub-layers are plain BERT layers over cat(text, vision)."""
    cd = dict(vocab_size=50, hidden_size=32, num_attention_heads=4, intermediate_size=64, pooler_size=16, v_pooler_size=16,
              v_hidden_size=32, v_num_attention_heads=4, v_intermediate_size=64, v_feature_size=16, image_embeddings="uniter",
              add_global_imgfeat="first", tt_attn_sublayers=[0], tv_attn_sublayers=[0], vt_attn_sublayers=[0], vv_attn_sublayers=[0],
              t_ff_sublayers=[1], v_ff_sublayers=[1], shared_sublayers=[0, 1], single_ln_sublayers=[0, 1])
    cfg = R.RefConfig(cd)
    sd = R.make_weights(cfg, seed=1)
    g = torch.Generator().manual_seed(0)
    t, v = torch.randn(2, 5, 32, generator=g), torch.randn(2, 4, 32, generator=g)
    tm = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]])
    vm = torch.tensor([[1, 1, 1, 1], [1, 1, 0, 0]])
    t_mask = (1.0 - tm[:, None, None, :].float()) * -10000.0
    v_mask = (1.0 - vm[:, None, None, :].float()) * -10000.0
    to, vo = R.gated_attention(sd, cfg, 0, t, v, t_mask, v_mask, R.Dropper(False))
    x = torch.cat([t, v], 1)
    m = torch.cat([t_mask, v_mask], -1)
    p = "bert.encoder.layer.0."
    q, k, val = (R._heads(R.linear(x, sd, p + "attention_self." + n), 4) for n in ("query", "key", "value"))
    pr = torch.softmax(q @ k.transpose(-1, -2) / np.sqrt(8.0) + m, -1)
    ctx = R._merge(pr @ val)
    y = R.layer_norm(R.linear(ctx, sd, p + "attention_output.dense") + x, sd[p + "attention_output.LayerNorm.weight"],
                     sd[p + "attention_output.LayerNorm.bias"])
    np.testing.assert_allclose(torch.cat([to, vo], 1).numpy(), y.numpy(), atol=2e-6)


def test_masked_rows_only_losses_equal_all_rows():
    """The engine evaluates the LM / region heads on labelled rows only; same loss as all rows."""
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(4, 6, 50, generator=g)
    labels = torch.full((4, 6), -1)
    labels[0, 2], labels[3, 1], labels[2, 5] = 7, 9, 11
    full = torch.nn.functional.cross_entropy(logits.view(-1, 50), labels.view(-1), ignore_index=-1)
    rows = (labels.view(-1) != -1).nonzero()[:, 0]
    sub = torch.nn.functional.cross_entropy(logits.view(-1, 50)[rows], labels.view(-1)[rows])
    np.testing.assert_allclose(full.numpy(), sub.numpy(), rtol=1e-6)
    pred = torch.randn(4, 6, 1601, generator=g)
    tgt = torch.softmax(torch.randn(4, 6, 1601, generator=g), -1)
    lab = torch.where(torch.rand(4, 6, generator=g) < 0.3, 1, -1)
    ref = torch.nn.KLDivLoss(reduction="none")(torch.log_softmax(pred, 2), tgt)
    ref = (ref * (lab == 1)[..., None].float()).sum() / max(int((lab == 1).sum()), 1)
    np.testing.assert_allclose(R.kl_1601(pred, 1.0, lab, tgt).numpy(), ref.numpy(), rtol=1e-6)


def test_adamw_matches_torch_adam_when_no_decay():
    """pytorch-transformers AdamW differs from torch.optim.Adam only by eps placement:
    sqrt(v)+eps vs sqrt(v_hat)+eps; identical when eps is rescaled by sqrt(1-b2^t)."""
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(1000, generator=g)
    grads = [torch.randn(1000, generator=g) for _ in range(5)]
    p, m, v = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    q = p0.clone().double()
    mq, vq = torch.zeros(1000).double(), torch.zeros(1000).double()
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-6
    for t, gr in enumerate(grads, 1):
        R.adamw_step(p, gr, m, v, t, lr, b1, b2, eps, 0.0, True)
        gd = gr.double()
        mq = b1 * mq + (1 - b1) * gd
        vq = b2 * vq + (1 - b2) * gd * gd
        q = q - lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * mq / (vq.sqrt() + eps)
    np.testing.assert_allclose(p.numpy(), q.float().numpy(), atol=1e-6)
    # decoupled decay is applied AFTER the adam update with the un-corrected lr
    p2, m2, v2 = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    R.adamw_step(p2, grads[0], m2, v2, 1, lr, b1, b2, eps, 0.01, True)
    p3, m3, v3 = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    R.adamw_step(p3, grads[0], m3, v3, 1, lr, b1, b2, eps, 0.0, True)
    np.testing.assert_allclose(p2.numpy(), (p3 * (1 - lr * 0.01)).numpy(), atol=1e-7)


def test_adamw_matches_torch_optim_adam_step_by_step():
    """The same identity against the real torch.optim.Adam (not a restated formula): with eps_t = eps / sqrt(1 - b2^t) set before every
    step, torch's update  lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps_t)  IS pytorch-transformers'  lr sqrt(bc2) / bc1 * m / (sqrt(v) + eps)."""
    g = torch.Generator().manual_seed(1)
    p0 = torch.randn(4096, generator=g)
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-6
    q = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([q], lr=lr, betas=(b1, b2), eps=eps)
    p, m, v = p0.clone(), torch.zeros(4096), torch.zeros(4096)
    for t in range(1, 21):
        gr = torch.randn(4096, generator=g) * (1.0 + 0.1 * t)
        R.adamw_step(p, gr, m, v, t, lr, b1, b2, eps, 0.0, True)
        opt.param_groups[0]["eps"] = eps / np.sqrt(1.0 - b2 ** t)
        q.grad = gr.double()
        opt.step()
        np.testing.assert_allclose(p.numpy(), q.detach().float().numpy(), atol=2e-6, rtol=0)
    st = opt.state[q]
    np.testing.assert_allclose(m.numpy(), st["exp_avg"].float().numpy(), atol=1e-6)
    np.testing.assert_allclose(v.numpy(), st["exp_avg_sq"].float().numpy(), atol=1e-6)


def test_warmup_linear_schedule_matches_the_successor_package():
    """pytorch-transformers 1.1.0 (absent here) was renamed `transformers`; its WarmupLinearSchedule lives on there as
    get_linear_schedule_with_warmup.  The image has transformers: the oracle's multiplier and the product's WarmupLinearSchedule are
    held against it step by step (same lineage, later version -- the strongest pin available for this piece of SURVEY.md 8a-17)."""
    transformers = pytest.importorskip("transformers")
    from volta_amd.optimization import WarmupLinearSchedule
    for warm, total in ((10, 100), (0, 50), (100, 100000), (7, 8)):
        mk = lambda: torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        o_ref, o_own = mk(), mk()
        ref = transformers.get_linear_schedule_with_warmup(o_ref, num_warmup_steps=warm, num_training_steps=total)
        own = WarmupLinearSchedule(o_own, warmup_steps=warm, t_total=total)
        for step in range(0, min(total + 5, 400)):
            want = o_ref.param_groups[0]["lr"]
            assert abs(o_own.param_groups[0]["lr"] - want) <= 1e-12, (warm, total, step)
            assert abs(R.warmup_linear(step, warm, total) - want) <= 1e-12, (warm, total, step)
            o_ref.step(); ref.step(); o_own.step(); own.step()


def test_warmup_linear_schedule():
    assert R.warmup_linear(0, 10, 100) == 0.0
    assert R.warmup_linear(5, 10, 100) == 0.5
    assert R.warmup_linear(10, 10, 100) == 1.0
    assert abs(R.warmup_linear(55, 10, 100) - 0.5) < 1e-12
    assert R.warmup_linear(100, 10, 100) == 0.0


def test_philox_known_answer():
    """Philox4x32 known-answer vectors (Random123 kat_vectors): counter 0/key 0, all-ones and the pi digits."""
    z = np.zeros(1, np.uint32)
    assert [hex(int(x)) for x in R.philox_raw(z, z, z, z, 0, 0)[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = np.full(1, 0xFFFFFFFF, np.uint32)
    assert [hex(int(x)) for x in R.philox_raw(f, f, f, f, 0xFFFFFFFF, 0xFFFFFFFF)[0]] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    # the 7-round variant of the dropout streams (same kat_vectors file, "philox4x32 7" lines)
    assert [hex(int(x)) for x in R.philox_raw(z, z, z, z, 0, 0, rounds=7)[0]] == ["0x5f6fb709", "0xd893f64", "0x4f121f81", "0x4f730a48"]
    assert [hex(int(x)) for x in R.philox_raw(f, f, f, f, 0xFFFFFFFF, 0xFFFFFFFF, rounds=7)[0]] == ["0x5207ddc2", "0x45165e59", "0x4d8ee751", "0x8c52f662"]
    a = lambda v: np.array([v], np.uint32)
    assert [hex(int(x)) for x in R.philox_raw(a(0x243f6a88), a(0x85a308d3), a(0x13198a2e), a(0x03707344), 0xa4093822, 0x299f31d0, rounds=7)[0]] == \
        ["0x4dfccaba", "0x190a87f0", "0xc47362ba", "0xb6b5242a"]
    # engine contract: element (row, c) of a [rows, C] site = word c&3 of Philox-4x32-7(counter (c>>2, row, site, 0))
    u = R.philox_u32(0x1234567800000042, 5, 3, 10)
    w = R.philox_raw(np.array([2], np.uint32), np.array([1], np.uint32), np.array([5], np.uint32), z, 0x42, 0x12345678, rounds=R.DROPOUT_PHILOX_ROUNDS)[0]
    assert int(u[1, 9]) == int(w[1])
    keep = R.philox_keep_mask(1234, 3, (1000, 100), 0.1)
    assert abs(float(keep.float().mean()) - 0.9) < 5e-3


# ---- the other fusion methods / global-feature placements / visual targets
VARIANTS = ["var_lxmert_text", "var_vlbert_none", "var_sum_mse_kl", "var_vqa_nce", "var_wide"]


def _variant(z):
    cfg = R.RefConfig(json.loads(str(z["cfg_json"])))
    sd = {k: v.clone().requires_grad_(True) for k, v in R.make_weights(cfg, seed=int(z["weights_seed"][0])).items() if k in R.param_shapes(cfg)}
    for alias, target in R.param_aliases(cfg).items():
        sd[alias] = sd[target]
    batch = {k[4:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("in::")}
    kw = {}
    if "draw::row_across" in z:
        draws = {k[6:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("draw::")}
        B, Rn = batch["image_label"].shape
        # the stored draws ARE what nce_draws produces from the seed (the product's device generator is tested against the same function)
        again = R.nce_draws(seed=0x1234ABCD5, site=0, B=B, R=Rn)
        assert all(torch.equal(draws[k], again[k]) for k in draws)
        kw["nce_index"] = R.nce_negative_index(draws, B, Rn)
    return cfg, sd, batch, kw


@pytest.mark.parametrize("name", VARIANTS)
def test_variant_heads_match_reference(golden_dir, name):
    """Text-only / sum / VL-BERT-VQA / no fusion, global feature last or absent, and the mse / nce / xent_1600 / xent_400 / huber /
    xent_1601 targets (volta/losses.py:25-126) against the real model; nce_2048 through the harness shim described in make_golden.py."""
    z = load(golden_dir, name)
    cfg, sd, batch, kw = _variant(z)
    assert set(R.param_shapes(cfg)) | set(R.param_aliases(cfg)) == set(str(k) for k in z["ref_keys"])
    taps = {}
    lm, img, nsp = R.forward_from_batch(sd, cfg, batch, taps=taps, **kw)
    for got, key in ((lm, "loss_lm"), (img, "loss_img"), (nsp, "loss_nsp")):
        np.testing.assert_allclose(got.detach().numpy(), z["out::" + key], rtol=3e-6, atol=2e-6)
    for key in ("seq_t", "seq_v", "pooled_t", "pooled_v"):
        if "out::" + key in z:
            np.testing.assert_allclose(taps[key].detach().numpy(), z["out::" + key], rtol=0, atol=5e-6)
        else:
            assert taps[key] is None
    (lm + img + nsp).sum().backward()
    checked = 0
    for k, v in z.items():
        if k.startswith("out::grad::"):
            g = sd[k[len("out::grad::"):]].grad.numpy()
            np.testing.assert_allclose(g, v, rtol=0, atol=2e-6 + 2e-5 * np.abs(v).max())
            checked += 1
        elif k.startswith("out::gradslice::"):
            g = sd[k[len("out::gradslice::"):]].grad.numpy()
            np.testing.assert_allclose(g.reshape(g.shape[0], -1)[:16, :64], v, rtol=0, atol=2e-6 + 2e-5 * np.abs(v).max())
            np.testing.assert_allclose(np.sqrt((g.astype(np.float64) ** 2).sum()), z["out::gradnorm::" + k[len("out::gradslice::"):]][0], rtol=2e-5)
            checked += 1
    assert checked >= 6
    none = set(str(k) for k in z["out::grad_none"])
    assert none == set(k for k, t in sd.items() if k in R.param_shapes(cfg) and t.grad is None)      # e.g. the VQA text pooler in pre-training
    uniq = {id(t): t for t in sd.values()}
    total = np.sqrt(sum(float((t.grad.double() ** 2).sum()) for t in uniq.values() if t.grad is not None))
    np.testing.assert_allclose(total, z["out::grad_norm"][0], rtol=2e-5)


@pytest.mark.parametrize("name", ["lxmert", "vl-bert_base", "vilbert_base"])
def test_non_ctrl_config_matches_reference(golden_dir, name):
    """config/lxmert.json (text fusion, xent_1600 + xent_400 + huber_2048), config/vl-bert_base.json (no fusion, global feature last,
    xent_1601 + the masked-region word embedding) and config/vilbert_base.json (1024-wide vision stream with 8 heads of 128, co-attention
    sub-layers at 1024 / 8 heads for both streams) at full width, B=2."""
    z = load(golden_dir, "full_" + name)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = R.RefConfig.from_json_file(os.path.join(root, "config", name + ".json"))
    shapes = R.param_shapes(cfg)
    assert set(shapes) | set(R.param_aliases(cfg)) == set(str(k) for k in z["ref_keys"])
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(z["n_params"][0])
    sd = R.make_weights(cfg, seed=3, std=0.03)
    batch = R.synthetic_batch(cfg, B=2, T=20, R=36, seed=7)
    taps = {}
    with torch.no_grad():
        lm, img, nsp = R.forward_from_batch(sd, cfg, batch, taps=taps)
    for got, key in ((lm, "loss_lm"), (img, "loss_img"), (nsp, "loss_nsp")):
        np.testing.assert_allclose(got.numpy(), z["out::" + key], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(taps["seq_t"].numpy()[:, :, :64], z["out::seq_t_slice"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(taps["seq_v"].numpy()[:, :8, :64], z["out::seq_v_slice"], rtol=0, atol=5e-5)


@pytest.mark.parametrize("name", ["tiny_vilbert", "tiny_lxmert", "tiny_gated"])
def test_attention_maps_match_reference(golden_dir, name):
    """The oracle's attention maps (probabilities, query and key layers per attention sub-layer) against the real reference's
    BertModel.forward(output_all_attention_masks=True) under config.visualization (encoders.py:342-358, 858-886): what pins the checker of
    the engine's attention-map output."""
    z = load(golden_dir, "attn_maps_" + name)
    cfg = R.RefConfig(json.loads(str(z["cfg_json"])))
    sd = R.make_weights(cfg, seed=7)
    batch = R.synthetic_batch(cfg, B=3, T=6, R=4, seed=11, pad=True)
    taps = {}
    with torch.no_grad():
        R.bert_model(sd, cfg, batch["input_ids"], batch["image_feat"].clone(), batch["image_loc"], batch["segment_ids"], batch["input_mask"],
                     batch["image_mask"], taps=taps)
    maps = taps["attn_maps"]
    assert len(maps[0]) == len(maps[1]) == int(z["n_layers"][0])
    seen = 0
    for tag, side in (("t", maps[0]), ("v", maps[1])):
        for i, d in enumerate(side):
            for key in ("intra_attn", "inter_attn", "queries", "keys"):
                ref = z.get("%s%d::%s" % (tag, i, key))
                if ref is None:
                    assert d[key] is None, (tag, i, key)
                else:
                    np.testing.assert_allclose(d[key].numpy(), ref, rtol=0, atol=2e-6)
                    seen += 1
    assert seen > 0
