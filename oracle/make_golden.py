"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference) on CPU.

Build-container only (the reference never travels to the GPU box).  Nothing of the reference is
copied: this script imports it, feeds it weights/batches produced by oracle/volta_ref.py's
deterministic generators and stores inputs + expected outputs.

Harness-side shims (no reference file is modified):
  * empty stub modules for boto3 / botocore / requests-free import of volta/utils.py:20-22
  * Tensor.cuda -> identity, because volta/embeddings.py:383 hard-codes .cuda()

Usage:  python oracle/make_golden.py            (writes every fixture)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import volta_ref as R  # noqa: E402


def import_reference():
    sys.dont_write_bytecode = True
    for name in ("boto3", "botocore", "botocore.exceptions"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["botocore.exceptions"].ClientError = type("ClientError", (Exception,), {})
    sys.modules["botocore"].exceptions = sys.modules["botocore.exceptions"]
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from volta.config import BertConfig
    from volta.encoders import BertForVLPreTraining
    return BertConfig, BertForVLPreTraining


TINY_BASE = dict(
    vocab_size=200, hidden_size=64, num_attention_heads=4, intermediate_size=128, pooler_size=48,
    max_position_embeddings=64, type_vocab_size=2, num_locs=5, add_global_imgfeat="first",
    v_feature_size=32, v_hidden_size=64, v_num_attention_heads=4, v_intermediate_size=128, v_pooler_size=48,
    visual_target_weights={"0": 1.0}, fusion_method="mul", v_initializer_range=0.02,
)


def tiny_configs():
    c = {}
    c["tiny_vilbert"] = dict(TINY_BASE, image_embeddings="vilbert",
                             tt_attn_sublayers=[0, 2, 6], tv_attn_sublayers=[4], vt_attn_sublayers=[4], vv_attn_sublayers=[6],
                             t_ff_sublayers=[1, 3, 5, 7], v_ff_sublayers=[5, 7])
    c["tiny_lxmert"] = dict(TINY_BASE, image_embeddings="lxmert",
                            tt_attn_sublayers=[0, 2, 5], vv_attn_sublayers=[0, 5], tv_attn_sublayers=[4], vt_attn_sublayers=[4],
                            shared_sublayers=[4], t_ff_sublayers=[1, 3, 6], v_ff_sublayers=[1, 6])
    single = dict(tt_attn_sublayers=[0, 2], tv_attn_sublayers=[0, 2], vt_attn_sublayers=[0, 2], vv_attn_sublayers=[0, 2],
                  t_ff_sublayers=[1, 3], v_ff_sublayers=[1, 3], shared_sublayers=[0, 1, 2, 3], single_ln_sublayers=[0, 1, 2, 3])
    c["tiny_uniter"] = dict(TINY_BASE, image_embeddings="uniter", **single)
    c["tiny_visualbert"] = dict(TINY_BASE, image_embeddings="visualbert", **single)
    c["tiny_vlbert"] = dict(TINY_BASE, image_embeddings="vl-bert", type_vocab_size=3, image_head_ln=False,
                            v_coordinate_embeddings_dim=4, **single)
    # gated general case: text attends to text AND vision with UNSHARED weights (joint softmax over two key sets)
    c["tiny_gated"] = dict(TINY_BASE, image_embeddings="vilbert",
                           tt_attn_sublayers=[0], tv_attn_sublayers=[0], vt_attn_sublayers=[0], vv_attn_sublayers=[0],
                           t_ff_sublayers=[1], v_ff_sublayers=[1])
    return c


GRAD_KEYS = ["bert.embeddings.word_embeddings.weight", "bert.encoder.layer.0.attention_self.query.weight",
             "bert.encoder.layer.0.attention_output.LayerNorm.weight", "bert.encoder.layer.1.output.dense.bias",
             "cls.predictions.transform.dense.weight", "cls.imagePredictions.decoder_dict.0.weight",
             "bert.t_pooler.dense.weight", "cls.bi_seq_relationship.weight"]


def run_reference(BertConfig, Model, cfg_dict, sd, batch, want_grads=True):
    cfg = BertConfig.from_dict(cfg_dict)
    model = Model(cfg)
    missing = model.load_state_dict(sd, strict=True)
    model.eval()
    b = {k: v.clone() for k, v in batch.items()}
    lm, img, nsp = model(b["input_ids"], b["image_feat"], b["image_loc"], b["segment_ids"], b["input_mask"],
                         b["image_mask"], b["lm_label_ids"], b["image_label"], b["image_cls"], None, None, None,
                         None, None, b["is_match"])
    seq_t, seq_v, pt, pv, _ = model.bert(b["input_ids"], batch["image_feat"].clone(), b["image_loc"], b["segment_ids"],
                                         b["input_mask"], b["image_mask"])
    out = dict(loss_lm=lm.detach().numpy(), loss_img=img.detach().numpy(), loss_nsp=nsp.detach().numpy(),
               seq_t=seq_t.detach().numpy(), seq_v=seq_v.detach().numpy(),
               pooled_t=pt.detach().numpy(), pooled_v=pv.detach().numpy())
    if want_grads:
        (lm + img + nsp).sum().backward()
        named = dict(model.named_parameters())
        for k in GRAD_KEYS:
            if k in named and named[k].grad is not None:
                out["grad::" + k] = named[k].grad.numpy().copy()
        total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))
        out["grad_norm"] = np.array([float(total)])
    return out, model


def write_tiny(BertConfig, Model):
    for name, cd in tiny_configs().items():
        cfg = R.RefConfig(cd)
        sd = R.make_weights(cfg, seed=7)
        batch = R.synthetic_batch(cfg, B=3, T=6, R=4, seed=11, pad=True)
        out, model = run_reference(BertConfig, Model, cd, sd, batch)
        # the reference's own key order / shapes pin param_shapes()
        ref_keys = list(model.state_dict().keys())
        blob = {"cfg_json": np.array(__import__("json").dumps(cd)), "ref_keys": np.array(ref_keys)}
        blob.update({"in::" + k: v.numpy() for k, v in batch.items()})
        blob.update({"w::" + k: v.numpy() for k, v in sd.items()})
        blob.update({"out::" + k: v for k, v in out.items()})
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, "losses", out["loss_lm"], out["loss_img"], out["loss_nsp"], os.path.getsize(path) // 1024, "KB")


def write_ctrl(BertConfig, Model):
    """The five real ctrl_* configs at B=2 (T=20, R=36 / 100 for vl-bert): weights come from the seed
    generator (not stored), so only losses, pooled vectors, a hidden-state slice and checksums are."""
    import json
    for name in ("ctrl_vilbert_base", "ctrl_lxmert", "ctrl_uniter_base", "ctrl_visualbert_base", "ctrl_vl-bert_base"):
        cd = json.load(open(os.path.join(ROOT, "config", name + ".json")))
        cfg = R.RefConfig(cd)
        sd = R.make_weights(cfg, seed=3, std=0.03)
        Rn = 100 if "vl-bert" in name else 36
        batch = R.synthetic_batch(cfg, B=2, T=20, R=Rn, seed=7)
        out, model = run_reference(BertConfig, Model, cd, sd, batch, want_grads=True)
        n_params = sum(p.numel() for p in model.parameters())
        blob = {"n_params": np.array([n_params]), "n_keys": np.array([len(model.state_dict())]),
                "ref_keys": np.array(list(model.state_dict().keys()))}
        for k in ("loss_lm", "loss_img", "loss_nsp", "pooled_t", "pooled_v", "grad_norm"):
            blob["out::" + k] = out[k]
        blob["out::seq_t_slice"] = out["seq_t"][:, :, :64]
        blob["out::seq_v_slice"] = out["seq_v"][:, :8, :64]
        blob["out::seq_t_sum"] = np.array([out["seq_t"].astype(np.float64).sum()])
        blob["out::seq_v_sum"] = np.array([out["seq_v"].astype(np.float64).sum()])
        for k in GRAD_KEYS[1:]:
            if "grad::" + k in out:
                g = out["grad::" + k]
                blob["out::gradslice::" + k] = g.reshape(g.shape[0], -1)[:16, :64] if g.ndim > 1 else g[:64]
                blob["out::gradnorm::" + k] = np.array([np.sqrt((g.astype(np.float64) ** 2).sum())])
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, n_params, "losses", out["loss_lm"], out["loss_img"], out["loss_nsp"], os.path.getsize(path) // 1024, "KB")
        del model, sd


# ------------------------------------------------------------------------------------------------------------------
# the other fusion methods, global-feature placements and visual targets (volta/losses.py:25-126)
def variant_configs():
    big = dict(TINY_BASE, v_feature_size=2048)           # the feature-regression targets predict 2048 values per region
    enc = dict(tt_attn_sublayers=[0, 2, 6], tv_attn_sublayers=[4], vt_attn_sublayers=[4], vv_attn_sublayers=[6],
               t_ff_sublayers=[1, 3, 5, 7], v_ff_sublayers=[5, 7])
    single = dict(tt_attn_sublayers=[0, 2], tv_attn_sublayers=[0, 2], vt_attn_sublayers=[0, 2], vv_attn_sublayers=[0, 2],
                  t_ff_sublayers=[1, 3], v_ff_sublayers=[1, 3], shared_sublayers=[0, 1, 2, 3], single_ln_sublayers=[0, 1, 2, 3])
    c = {}
    # lxmert.json's heads: text-only fusion, no global feature, 4 box coordinates, detector-label + regression targets
    c["var_lxmert_text"] = dict(big, image_embeddings="lxmert", fusion_method="text", add_global_imgfeat=None, num_locs=4,
                                visual_target_weights={"3": 6.667, "4": 6.667, "5": 6.667},
                                tt_attn_sublayers=[0, 2, 5], vv_attn_sublayers=[0, 5], tv_attn_sublayers=[4], vt_attn_sublayers=[4],
                                shared_sublayers=[4], t_ff_sublayers=[1, 3, 6], v_ff_sublayers=[1, 6])
    # vl-bert_base.json's heads: no poolers, no ITM head, global feature LAST, hard labels over 1601 classes
    c["var_vlbert_none"] = dict(TINY_BASE, image_embeddings="vl-bert", type_vocab_size=3, image_head_ln=False, v_coordinate_embeddings_dim=4,
                                fusion_method="none", add_global_imgfeat="last", num_locs=4, visual_target_weights={"6": 1.0}, **single)
    c["var_sum_mse_kl"] = dict(big, image_embeddings="vilbert", fusion_method="sum", add_global_imgfeat=None,
                               visual_target_weights={"1": 2.0, "0": 0.5}, **enc)
    c["var_vqa_nce"] = dict(big, image_embeddings="vilbert", fusion_method="vl-bert_vqa", add_global_imgfeat=None,
                            visual_target_weights={"2": 1.5}, **enc)
    # config/vilbert_base.json's geometry in small: a wider vision stream with its own head count and intermediate size, co-attention
    # sub-layers that project BOTH streams to a third width (sublayer2attn_hidden_size / sublayer2num_attention_heads), equal poolers
    c["var_wide"] = dict(TINY_BASE, image_embeddings="vilbert", v_hidden_size=96, v_num_attention_heads=3, v_intermediate_size=80,
                         pooler_size=48, v_pooler_size=48, sublayer2attn_hidden_size={"4": 96}, sublayer2num_attention_heads={"4": 3}, **enc)
    return c


VARIANT_GRAD_KEYS = GRAD_KEYS + ["cls.imagePredictions.decoder_dict.%s.weight" % ix for ix in "123456"] + \
    ["cls.imagePredictions.transform.dense.weight", "bert.v_pooler.dense.weight", "bert.encoder.layer.0.attention_self.v_query.weight"]


class _FedRandom:
    """Replays explicit draws through Tensor.random_ (losses.py:50-51,59): the n-th call returns the n-th queued tensor.
    nce_2048 as written cannot run on any torch >= 0.4: its index tensors and its class target come from `image_target.new(...)`,
    i.e. they are FLOAT tensors, and `flat_image_target[neg_index_v]` (losses.py:75) raises "tensors used as indices must be long".
    Inside this context `Tensor.new(*sizes)` returns int64 storage, which is what the code evidently intends; nothing else changes."""

    def __init__(self, queue):
        self.queue, self.orig, self.orig_new = list(queue), torch.Tensor.random_, torch.Tensor.new

    def __enter__(self):
        q, orig_new = self.queue, self.orig_new

        def fake(t, *a, **k):
            v = q.pop(0)
            assert tuple(v.shape) == tuple(t.shape), (v.shape, t.shape)
            return t.copy_(v.to(t.dtype))

        def new_long(t, *sizes, **k):
            if sizes and all(isinstance(x, int) for x in sizes) and not k:
                return torch.zeros(*sizes, dtype=torch.long)
            return orig_new(t, *sizes, **k)
        torch.Tensor.random_ = fake
        torch.Tensor.new = new_long
        return self

    def __exit__(self, *exc):
        torch.Tensor.random_ = self.orig
        torch.Tensor.new = self.orig_new
        assert exc[0] is not None or not self.queue, "unused draws"


def run_reference_variant(BertConfig, Model, cd, sd, batch, draws=None):
    cfg = BertConfig.from_dict(cd)
    model = Model(cfg)
    model.load_state_dict(sd, strict=True)
    model.eval()
    b = {k: v.clone() for k, v in batch.items()}
    args = (b["input_ids"], b["image_feat"], b["image_loc"], b["segment_ids"], b["input_mask"], b["image_mask"], b["lm_label_ids"],
            b["image_label"], b["image_cls"], b.get("obj_labels"), b.get("obj_confs"), b.get("attr_labels"), b.get("attr_confs"), None, b["is_match"])
    if draws is not None:
        with _FedRandom([draws["row_across"], draws["col_across"], draws["col_inside"]]):
            lm, img, nsp = model(*args)
    else:
        lm, img, nsp = model(*args)
    seq_t, seq_v, pt, pv, _ = model.bert(b["input_ids"], batch["image_feat"].clone(), b["image_loc"], b["segment_ids"], b["input_mask"], b["image_mask"])
    out = dict(loss_lm=lm.detach().numpy(), loss_img=img.detach().numpy(), loss_nsp=nsp.detach().numpy(),
               seq_t=seq_t.detach().numpy(), seq_v=seq_v.detach().numpy())
    if pt is not None:
        out["pooled_t"] = pt.detach().numpy()
    if pv is not None:
        out["pooled_v"] = pv.detach().numpy()
    (lm + img + nsp).sum().backward()
    named = dict(model.named_parameters())
    for k in VARIANT_GRAD_KEYS:
        if k in named and named[k].grad is not None:
            out["grad::" + k] = named[k].grad.numpy().copy()
    out["grad_none"] = np.array(sorted(k for k, p in named.items() if p.grad is None))      # parameters the step must leave alone
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))
    out["grad_norm"] = np.array([float(total)])
    return out, model


def write_variants(BertConfig, Model):
    import json
    for name, cd in variant_configs().items():
        cfg = R.RefConfig(cd)
        sd = R.make_weights(cfg, seed=7)
        B, T, Rn = 3, 6, 7
        seed = 11
        while True:                                    # a batch with a few masked regions left after the objective-1 relabel
            batch = R.synthetic_batch(cfg, B=B, T=T, R=Rn, seed=seed, pad=True)
            if int((batch["image_label"] == 1).sum()) >= 3 and int((batch["lm_label_ids"] != -1).sum()) >= 2:
                break
            seed += 1
        draws = None
        if "2" in cd["visual_target_weights"]:
            draws = R.nce_draws(seed=0x1234ABCD5, site=0, B=B, R=Rn)
        blob_seed = seed
        out, model = run_reference_variant(BertConfig, Model, cd, sd, batch, draws)
        blob = {"cfg_json": np.array(json.dumps(cd)), "ref_keys": np.array(list(model.state_dict().keys())), "batch_seed": np.array([blob_seed])}
        blob.update({"in::" + k: v.numpy() for k, v in batch.items()})
        if draws is not None:
            blob.update({"draw::" + k: v.numpy() for k, v in draws.items()})
        blob["weights_seed"] = np.array([7])             # R.make_weights(cfg, seed=7): the 2048-wide tensors are not stored
        for k, v in out.items():
            if k.startswith("grad::") and v.size > 8192:   # large gradients: a slice and the norm
                blob["out::gradslice::" + k[6:]] = v.reshape(v.shape[0], -1)[:16, :64]
                blob["out::gradnorm::" + k[6:]] = np.array([np.sqrt((v.astype(np.float64) ** 2).sum())])
            else:
                blob["out::" + k] = v
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, "losses", out["loss_lm"], out["loss_img"], out["loss_nsp"], os.path.getsize(path) // 1024, "KB", "no grad:", list(out["grad_none"]))
    # the two real non-ctrl configs that keep the 768-wide geometry, B=2 (weights from the seed generator, as write_ctrl)
    for name, Rn in (("lxmert", 36), ("vl-bert_base", 36), ("vilbert_base", 36)):
        cd = json.load(open(os.path.join(REF, "config", name + ".json")))
        cfg = R.RefConfig(cd)
        sd = R.make_weights(cfg, seed=3, std=0.03)
        batch = R.synthetic_batch(cfg, B=2, T=20, R=Rn, seed=7)
        out, model = run_reference_variant(BertConfig, Model, cd, sd, batch)
        blob = {"cfg_json": np.array(json.dumps(cd)), "n_params": np.array([sum(p.numel() for p in model.parameters())]),
                "ref_keys": np.array(list(model.state_dict().keys())), "out::grad_none": out["grad_none"]}
        for k in ("loss_lm", "loss_img", "loss_nsp", "pooled_t", "pooled_v", "grad_norm"):
            if k in out:
                blob["out::" + k] = out[k]
        blob["out::seq_t_slice"] = out["seq_t"][:, :, :64]
        blob["out::seq_v_slice"] = out["seq_v"][:, :8, :64]
        for k in VARIANT_GRAD_KEYS[1:]:
            if "grad::" + k in out:
                g = out["grad::" + k]
                blob["out::gradslice::" + k] = g.reshape(g.shape[0], -1)[:16, :64] if g.ndim > 1 else g[:64]
                blob["out::gradnorm::" + k] = np.array([np.sqrt((g.astype(np.float64) ** 2).sum())])
        path = os.path.join(OUT, "full_" + name + ".npz")
        np.savez_compressed(path, **blob)
        print(name, int(blob["n_params"][0]), "losses", out["loss_lm"], out["loss_img"], out["loss_nsp"], os.path.getsize(path) // 1024, "KB")
        del model, sd


def write_fullsize(BertConfig, Model):
    """ctrl_vilbert_base at the BASELINE batch (B=256, T=20, R=36; forward only) and at B=32 (forward + backward): the
    contract check of north_star's "loss matching reference to 1e-3 rel" at the size the bench line is quoted on, and
    gradients at a batch where the ITM path's bf16 noise has averaged out.  Weights and batch come from the seed generators,
    only losses, small slices, checksums and gradient norms / slices are stored (a few KB)."""
    import json
    name = "ctrl_vilbert_base"
    cd = json.load(open(os.path.join(ROOT, "config", name + ".json")))
    cfg = R.RefConfig(cd)
    sd = R.make_weights(cfg, seed=3, std=0.03)
    for B, want_grads in ((32, True), (256, False)):
        batch = R.synthetic_batch(cfg, B=B, T=20, R=36, seed=7)
        with torch.set_grad_enabled(want_grads):
            out, model = run_reference(BertConfig, Model, cd, sd, batch, want_grads=want_grads)
        blob = {"B": np.array([B])}
        for k in ("loss_lm", "loss_img", "loss_nsp", "grad_norm"):
            if k in out:
                blob["out::" + k] = out[k]
        blob["out::pooled_t_slice"] = out["pooled_t"][:, :64]
        blob["out::pooled_v_slice"] = out["pooled_v"][:, :64]
        blob["out::seq_t_slice"] = out["seq_t"][::16, :, :64]
        blob["out::seq_v_slice"] = out["seq_v"][::16, :8, :64]
        blob["out::seq_t_sum"] = np.array([out["seq_t"].astype(np.float64).sum()])
        blob["out::seq_v_sum"] = np.array([out["seq_v"].astype(np.float64).sum()])
        blob["out::seq_t_abs"] = np.array([np.abs(out["seq_t"].astype(np.float64)).sum()])
        blob["out::seq_v_abs"] = np.array([np.abs(out["seq_v"].astype(np.float64)).sum()])
        for k in GRAD_KEYS[1:]:
            if "grad::" + k in out:
                g = out["grad::" + k]
                blob["out::gradslice::" + k] = g.reshape(g.shape[0], -1)[:16, :64] if g.ndim > 1 else g[:64]
                blob["out::gradnorm::" + k] = np.array([np.sqrt((g.astype(np.float64) ** 2).sum())])
        path = os.path.join(OUT, "%s_b%d.npz" % (name, B))
        np.savez_compressed(path, **blob)
        print(name, "B", B, "losses", out["loss_lm"], out["loss_img"], out["loss_nsp"], os.path.getsize(path) // 1024, "KB", flush=True)
        del model, out


def write_init_stats(BertConfig, Model):
    """What the reference's CONSTRUCTOR leaves in every tensor of the five real ctrl_* models (encoders.py:904-915 normal / zero /
    one init, :753-764 xavier heads, embeddings.py:229-238,328-334,428-431 family-specific copies and zero LayerNorm weights):
    per state_dict key [mean, std, min, max] plus the groups of keys holding identical values (aliases and copy-initialised
    tables).  Statistics, not values: the two code bases draw from the generator in different orders."""
    import json
    for name in ("ctrl_vilbert_base", "ctrl_lxmert", "ctrl_uniter_base", "ctrl_visualbert_base", "ctrl_vl-bert_base"):
        cd = json.load(open(os.path.join(ROOT, "config", name + ".json")))
        torch.manual_seed(1234)
        model = Model(BertConfig.from_dict(cd))
        sd = model.state_dict()
        keys = list(sd.keys())
        stats = np.array([[float(v.double().mean()), float(v.double().std(unbiased=False)), float(v.min()), float(v.max())] for v in sd.values()])
        # groups of equal tensors among same-shaped, non-constant ones
        sig = {}
        for k, v in sd.items():
            if float(v.double().std(unbiased=False)) > 0:
                sig.setdefault((tuple(v.shape), float(v.double().sum()), float(v.reshape(-1)[0])), []).append(k)
        groups = [g for g in sig.values() if len(g) > 1 and all(torch.equal(sd[g[0]], sd[k]) for k in g)]
        path = os.path.join(OUT, "init_" + name + ".npz")
        np.savez_compressed(path, keys=np.array(keys), stats=stats, shapes=np.array([json.dumps(list(v.shape)) for v in sd.values()]),
                            equal_groups=np.array(json.dumps(groups)))
        print("init", name, len(keys), "keys,", len(groups), "groups of equal tensors", os.path.getsize(path) // 1024, "KB", flush=True)
        del model


def write_hf_remap(BertConfig, Model):
    """What the reference's from_pretrained(..., from_hf=True) (volta/utils.py:458-498) makes of a HuggingFace-layout
    BERT checkpoint: per model key the checksum of the tensor that landed there + the loader's missing / unexpected lists."""
    import json, tempfile
    cd = dict(tiny_configs()["tiny_vilbert"], bert_layer2attn_sublayer={"0": 0, "1": 2, "2": 6}, bert_layer2ff_sublayer={"0": 1, "1": 3, "2": 7})
    for tag, with_prefix in (("hf_remap_prefixed", True), ("hf_remap_bare", False)):
        cfg = R.RefConfig(cd)
        hf = R.hf_style_bert_state_dict(cfg, n_layers=3, seed=5, with_prefix=with_prefix)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "pytorch_model.bin")
            torch.save(hf, path)
            torch.manual_seed(0)
            model, info = Model.from_pretrained(path, config=BertConfig.from_dict(cd), from_hf=True, output_loading_info=True, default_gpu=False)
        sd = model.state_dict()
        keys = list(sd.keys())
        miss = set(info["missing_keys"])
        if with_prefix:
            loaded = [k for k in keys if k not in miss]
        else:       # only model.bert was loaded (utils.py:513-516): the loader's key lists are relative to it
            loaded = [k for k in keys if k.startswith("bert.") and k[len("bert."):] not in miss]
        blob = {"cfg_json": np.array(json.dumps(cd)), "with_prefix": np.array([int(with_prefix)]),
                "missing": np.array(sorted(info["missing_keys"])), "unexpected": np.array(sorted(info["unexpected_keys"])),
                "loaded_keys": np.array(loaded),
                "loaded_sum": np.array([float(sd[k].double().sum()) for k in loaded]),
                "loaded_first": np.array([float(sd[k].reshape(-1)[0]) for k in loaded]),
                "training": np.array([int(model.training)])}
        path = os.path.join(OUT, tag + ".npz")
        np.savez_compressed(path, **blob)
        print(tag, len(loaded), "loaded,", len(info["missing_keys"]), "missing,", len(info["unexpected_keys"]), "unexpected", os.path.getsize(path) // 1024, "KB")


TASK_CFG = {"TASK1": {"type": "VL-classifier", "num_labels": 37}, "TASK9": {"type": "V-logit"}, "TASK10": {"type": "V-logit", "num_clf_layers": 2},
            "TASK12": {"type": "VL-binary-classifier"}, "TASK13": {"type": "VL-tri-classifier"}, "TASK8": {"type": "VL-logit"}}


def write_tasks(BertConfig):
    """BertForVLTasks of the real reference (volta/encoders.py:1117-1206) on two tiny configs, every head type, eval mode:
    predictions and the gradients of a few parameters for loss = sum(prediction * probe)."""
    import json
    from volta.encoders import BertForVLTasks
    variants = {"tiny_vilbert": {}, "tiny_uniter": {},
                # the other fusion methods of the task model (encoders.py:1184-1195); vl-bert_vqa pools the token before the caption's end
                "tiny_vilbert_vqa": dict(fusion_method="vl-bert_vqa"), "tiny_vilbert_sum": dict(fusion_method="sum"), "tiny_vilbert_text": dict(fusion_method="text")}
    for name, extra in variants.items():
        cd = dict(tiny_configs()[name[:12] if name.startswith("tiny_vilbert") else name], clf_hidden_size=96, **extra)
        cfg = R.RefConfig(cd)
        ids = list(TASK_CFG) if not extra else ["TASK1", "TASK9", "TASK12"]
        sd = R.make_task_weights(cfg, TASK_CFG, ids, seed=13)
        batch = R.synthetic_batch(cfg, B=4, T=6, R=4, seed=17, pad=True)
        model = BertForVLTasks(BertConfig.from_dict(cd), TASK_CFG, ids)
        model.load_state_dict(sd, strict=True)
        model.eval()
        blob = {"cfg_json": np.array(json.dumps(cd)), "task_cfg_json": np.array(json.dumps({t: TASK_CFG[t] for t in ids})), "ref_keys": np.array(list(model.state_dict().keys()))}
        for t in ids:
            model.zero_grad()
            pred = model(batch["input_ids"], batch["image_feat"].clone(), batch["image_loc"], t, batch["segment_ids"], batch["input_mask"], batch["image_mask"])[0]
            g = torch.Generator().manual_seed(hash(t) % 1000 if False else sum(map(ord, t)))
            probe = torch.randn(pred.shape, generator=g)
            (pred * probe).sum().backward()
            blob["pred::" + t] = pred.detach().numpy()
            named = dict(model.named_parameters())
            for k in ("bert.encoder.layer.0.attention_self.query.weight", "bert.embeddings.word_embeddings.weight", "bert.t_pooler.dense.weight",
                      "bert.v_pooler.dense.bias", [q for q in named if q.startswith("clfs_dict.%s." % t)][0]):
                if k in named and named[k].grad is not None:
                    blob["grad::%s::%s" % (t, k)] = named[k].grad.numpy().copy()
        path = os.path.join(OUT, "tasks_" + name + ".npz")
        np.savez_compressed(path, **blob)
        print("tasks", name, {t: blob["pred::" + t].shape for t in ids}, os.path.getsize(path) // 1024, "KB")


def concap_records(n_pairs, region_len, seed, F=2048, C=1601, ragged=False):
    """Raw per-pair records (what the LMDB rows hold, concept_cap_dataset.py:430-431) from a seed: shared with the tests."""
    rng = np.random.default_rng(seed)
    recs = []
    for b in range(n_pairs):
        # the reference only works with num_boxes == region_len (its `masked_label` / `overlaps` shapes, :645-659, differ otherwise)
        n = int(rng.integers(3, region_len + 1)) if ragged else region_len
        w, h = float(rng.integers(300, 800)), float(rng.integers(300, 800))
        xy = rng.uniform(0, 0.6, (n, 2)) * np.array([w, h])
        wh = rng.uniform(0.1, 0.4, (n, 2)) * np.array([w, h])
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        if n > 3:
            boxes[1] = boxes[0] + rng.uniform(-4, 4, 4).astype(np.float32)          # a heavily overlapping pair (IoU > 0.4)
        cls = rng.random((n, C), dtype=np.float32)
        recs.append(dict(caption_index=b, feat=rng.random((n, F), dtype=np.float32), cls=cls / cls.sum(1, keepdims=True), boxes=boxes, w=w, h=h))
    caps = [[int(t) for t in rng.integers(1000, 3000, int(rng.integers(2, 16)))] for _ in range(n_pairs + 5)]
    return recs, caps


def write_concap():
    """The REAL BertPreprocessBatch.__call__ and ConceptCapLoaderTrain.__iter__ (volta/datasets/concept_cap_dataset.py) on synthetic raw
    records, with their random draws replaced by the producer's word streams (oracle/volta_ref.py:concap_words): pins the oracle's
    restatement of the masking policy, the box normalisation, IoU co-masking and the global-feature row.  The module is loaded by file
    path with empty stand-ins for `tensorpack` / `msgpack_numpy` (LMDB plumbing the called code never touches)."""
    import importlib.util, random as pyrandom
    for name in ("tensorpack", "tensorpack.dataflow", "msgpack_numpy"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["tensorpack"].dataflow = sys.modules["tensorpack.dataflow"]
    sys.modules["msgpack_numpy"].patch = lambda: None
    spec = importlib.util.spec_from_file_location("ref_concap", os.path.join(REF, "volta", "datasets", "concept_cap_dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    T, Rl, V, seed, B = 14, 10, 3000, 77, 8
    recs, caps = concap_records(B, Rl, seed=3)
    w_tok, w_rnd = R.concap_words(seed, R.CC_SITE_TOKEN, B, T), R.concap_words(seed, R.CC_SITE_RANDTOK, B, T)
    w_reg, w_cap = R.concap_words(seed, R.CC_SITE_REGION, B, Rl), R.concap_words(seed, R.CC_SITE_CAPTION, B, 2)

    class Tok:
        mask_token = "[MASK]"
        def encode(self, caption): return list(caption)
        def add_special_tokens_single_sentence(self, tokens): return [101] + list(tokens) + [102]
        def convert_tokens_to_ids(self, tok): return 103
        def __len__(self): return V

    class Draws:      # stands in for the `random` module inside the reference file: replays the word streams in call order
        def __init__(self, b): self.b, self.k = b, 0
        def random(self):
            k, self.k = self.k, self.k + 1
            if k == 0:
                return float(w_cap[self.b, 0]) / 2.0 ** 32
            ntok = self.ntok
            return float(w_tok[self.b, k - 1]) / 2.0 ** 32 if k - 1 < ntok else float(w_reg[self.b, k - 1 - ntok]) / 2.0 ** 32
        def randint(self, lo, hi):
            return int(w_cap[self.b, 1]) % (hi - lo + 1) + lo

    proc = object.__new__(mod.BertPreprocessBatch)
    proc.split, proc.seq_len, proc.region_len, proc.tokenizer, proc.num_caps = "Train", T, Rl, Tok(), len(caps)
    proc.captions, proc.visualization, proc.objective, proc.bert_model, proc.num_locs = caps, False, 0, "bert-base-uncased", 5
    rows = []
    real_randint = np.random.randint
    for b, rec in enumerate(recs):
        n = rec["feat"].shape[0]
        d = Draws(b)
        swapped = float(w_cap[b, 0]) / 2.0 ** 32 > 0.5
        d.ntok = min(len(caps[int(w_cap[b, 1]) % len(caps)] if swapped else caps[rec["caption_index"]]), T - 2)
        mod.random = d
        np.random.randint = lambda hi, _d=d: int(w_rnd[_d.b, _d.k - 2]) % hi        # token index of the draw just made
        try:
            data = (rec["feat"], rec["cls"], np.zeros(n, np.int64), np.zeros(n, np.float32), np.zeros(n, np.int64), np.zeros(n, np.float32),
                    np.zeros((n, 401), np.float32), rec["boxes"].copy(), n, rec["h"], rec["w"], b, caps[rec["caption_index"]])
            rows.append(proc(data))
        finally:
            np.random.randint = real_randint
    mod.random = pyrandom
    batch = [np.stack([np.asarray(r[i]) for r in rows]) for i in range(17)]
    loader = object.__new__(mod.ConceptCapLoaderTrain)
    loader.add_global_imgfeat, loader.num_locs = "first", 5
    loader.ds = types.SimpleNamespace(get_data=lambda: iter([batch]))
    out = next(iter(loader))
    names = ["input_ids", "input_mask", "segment_ids", "lm_label_ids", "is_next", "image_feat", "image_loc", "image_cls", "obj_labels", "obj_confs",
             "attr_labels", "attr_confs", "image_attrs", "image_label", "image_mask"]
    o = {k: v.numpy() for k, v in zip(names, out[:15])}
    blob = {"params": np.array([T, Rl, V, seed, B, 3]), "masked_label": batch[15].astype(np.int8)}
    for k in ("input_ids", "input_mask", "segment_ids", "lm_label_ids", "is_next", "image_loc", "image_label", "image_mask"):
        blob[k] = o[k]
    blob["image_feat_rowsum"] = o["image_feat"].astype(np.float64).sum(2)
    blob["image_feat_global_head"] = o["image_feat"][:, 0, :128]
    blob["image_cls_rowsum"] = o["image_cls"].astype(np.float64).sum(2)
    path = os.path.join(OUT, "concap_batch.npz")
    np.savez_compressed(path, **blob)
    print("concap", {k: v.shape for k, v in blob.items()}, "swapped", o["is_next"].tolist(), "masked tokens", int((o["lm_label_ids"] != -1).sum()),
          "masked regions", int((o["image_label"] == 1).sum()), os.path.getsize(path) // 1024, "KB")



def feature_store_records(seed=11, n_images=4, F=16):
    """Per-image records of a task feature store (what data/*/convert_*_lmdb.py pickles: the TSV row, every field a string): shared with the tests."""
    import base64
    rng = np.random.default_rng(seed)
    recs = {}
    for i in range(n_images):
        n = int(rng.integers(2, 9))
        w, h = int(rng.integers(300, 800)), int(rng.integers(300, 800))
        xy = rng.uniform(0, 0.6, (n, 2)) * np.array([w, h])
        wh = rng.uniform(0.1, 0.4, (n, 2)) * np.array([w, h])
        boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        feats = rng.standard_normal((n, F)).astype(np.float32)
        recs[str(1000 + 37 * i)] = dict(img_id=str(1000 + 37 * i), img_h=str(h), img_w=str(w), num_boxes=str(n),
                                        boxes=base64.b64encode(boxes.tobytes()).decode(), features=base64.b64encode(feats.tobytes()).decode())
    return recs


def write_feature_reader():
    """The REAL ImageFeaturesH5Reader.__getitem__ (volta/datasets/_image_features_reader.py:69-196) on a small synthetic store, served by a
    dict-backed stand-in for the `lmdb` module (the container; `vk_lmdb_*` replaces it and is tested on files written by tests/lmdb_writer.py).
    Pins the box normalisation, the area column, the global feature / box rows and the dtypes numpy's promotions leave behind."""
    import importlib.util, pickle
    recs = feature_store_records()
    store = {k.encode(): pickle.dumps(v) for k, v in recs.items()}
    store[b"keys"] = pickle.dumps([k.encode() for k in recs])

    class Txn:
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def get(self, key): return store.get(key)

    fake = types.ModuleType("lmdb")
    fake.open = lambda *a, **k: types.SimpleNamespace(begin=lambda write=False: Txn())
    saved = sys.modules.get("lmdb")
    sys.modules["lmdb"] = fake
    try:
        spec = importlib.util.spec_from_file_location("ref_feature_reader", os.path.join(REF, "volta", "datasets", "_image_features_reader.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if saved is None:
            del sys.modules["lmdb"]
        else:
            sys.modules["lmdb"] = saved
    blob = {}
    for nl in (4, 5):
        for glob in (None, "first", "last"):
            for mem in (False, True):
                cfg = types.SimpleNamespace(v_feature_size=16, num_locs=nl, add_global_imgfeat=glob)
                rd = mod.ImageFeaturesH5Reader("unused", cfg, in_memory=mem)
                assert len(rd) == len(recs)
                for k in recs:
                    f, n, loc, ori = rd[k]
                    if mem:
                        f, n, loc, ori = rd[k]          # second read comes from the in-memory copy
                    tag = "l%d_%s_%d_%s" % (nl, glob, mem, k)
                    blob[tag + "_features"], blob[tag + "_num"], blob[tag + "_loc"], blob[tag + "_ori"] = f, np.array(n), np.asarray(loc), np.asarray(ori)
    path = os.path.join(OUT, "feature_reader.npz")
    np.savez_compressed(path, **blob)
    print("feature_reader", len(blob), "arrays", {str(v.dtype) for v in blob.values()}, os.path.getsize(path) // 1024, "KB")


def write_attn_maps(BertConfig, Model):
    """Attention maps of BertModel.forward(output_all_attention_masks=True) under config.visualization (volta/encoders.py:342-358,
    858-886) for the tiny dual-stream and LXMERT-like configs: probabilities, query and key layers of every attention sub-layer."""
    for name in ("tiny_vilbert", "tiny_lxmert", "tiny_gated"):
        cd = dict(tiny_configs()[name], visualization=True)
        cfg = R.RefConfig(cd)
        sd = R.make_weights(cfg, seed=7)
        batch = R.synthetic_batch(cfg, B=3, T=6, R=4, seed=11, pad=True)
        model = Model(BertConfig.from_dict(cd))
        model.load_state_dict(sd, strict=True)
        model.eval()
        with torch.no_grad():
            out = model.bert(batch["input_ids"], batch["image_feat"].clone(), batch["image_loc"], batch["segment_ids"], batch["input_mask"],
                             batch["image_mask"], output_all_attention_masks=True)
        maps_t, maps_v = out[4]
        blob = {"cfg_json": np.array(__import__("json").dumps(cd)), "n_layers": np.array([len(maps_t)])}
        for tag, maps in (("t", maps_t), ("v", maps_v)):
            for i, d in enumerate(maps):
                for key in ("intra_attn", "inter_attn", "queries", "keys"):
                    if d[key] is not None:
                        blob["%s%d::%s" % (tag, i, key)] = d[key].numpy()
        path = os.path.join(OUT, "attn_maps_" + name + ".npz")
        np.savez_compressed(path, **blob)
        print("attn_maps", name, len(maps_t), "attention sub-layers", os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    BertConfig, Model = import_reference()
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "tiny"):
        write_tiny(BertConfig, Model)
    if which in ("all", "ctrl"):
        write_ctrl(BertConfig, Model)
    if which in ("all", "variants"):
        write_variants(BertConfig, Model)
    if which in ("all", "full"):
        write_fullsize(BertConfig, Model)
    if which in ("all", "init"):
        write_init_stats(BertConfig, Model)
    if which in ("all", "hf"):
        write_hf_remap(BertConfig, Model)
    if which in ("all", "tasks"):
        write_tasks(BertConfig)
    if which in ("all", "concap"):
        write_concap()
    if which in ("all", "reader"):
        write_feature_reader()
    if which in ("all", "attn"):
        write_attn_maps(BertConfig, Model)
