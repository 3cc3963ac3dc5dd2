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
    for name in ("tiny_vilbert", "tiny_uniter"):
        cd = dict(tiny_configs()[name], clf_hidden_size=96)
        cfg = R.RefConfig(cd)
        ids = list(TASK_CFG)
        sd = R.make_task_weights(cfg, TASK_CFG, ids, seed=13)
        batch = R.synthetic_batch(cfg, B=4, T=6, R=4, seed=17, pad=True)
        model = BertForVLTasks(BertConfig.from_dict(cd), TASK_CFG, ids)
        model.load_state_dict(sd, strict=True)
        model.eval()
        blob = {"cfg_json": np.array(json.dumps(cd)), "task_cfg_json": np.array(json.dumps(TASK_CFG)), "ref_keys": np.array(list(model.state_dict().keys()))}
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
                if named[k].grad is not None:
                    blob["grad::%s::%s" % (t, k)] = named[k].grad.numpy().copy()
        path = os.path.join(OUT, "tasks_" + name + ".npz")
        np.savez_compressed(path, **blob)
        print("tasks", name, {t: blob["pred::" + t].shape for t in ids}, os.path.getsize(path) // 1024, "KB")


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
    if which in ("all", "hf"):
        write_hf_remap(BertConfig, Model)
    if which in ("all", "tasks"):
        write_tasks(BertConfig)
