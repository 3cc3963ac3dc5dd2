"""Where does the bf16 noise of the ITM loss enter?    CPU study, test infrastructure only.

Runs the oracle's ctrl_vilbert_base forward at the contract size (B=256, weights / batch of tests/test_fullsize_golden_gpu.py) with
bf16 rounding switched on at chosen storage points -- the points at which the HIP engine stores bf16 -- and prints the relative
error of the ITM loss (and of the pooled rows) against the unrounded fp32 run.  Usage:
    python tests/study_itm_noise.py [B]
Not collected by pytest (no test_ prefix)."""
import json
import re
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import volta_ref as R  # noqa: E402

POINTS = ("w", "feat", "emb", "qkv", "p", "ctx", "d", "z32", "y", "h", "pool")


class Q:
    """q(x, point): round to bf16 when the point is enabled."""

    def __init__(self, on, wfilter=None):
        self.on = set(on)
        self.wfilter = wfilter          # regex: only the weights whose name matches are rounded

    def __call__(self, x, point):
        return x.bfloat16().float() if point in self.on else x

    def weight(self, w, name):
        if "w" in self.on and (self.wfilter is None or re.search(self.wfilter, name)):
            return w.bfloat16().float()
        return w


def lin(x, sd, name, q):
    return F.linear(x, q.weight(sd[name + ".weight"], name), sd.get(name + ".bias"))


def attn(sd, cfg, n, t, v, t_mask, v_mask, q):
    p = "bert.encoder.layer.%d." % n
    has_tt, has_tv = n in cfg.tt_attn_sublayers, n in cfg.tv_attn_sublayers
    has_vt, has_vv = n in cfg.vt_attn_sublayers, n in cfg.vv_attn_sublayers
    has_t, has_v = has_tt or has_tv, has_vv or has_vt
    a = p + "attention_self."
    nh = cfg.num_attention_heads
    if has_t:
        tq, tk, tv_ = (R._heads(q(lin(t, sd, a + k, q), "qkv"), nh) for k in ("query", "key", "value"))
    if has_v:
        vq, vk, vv_ = (R._heads(q(lin(v, sd, a + "v_" + k, q), "qkv"), nh) for k in ("query", "key", "value"))

    def scores(qq, k, mask):
        return qq @ k.transpose(-1, -2) / math.sqrt(qq.shape[-1]) + mask

    def ctx_of(blocks):
        probs = torch.softmax(torch.cat([b[0] for b in blocks], -1), -1).split([b[0].shape[-1] for b in blocks], -1)
        return q(sum(R._merge(q(pr, "p") @ b[1]) for pr, b in zip(probs, blocks)), "ctx")

    o = p + "attention_output."
    t_out, v_out = t, v
    if has_t:
        blocks = ([(scores(tq, tk, t_mask), tv_)] if has_tt else []) + ([(scores(tq, vk, v_mask), vv_)] if has_tv else [])
        d = q(lin(ctx_of(blocks), sd, o + "dense", q), "d")
        t_out = q(R.layer_norm(d + t, sd[o + "LayerNorm.weight"], sd[o + "LayerNorm.bias"]), "y")
    if has_v:
        blocks = ([(scores(vq, tk, t_mask), tv_)] if has_vt else []) + ([(scores(vq, vk, v_mask), vv_)] if has_vv else [])
        d = q(lin(ctx_of(blocks), sd, o + "v_dense", q), "d")
        v_out = q(R.layer_norm(d + v, sd[o + "v_LayerNorm.weight"], sd[o + "v_LayerNorm.bias"]), "y")
    return t_out, v_out


def ffn(sd, cfg, n, t, v, q):
    p = "bert.encoder.layer.%d." % n
    t_out, v_out = t, v
    o = p + "output."
    if n in cfg.t_ff_sublayers:
        h = q(R.gelu(lin(t, sd, p + "intermediate.dense", q)), "h")
        d = q(lin(h, sd, p + "output.dense", q), "d")
        t_out = q(R.layer_norm(d + t, sd[o + "LayerNorm.weight"], sd[o + "LayerNorm.bias"]), "y")
    if n in cfg.v_ff_sublayers:
        h = q(R.gelu(lin(v, sd, p + "intermediate.v_dense", q)), "h")
        d = q(lin(h, sd, p + "output.v_dense", q), "d")
        v_out = q(R.layer_norm(d + v, sd[o + "v_LayerNorm.weight"], sd[o + "v_LayerNorm.bias"]), "y")
    return t_out, v_out


def forward(sd, cfg, b, on, wfilter=None):
    q = Q(on, wfilter)
    drop = R.Dropper(False)
    t = q(R.emb_text_bert(sd, cfg, b["input_ids"], b["segment_ids"], drop), "emb")
    feat = q(b["image_feat"], "feat")               # the engine casts the region features to bf16 for the projection GEMM
    e = F.linear(feat, q.weight(sd["bert.v_embeddings.image_embeddings.weight"], "bert.v_embeddings.image_embeddings"), sd["bert.v_embeddings.image_embeddings.bias"]) + \
        F.linear(b["image_loc"], sd["bert.v_embeddings.image_location_embeddings.weight"], sd["bert.v_embeddings.image_location_embeddings.bias"])
    v = q(R.layer_norm(e, sd["bert.v_embeddings.LayerNorm.weight"], sd["bert.v_embeddings.LayerNorm.bias"]), "emb")
    t_mask = (1.0 - b["input_mask"][:, None, None, :].float()) * -10000.0
    v_mask = (1.0 - b["image_mask"][:, None, None, :].float()) * -10000.0
    for n, typ in R.sublayer_schedule(cfg):
        t, v = attn(sd, cfg, n, t, v, t_mask, v_mask, q) if typ == "attn" else ffn(sd, cfg, n, t, v, q)
    pt = q(torch.relu(lin(t[:, 0], sd, "bert.t_pooler.dense", q)), "pool")
    pv = q(torch.relu(lin(v[:, 0], sd, "bert.v_pooler.dense", q)), "pool")
    itm = lin(q(pt * pv, "pool"), sd, "cls.bi_seq_relationship", q)
    loss = F.cross_entropy(itm.view(-1, 2), b["is_match"].view(-1))
    return float(loss), t[:, 0], v[:, 0], itm


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    torch.manual_seed(0)
    rcfg = R.RefConfig(json.load(open(os.path.join(ROOT, "config", "ctrl_vilbert_base.json"))))
    sd = R.make_weights(rcfg, seed=int(sys.argv[3]) if len(sys.argv) > 3 else 3, std=0.03)
    b = R.synthetic_batch(rcfg, B=B, T=20, R=36, seed=7)
    t0 = time.time()
    with torch.no_grad():
        ref, rt, rv, ritm = forward(sd, rcfg, b, ())
        print("fp32 ITM loss %.6f  (%.1f s)" % (ref, time.time() - t0))
        rel = lambda a, c: float((a - c).norm() / c.norm())
        sets = [("all", POINTS)] + [("only " + p, (p,)) for p in POINTS if p != "z32"] + [("all but " + p, tuple(x for x in POINTS if x != p)) for p in ("w", "y", "d", "qkv", "h")]
        if len(sys.argv) > 2 and sys.argv[2] == "groups":
            sets = []
            groups = [("pool/itm", r"pooler|bi_seq"), ("img emb", r"v_embeddings"), ("attn qkv", r"attention_self"), ("attn out", r"attention_output"),
                      ("ffn up", r"intermediate"), ("ffn down", r"\.output\."), ("layers 0-11", r"layer\.([0-9]|1[01])\."), ("layers 12-23", r"layer\.(1[2-9]|2[0-3])\."),
                      ("layers 24-35", r"layer\.(2[4-9]|3[0-5])\."), ("text side 12+", r"layer\.(1[2-9]|[23][0-9])\.[a-z_]*\.(?!v_)"), ("vision side", r"\.v_")]
            for gname, pat in groups:
                l, t, v, itm = forward(sd, rcfg, b, ("w",), pat)
                print("w only %-14s ITM rel %.2e   row0 t %.2e  v %.2e   logit-diff shift mean %.2e rms %.2e" % (
                    gname, abs(l - ref) / ref, rel(t, rt), rel(v, rv), float(((itm - ritm)[:, 1] - (itm - ritm)[:, 0]).mean()),
                    float(((itm - ritm)[:, 1] - (itm - ritm)[:, 0]).pow(2).mean().sqrt())))
        for name, on in sets:
            l, t, v, itm = forward(sd, rcfg, b, on)
            print("%-14s ITM rel %.2e   row0 t %.2e  v %.2e   logits rms %.2e (|logit| rms %.2e)" % (
                name, abs(l - ref) / ref, rel(t, rt), rel(v, rv), float((itm - ritm).pow(2).mean().sqrt()), float(ritm.pow(2).mean().sqrt())))


if __name__ == "__main__":
    main()
