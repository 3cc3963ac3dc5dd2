"""CPU oracle for the VOLTA pre-training step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch, functional (no nn.Module) fp32 restatement in stock PyTorch CPU ops of the
reference's `BertForVLPreTraining` forward pass, its three losses, gradient clipping and the
AdamW / warm-up-linear schedule used by the reference's pre-training driver.  It consumes a plain
``state_dict`` whose keys are the reference's own parameter names, so the same weights can be fed
to the imported reference (oracle/make_golden.py, build container only), to this oracle, and to the
HIP engine in ``volta_amd``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product path (``volta_amd``) never does and fails loudly without its HIP library.

Pinning: volta ships no tests for this path, so the oracle is pinned by outputs of
the reference itself run in the build container: ``tests/golden/*.npz`` written by
``oracle/make_golden.py`` (script committed, reference never copied).  The AdamW arithmetic lives in
the third-party ``pytorch-transformers==1.1.0`` which is absent from /root/reference and from this
image: that single function is "parity unpinned" and follows the published formula (see `adamw_step`).

Reference sites restated (paths relative to /root/reference):
  layer norm            volta/encoders.py:48-61      gelu                 volta/encoders.py:130-136
  text embeddings       volta/embeddings.py:55-70    vilbert image emb    volta/embeddings.py:139-146
  lxmert image emb      volta/embeddings.py:162-172  vl-bert emb          volta/embeddings.py:102-124,240-301
  visualbert emb        volta/embeddings.py:336-398  uniter emb           volta/embeddings.py:433-457
  gated attention       volta/encoders.py:228-358    attention output     volta/encoders.py:398-424
  gated intermediate    volta/encoders.py:486-501    gated output         volta/encoders.py:541-566
  encoder loop          volta/encoders.py:820-888    poolers              volta/encoders.py:596-637
  model fwd / masks     volta/encoders.py:954-1017   heads                volta/encoders.py:643-784
  losses                volta/encoders.py:1079-1109, volta/losses.py:16-22
  objective-1 relabel   train_concap.py:279-284      clip / step          train_concap.py:303-312
"""
import json
import math

import numpy as np
import torch
import torch.nn.functional as F

LN_EPS = 1e-12


# --------------------------------------------------------------------------------------- config
class RefConfig(dict):
    """Attribute bag with the defaults of volta/config.py:15-65 (defaults first, JSON overwrites)."""

    DEFAULTS = dict(
        vocab_size=-1, hidden_size=768, num_attention_heads=12, intermediate_size=3072, pooler_size=768,
        hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
        max_position_embeddings=512, type_vocab_size=2, num_locs=5, v_coordinate_embeddings_dim=None,
        add_global_imgfeat=None, image_embeddings="vilbert", initializer_range=0.02, v_feature_size=2048,
        v_hidden_size=768, v_num_attention_heads=12, v_intermediate_size=3072, v_pooler_size=1024,
        v_attention_probs_dropout_prob=0.1, v_hidden_act="gelu", v_hidden_dropout_prob=0.1,
        v_initializer_range=0.2, visual_target_weights={"0": 1}, fixed_layers=[], fusion_method="mul",
        objective=0, clf_hidden_size=1536, image_head_ln=True, model="bert", visualization=False,
        tt_attn_sublayers=[], tv_attn_sublayers=[], vt_attn_sublayers=[], vv_attn_sublayers=[],
        t_ff_sublayers=[], v_ff_sublayers=[], shared_sublayers=[], single_ln_sublayers=[],
        sublayer2attn_hidden_size={}, sublayer2num_attention_heads={}, sublayer2intermediate_size={},
        sublayer2v_attn_hidden_size={}, sublayer2v_num_attention_heads={}, sublayer2v_intermediate_size={},
        bert_layer2attn_sublayer={}, bert_layer2ff_sublayer={},
    )

    def __init__(self, d=None):
        super().__init__(json.loads(json.dumps(self.DEFAULTS)))
        if d:
            self.update(d)

    __getattr__ = dict.__getitem__

    @classmethod
    def from_json_file(cls, path):
        with open(path, "r", encoding="utf-8") as f:
            return cls(json.load(f))


def sublayer_schedule(cfg):
    """[(n, 'attn'|'ff')] in execution order (encoders.py:830-845)."""
    attn = set(cfg.tt_attn_sublayers) | set(cfg.tv_attn_sublayers) | set(cfg.vt_attn_sublayers) | set(cfg.vv_attn_sublayers)
    ff = set(cfg.t_ff_sublayers) | set(cfg.v_ff_sublayers)
    assert not (attn & ff), "Overlapping attn-ff sublayer numbers"
    nums = sorted(attn | ff)
    assert nums == list(range(len(nums))), "Non contiguous sublayer numbers"
    return [(n, "attn" if n in attn else "ff") for n in nums]


# --------------------------------------------------------------------------------------- primitives
def layer_norm(x, w, b):
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return w * ((x - u) / torch.sqrt(s + LN_EPS)) + b


def gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def linear(x, sd, name):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


class Dropper:
    """Dropout provider.  eval: identity.  train: either torch RNG (statistical tests / CPU timing) or
    the counter-based Philox stream of the HIP engine (`volta_amd/csrc/common.h`) so that a training
    step can be replayed bit-for-bit in its random choices (mask replay)."""

    def __init__(self, train=False, philox_seed=None):
        self.train = train
        self.seed = philox_seed
        self.site = 0

    def __call__(self, x, p):
        site = self.site
        self.site += 1
        if not self.train or p <= 0.0:
            return x
        if self.seed is None:
            return F.dropout(x, p, True)
        keep = philox_keep_mask(self.seed, site, tuple(x.shape), p)
        return x * keep.to(x.dtype) * (1.0 / (1.0 - p))


# Philox-4x32 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3").  Engine contract
# (volta_amd/csrc/common.h): a dropout site views its tensor as [rows, C] (C = last dim); element
# (row, c) takes word c&3 of Philox-4x32-7(counter=(c>>2, row, site, 0), key=(seed_lo, seed_hi)) and is
# kept iff word >= floor(p*2^32).  7 rounds (the smallest Crush-resistant variant) for the dropout streams,
# the default 10 for the batch producer's sampling policy and the known-answer vectors.
DROPOUT_PHILOX_ROUNDS = 7
_PH_M0, _PH_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PH_W0, _PH_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox_raw(c0, c1, c2, c3, k0, k1, rounds=10):
    """Vectorised Philox4x32-<rounds>: uint32 arrays (same shape) -> uint32[..., 4]."""
    c0, c1, c2, c3 = (np.asarray(c, np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(rounds):
            p0 = _PH_M0 * c0.astype(np.uint64)
            p1 = _PH_M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + _PH_W0)
            k1 = np.uint32(k1 + _PH_W1)
    return np.stack([c0, c1, c2, c3], -1)


def philox_u32(seed, site, rows, C):
    """uint32[rows, C]: the engine's random words for dropout site `site` on a [rows, C] tensor."""
    nb = (C + 3) // 4
    c0 = np.broadcast_to(np.arange(nb, dtype=np.uint32)[None, :], (rows, nb))
    c1 = np.broadcast_to(np.arange(rows, dtype=np.uint32)[:, None], (rows, nb))
    w = philox_raw(c0, c1, np.full((rows, nb), site, np.uint32), np.zeros((rows, nb), np.uint32),
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, rounds=DROPOUT_PHILOX_ROUNDS)
    return w.reshape(rows, nb * 4)[:, :C]


def philox_keep_mask(seed, site, shape, p):
    C = int(shape[-1])
    rows = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    u = philox_u32(seed, site, rows, C)
    thr = np.uint32(min(int(p * 4294967296.0), 0xFFFFFFFF))
    return torch.from_numpy(u >= thr).view(*shape)


# --------------------------------------------------------------------------------------- embeddings
def _text_sum(sd, pre, ids, type_ids):
    T = ids.shape[1]
    pos = torch.arange(T, dtype=torch.long)
    return (sd[pre + "word_embeddings.weight"][ids] + sd[pre + "position_embeddings.weight"][pos][None]
            + sd[pre + "token_type_embeddings.weight"][type_ids])


def emb_text_bert(sd, cfg, ids, type_ids, drop):
    pre = "bert.embeddings."
    e = layer_norm(_text_sum(sd, pre, ids, type_ids), sd[pre + "LayerNorm.weight"], sd[pre + "LayerNorm.bias"])
    return drop(e, cfg.hidden_dropout_prob)


def emb_image_vilbert(sd, cfg, feat, loc, drop):
    pre = "bert.v_embeddings."
    e = linear(feat, sd, pre + "image_embeddings") + linear(loc, sd, pre + "image_location_embeddings")
    e = layer_norm(e, sd[pre + "LayerNorm.weight"], sd[pre + "LayerNorm.bias"])
    return drop(e, cfg.v_hidden_dropout_prob)


def emb_image_lxmert(sd, cfg, feat, loc, drop):
    pre = "bert.v_embeddings."
    a = layer_norm(linear(feat, sd, pre + "image_embeddings"), sd[pre + "ImgLayerNorm.weight"], sd[pre + "ImgLayerNorm.bias"])
    b = layer_norm(linear(loc, sd, pre + "image_location_embeddings"), sd[pre + "LocLayerNorm.weight"], sd[pre + "LocLayerNorm.bias"])
    return drop((a + b) / 2, cfg.v_hidden_dropout_prob)


def emb_uniter(sd, cfg, ids, feat, loc, type_ids, drop):
    pre = "bert.embeddings."
    t = layer_norm(_text_sum(sd, pre, ids, type_ids), sd[pre + "LayerNorm.weight"], sd[pre + "LayerNorm.bias"])
    t = drop(t, cfg.hidden_dropout_prob)
    a = layer_norm(linear(feat, sd, pre + "image_embeddings"), sd[pre + "image_layer_norm.weight"], sd[pre + "image_layer_norm.bias"])
    b = layer_norm(linear(loc, sd, pre + "image_location_embeddings"), sd[pre + "image_location_layer_norm.weight"], sd[pre + "image_location_layer_norm.bias"])
    v = a + b + sd[pre + "token_type_embeddings.weight"][1]
    v = layer_norm(v, sd[pre + "v_LayerNorm.weight"], sd[pre + "v_LayerNorm.bias"])
    return t, drop(v, cfg.hidden_dropout_prob)


def emb_visualbert(sd, cfg, ids, feat, loc, type_ids, drop):
    pre = "bert.embeddings."
    T = ids.shape[1]
    t = _text_sum(sd, pre, ids, type_ids)
    v = (linear(feat, sd, pre + "projection") + sd[pre + "position_embeddings_visual.weight"][0]
         + sd[pre + "token_type_embeddings_visual.weight"][1])
    x = layer_norm(torch.cat([t, v], 1), sd[pre + "LayerNorm.weight"], sd[pre + "LayerNorm.bias"])
    x = drop(x, cfg.hidden_dropout_prob)
    return x[:, :T], x[:, T:]


def coordinate_embeddings(boxes, dim):
    x1, y1, x2, y2 = boxes[..., 0], boxes[..., 1], boxes[..., 2], boxes[..., 3]
    pos = torch.stack([(x1 + x2) / 2 * 100, (y1 + y2) / 2 * 100, (x2 - x1) * 100, (y2 - y1) * 100], -1)
    freq = 1000 ** (torch.arange(dim, dtype=boxes.dtype) / float(dim))
    ang = pos[..., None] / freq
    return torch.cat([ang.sin(), ang.cos()], -1)  # [B,K,4,2*dim]


def emb_vlbert(sd, cfg, ids, feat, loc, type_ids, drop):
    """NOTE: like the reference (embeddings.py:244) all-zero feature rows are replaced by the learned
    mask embedding; unlike it the caller's tensor is not mutated."""
    pre = "bert.embeddings."
    B, K, _ = feat.shape
    T = ids.shape[1]
    zero_rows = (feat == 0).all(-1)
    feat = torch.where(zero_rows[..., None], sd[pre + "object_mask_visual_embedding.weight"][0], feat)
    coord = coordinate_embeddings(loc, cfg.v_coordinate_embeddings_dim).reshape(B * K, -1)
    x = torch.cat([coord, feat.reshape(B * K, -1)], -1)
    x = drop(x, cfg.v_attention_probs_dropout_prob)
    final = torch.relu(linear(x, sd, pre + "obj_downsample.1")).view(B, K, -1)
    obj_vis = layer_norm(final, sd[pre + "visual_ln_object.weight"], sd[pre + "visual_ln_object.bias"])
    obj_ling = sd[pre + "object_linguistic_embeddings.weight"][0].expand(B, K, -1).clone()
    if cfg.visual_target_weights.get("6", 0) > 0:      # with the xent_1601 target, masked regions get their own word (embeddings.py:191,262-263)
        obj_ling = torch.where(zero_rows[..., None], sd[pre + "object_mask_word_embedding.weight"][0], obj_ling)
    obj_ling[:, -1] = sd[pre + "end_embedding.weight"][0]
    obj = obj_ling + obj_vis
    txt_vis = layer_norm(final[:, -1:].expand(B, T, -1), sd[pre + "visual_ln_text.weight"], sd[pre + "visual_ln_text.bias"])
    txt = sd[pre + "word_embeddings.weight"][ids] + txt_vis
    text_end = (ids != 0).sum(1, keepdim=True)
    # Reference quirk kept (embeddings.py:285-287): the masked "+= num_boxes" is applied through an
    # EXPANDED (stride-0) view, so the shift lands in the single shared row: every caption gets
    # position t+K wherever ANY caption of the batch has t >= its own length.
    ar = torch.arange(T, dtype=torch.long)
    shifted = (ar[None] >= text_end).any(0)
    tpos = (ar + K * shifted.long())[None].expand(B, T)
    opos = text_end.expand(B, K).clone()
    opos[:, -1] += 1
    ttab, ptab = sd[pre + "token_type_embeddings.weight"], sd[pre + "position_embeddings.weight"]
    t = txt + ptab[tpos] + ttab[type_ids]
    v = obj + ptab[opos] + ttab[2]
    x = layer_norm(torch.cat([t, v], 1), sd[pre + "LayerNorm.weight"], sd[pre + "LayerNorm.bias"])
    x = drop(x, cfg.hidden_dropout_prob)
    return x[:, :T], x[:, T:]


# --------------------------------------------------------------------------------------- encoder
def _heads(x, nh):
    B, L, H = x.shape
    return x.view(B, L, nh, H // nh).transpose(1, 2)


def _merge(x):
    B, nh, L, dh = x.shape
    return x.transpose(1, 2).reshape(B, L, nh * dh)


def gated_attention(sd, cfg, n, t, v, t_mask, v_mask, drop, maps=None):
    """One attention sub-layer: projections, up to four score blocks, joint softmax per query
    modality, per-block dropout, summed contexts, output dense + dropout + residual + LN."""
    p = "bert.encoder.layer.%d." % n
    has_tt, has_tv = n in cfg.tt_attn_sublayers, n in cfg.tv_attn_sublayers
    has_vt, has_vv = n in cfg.vt_attn_sublayers, n in cfg.vv_attn_sublayers
    has_t, has_v = has_tt or has_tv, has_vv or has_vt
    shared = n in cfg.shared_sublayers and has_t and has_v
    single_ln = n in cfg.single_ln_sublayers
    nh = cfg.sublayer2num_attention_heads.get(str(n), cfg.num_attention_heads)
    v_nh = cfg.sublayer2v_num_attention_heads.get(str(n), cfg.v_num_attention_heads)
    vp = "" if shared else "v_"
    a = p + "attention_self."
    if has_t:
        tq, tk, tv_ = (_heads(linear(t, sd, a + k), nh) for k in ("query", "key", "value"))
    if has_v:
        vq, vk, vv_ = (_heads(linear(v, sd, a + vp + k), v_nh) for k in ("query", "key", "value"))

    def scores(q, k, mask):
        return q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]) + mask

    t_ctx = v_ctx = None
    t_data = dict(intra_attn=None, inter_attn=None, queries=tq if has_t else None, keys=tk if has_t else None)      # encoders.py:342-356
    v_data = dict(intra_attn=None, inter_attn=None, queries=vq if has_v else None, keys=vk if has_v else None)
    if has_t:
        blocks = []
        if has_tt:
            blocks.append((scores(tq, tk, t_mask), tv_))
        if has_tv:
            blocks.append((scores(tq, vk, v_mask), vv_))
        probs = torch.softmax(torch.cat([b[0] for b in blocks], -1), -1).split([b[0].shape[-1] for b in blocks], -1)
        dropped_t = [drop(pr, cfg.attention_probs_dropout_prob) for pr in probs]
        t_ctx = sum(_merge(pr @ b[1]) for pr, b in zip(dropped_t, blocks))
        it = iter(dropped_t)
        if has_tt:
            t_data["intra_attn"] = next(it)
        if has_tv:
            t_data["inter_attn"] = next(it)
    if has_v:
        blocks = []
        if has_vt:
            blocks.append((scores(vq, tk, t_mask), tv_))
        if has_vv:
            blocks.append((scores(vq, vk, v_mask), vv_))
        probs = torch.softmax(torch.cat([b[0] for b in blocks], -1), -1).split([b[0].shape[-1] for b in blocks], -1)
        # reference draws the vv mask before the vt mask (encoders.py:309-310); irrelevant in eval
        dropped = [drop(pr, cfg.v_attention_probs_dropout_prob) for pr in reversed(probs)][::-1]
        v_ctx = sum(_merge(pr @ b[1]) for pr, b in zip(dropped, blocks))
        it = iter(dropped)
        if has_vt:
            v_data["inter_attn"] = next(it)
        if has_vv:
            v_data["intra_attn"] = next(it)
    if maps is not None:
        maps[0].append(t_data)
        maps[1].append(v_data)

    o = p + "attention_output."
    t_out, v_out = t, v
    if has_t:
        t_h = drop(linear(t_ctx, sd, o + "dense"), cfg.hidden_dropout_prob)
    if has_v:
        v_h = drop(linear(v_ctx, sd, o + vp + "dense"), cfg.v_hidden_dropout_prob)
    if single_ln:
        T = t.shape[1]
        x = layer_norm(torch.cat([t_h, v_h], 1) + torch.cat([t, v], 1), sd[o + "LayerNorm.weight"], sd[o + "LayerNorm.bias"])
        return x[:, :T], x[:, T:]
    if has_t:
        t_out = layer_norm(t_h + t, sd[o + "LayerNorm.weight"], sd[o + "LayerNorm.bias"])
    if has_v:
        v_out = layer_norm(v_h + v, sd[o + vp + "LayerNorm.weight"], sd[o + vp + "LayerNorm.bias"])
    return t_out, v_out


def gated_ffn(sd, cfg, n, t, v, drop):
    p = "bert.encoder.layer.%d." % n
    has_t, has_v = n in cfg.t_ff_sublayers, n in cfg.v_ff_sublayers
    shared = n in cfg.shared_sublayers and has_t and has_v
    single_ln = n in cfg.single_ln_sublayers
    vp = "" if shared else "v_"
    t_out, v_out = t, v
    if has_t:
        t_h = drop(linear(gelu(linear(t, sd, p + "intermediate.dense")), sd, p + "output.dense"), cfg.hidden_dropout_prob)
    if has_v:
        v_h = drop(linear(gelu(linear(v, sd, p + "intermediate." + vp + "dense")), sd, p + "output." + vp + "dense"), cfg.v_hidden_dropout_prob)
    o = p + "output."
    if single_ln:
        T = t.shape[1]
        x = layer_norm(torch.cat([t_h, v_h], 1) + torch.cat([t, v], 1), sd[o + "LayerNorm.weight"], sd[o + "LayerNorm.bias"])
        return x[:, :T], x[:, T:]
    if has_t:
        t_out = layer_norm(t_h + t, sd[o + "LayerNorm.weight"], sd[o + "LayerNorm.bias"])
    if has_v:
        v_out = layer_norm(v_h + v, sd[o + vp + "LayerNorm.weight"], sd[o + vp + "LayerNorm.bias"])
    return t_out, v_out


def bert_model(sd, cfg, input_ids, image_feat, image_loc, token_type_ids=None, attention_mask=None,
               image_attention_mask=None, drop=None, taps=None):
    """-> (seq_t [B,T,H], seq_v [B,Rv,Hv], pooled_t, pooled_v).  `taps`: optional dict that receives
    the hidden states after the embeddings and after every sub-layer (for layer-by-layer parity)."""
    drop = drop or Dropper(False)
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    if image_attention_mask is None:
        image_attention_mask = torch.ones(image_feat.shape[:2], dtype=input_ids.dtype)
    kind = cfg.image_embeddings
    if kind == "vilbert":
        t = emb_text_bert(sd, cfg, input_ids, token_type_ids, drop)
        v = emb_image_vilbert(sd, cfg, image_feat, image_loc, drop)
    elif kind == "lxmert":
        t = emb_text_bert(sd, cfg, input_ids, token_type_ids, drop)
        v = emb_image_lxmert(sd, cfg, image_feat, image_loc, drop)
    elif kind == "uniter":
        t, v = emb_uniter(sd, cfg, input_ids, image_feat, image_loc, token_type_ids, drop)
    elif kind == "visualbert":
        t, v = emb_visualbert(sd, cfg, input_ids, image_feat, image_loc, token_type_ids, drop)
    elif kind == "vl-bert":
        t, v = emb_vlbert(sd, cfg, input_ids, image_feat, image_loc, token_type_ids, drop)
    else:
        raise ValueError(kind)
    t_mask = (1.0 - attention_mask[:, None, None, :].to(t.dtype)) * -10000.0
    v_mask = (1.0 - image_attention_mask[:, None, None, :].to(t.dtype)) * -10000.0
    if taps is not None:
        taps["emb_t"], taps["emb_v"] = t, v
    for n, typ in sublayer_schedule(cfg):
        if typ == "attn":
            t, v = gated_attention(sd, cfg, n, t, v, t_mask, v_mask, drop, maps=taps.setdefault("attn_maps", ([], [])) if taps is not None else None)
        else:
            t, v = gated_ffn(sd, cfg, n, t, v, drop)
        if taps is not None:
            taps["t%d" % n], taps["v%d" % n] = t, v
    # poolers by fusion method (encoders.py:936-947,1005-1011): "none" has neither, "text" / "vl-bert_vqa" no vision pooler;
    # VLBertTextPooler (:610-623) pools the token two places before the end of the caption (text_end = number of non-zero ids)
    fm = cfg.fusion_method
    if fm == "none":
        pooled_t = None
    elif fm == "vl-bert_vqa":
        text_end = (input_ids != 0).sum(1)
        idx = text_end - 2
        if bool((idx < 0).any()):
            raise ValueError("VLBertTextPooler: a caption with fewer than two tokens selects no row (the reference's batch would shrink)")
        pooled_t = torch.relu(linear(t[torch.arange(t.shape[0]), idx], sd, "bert.t_pooler.dense"))
    else:
        pooled_t = torch.relu(linear(t[:, 0], sd, "bert.t_pooler.dense"))
    pooled_v = None if fm in ("none", "text", "vl-bert_vqa") else torch.relu(linear(v[:, 0], sd, "bert.v_pooler.dense"))
    return t, v, pooled_t, pooled_v


def fuse_pooled(cfg, pooled_t, pooled_v, drop=None, p=0.1):
    """BertPreTrainingHeads.forward / BertForVLTasks.forward (encoders.py:766-778,1184-1195)."""
    fm = cfg.fusion_method
    if fm == "sum":
        x = pooled_t + pooled_v
    elif fm == "mul":
        x = pooled_t * pooled_v
    elif fm in ("text", "vl-bert_vqa"):
        x = pooled_t
    elif fm == "none":
        return None
    else:
        raise ValueError("Invalid fusion method: %s" % fm)
    return drop(x, p) if drop is not None else x


def kl_1601(pred, weight, label, target):
    """losses.py:16-22: sum over masked regions of KL(target || softmax(pred)) / max(#masked, 1)."""
    logp = F.log_softmax(pred, dim=2)
    kl = torch.where(target > 0, target * (target.clamp_min(1e-38).log() - logp), torch.zeros_like(logp))
    m = (label == 1)
    return weight * (kl * m[..., None].to(kl.dtype)).sum() / max(int(m.sum()), 1)


# ---- the other visual targets (losses.py:25-126); `label` [B,R] marks masked regions with 1
VIS_TARGET_WIDTH = {"0": 1601, "1": 2048, "2": 2048, "3": 1600, "4": 400, "5": 2048, "6": 1601}      # losses.py:129-137
NCE_ACROSS, NCE_INSIDE = int(128 * 0.7), int(128 * 0.3)      # 89 + 38 negatives per masked region (losses.py:39,45-46)


def mse_2048(pred, weight, label, feat):
    """losses.py:25-33: squared error against the (input) region features, mean over the masked regions' elements."""
    m = (label == 1)
    return weight * (((pred - feat) ** 2) * m[..., None].to(pred.dtype)).sum() / max(int(m.sum()) * pred.shape[2], 1)


def huber_2048(pred, weight, label, feat):
    """losses.py:105-113 (SmoothL1, beta 1)."""
    m = (label == 1)
    d = (pred - feat).abs()
    l = torch.where(d < 1.0, 0.5 * d * d, d - 0.5)
    return weight * (l * m[..., None].to(pred.dtype)).sum() / max(int(m.sum()) * pred.shape[2], 1)


def xent_hard(pred, weight, label, target, conf=None):
    """xent_1600 / xent_400 (losses.py:83-102: per-region cross entropy x detector confidence) and xent_1601 (:116-124, no confidence),
    summed over the masked regions / max(#masked, 1)."""
    l = F.cross_entropy(pred.reshape(-1, pred.shape[-1]), target.reshape(-1), reduction="none")
    if conf is not None:
        l = l * conf.reshape(-1)
    m = (label.reshape(-1) == 1)
    return weight * (l * m.to(l.dtype)).sum() / max(int(m.sum()), 1)


def nce_negative_index(draws, B, R):
    """losses.py:47-69 from the raw draws: draws["row_across"] in [0, B-1), ["col_across"] in [0, R) ([B,R,89]), ["col_inside"] in
    [0, R-1) ([B,R,38]); the fix-ups steer a negative away from the sample's own image (across) / own region (inside).
    -> flat indices [B, R, 127] into the [B*R, F] feature matrix."""
    row = draws["row_across"].clone()
    for i in range(B - 1):
        row[i][row[i] == i] = B - 1
    across = row * R + draws["col_across"]
    col = draws["col_inside"].clone()
    for i in range(R - 1):
        c = col[:, i, :]
        c[c == i] = R - 1
    inside = torch.arange(B)[:, None, None] * R + col
    return torch.cat([across, inside], 2)


def nce_draws(seed, site, B, R):
    """The product's negatives: word k of region (b, r) = Philox-4x32-10 word (k & 3) at counter (k >> 2, b * R + r, site, 0), reduced
    modulo the range (torch's random_(0, n) is the same reduction of a 32/64-bit word)."""
    n = NCE_ACROSS * 2 + NCE_INSIDE
    c = np.arange((n + 3) // 4, dtype=np.uint32)
    rows = np.arange(B * R, dtype=np.uint32)
    w = philox_raw(np.broadcast_to(c[None, :], (B * R, len(c))).copy(), np.broadcast_to(rows[:, None], (B * R, len(c))).copy(),
                   np.full((B * R, len(c)), site, np.uint32), np.zeros((B * R, len(c)), np.uint32),
                   np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    words = w.reshape(B * R, -1)[:, :n].astype(np.int64).reshape(B, R, n)
    words = torch.from_numpy(words)
    return dict(row_across=words[..., :NCE_ACROSS] % max(B - 1, 1), col_across=words[..., NCE_ACROSS:2 * NCE_ACROSS] % R,
                col_inside=words[..., 2 * NCE_ACROSS:] % max(R - 1, 1))


def nce_2048(pred, weight, label, feat, neg_index):
    """losses.py:36-80: cross entropy (target 0) of <sample_j, prediction> over [own feature, 127 negatives], mean over the masked regions."""
    m = (label == 1)
    predict = pred[m]
    flat = feat.reshape(-1, feat.shape[-1])
    sample = torch.cat([feat[m].unsqueeze(1), flat[neg_index[m]]], 1)
    score = torch.bmm(sample, predict.unsqueeze(2)).squeeze(2)
    return weight * F.cross_entropy(score, torch.zeros(score.shape[0], dtype=torch.long))


def visual_losses(cfg, scores_v, image_label, image_cls=None, image_feat=None, obj_labels=None, obj_confs=None, attr_labels=None,
                  attr_confs=None, nce_index=None):
    """encoders.py:1079-1087: sum of the configured targets' losses; a target whose inputs are missing contributes 0."""
    total = torch.zeros(())
    for ix, w in cfg.visual_target_weights.items():
        if not w > 0:        # (the reference has no decoder for such an entry and fails on it, encoders.py:1080-1084)
            continue
        sv = scores_v[ix]
        sv = sv[:, :-1] if cfg.add_global_imgfeat == "last" else sv[:, int(cfg.add_global_imgfeat is not None):]
        if ix == "0" and image_cls is not None:
            total = total + kl_1601(sv, w, image_label, image_cls)
        elif ix == "1" and image_feat is not None:
            total = total + mse_2048(sv, w, image_label, image_feat)
        elif ix == "2" and image_feat is not None:
            total = total + nce_2048(sv, w, image_label, image_feat, nce_index)
        elif ix == "3" and obj_labels is not None and obj_confs is not None:
            total = total + xent_hard(sv, w, image_label, obj_labels, obj_confs)
        elif ix == "4" and attr_labels is not None and attr_confs is not None:
            total = total + xent_hard(sv, w, image_label, attr_labels, attr_confs)
        elif ix == "5" and image_feat is not None:
            total = total + huber_2048(sv, w, image_label, image_feat)
        elif ix == "6" and obj_labels is not None:
            total = total + xent_hard(sv, w, image_label, obj_labels)
    return total


def pretrain_forward(sd, cfg, input_ids, image_feat, image_loc, token_type_ids=None, attention_mask=None,
                     image_attention_mask=None, masked_lm_labels=None, image_label=None, image_cls=None,
                     next_sentence_label=None, train=False, philox_seed=None, taps=None, obj_labels=None, obj_confs=None,
                     attr_labels=None, attr_confs=None, nce_index=None):
    """-> (masked_lm_loss[1], img_loss[1], next_sentence_loss[1]) as the reference's
    BertForVLPreTraining.forward (encoders.py:1044-1112), every fusion method and visual target."""
    drop = Dropper(train, philox_seed)
    t, v, pt, pv = bert_model(sd, cfg, input_ids, image_feat, image_loc, token_type_ids, attention_mask,
                              image_attention_mask, drop, taps)
    pooled = fuse_pooled(cfg, pt, pv, drop, 0.1)
    c = "cls.predictions."
    h = layer_norm(gelu(linear(t, sd, c + "transform.dense")), sd[c + "transform.LayerNorm.weight"], sd[c + "transform.LayerNorm.bias"])
    scores_t = F.linear(h, sd["bert.embeddings.word_embeddings.weight"]) + sd[c + "bias"]
    ci = "cls.imagePredictions."
    hv = gelu(linear(v, sd, ci + "transform.dense"))
    if cfg.image_head_ln:
        hv = layer_norm(hv, sd[ci + "transform.LayerNorm.weight"], sd[ci + "transform.LayerNorm.bias"])
    scores_v = {ix: linear(hv, sd, ci + "decoder_dict." + ix) for ix, w in cfg.visual_target_weights.items() if w > 0}
    # the ITM head exists unless the fusion method is "none" / "vl-bert_vqa" (encoders.py:744-747)
    itm = linear(pooled, sd, "cls.bi_seq_relationship") if cfg.fusion_method not in ("none", "vl-bert_vqa") else None
    if taps is not None:
        taps.update(seq_t=t, seq_v=v, pooled_t=pt, pooled_v=pv, scores_t=scores_t, scores_v=scores_v.get("0"), scores_v_dict=scores_v, itm=itm)
    img_loss = visual_losses(cfg, scores_v, image_label, image_cls, image_feat, obj_labels, obj_confs, attr_labels, attr_confs, nce_index)
    if not float(img_loss.detach()) > 0:                       # encoders.py:1089-1093
        img_loss = torch.zeros(())
    lm_loss = F.cross_entropy(scores_t.reshape(-1, scores_t.shape[-1]), masked_lm_labels.reshape(-1), ignore_index=-1)
    if itm is not None and next_sentence_label is not None:
        nsp_loss = F.cross_entropy(itm.view(-1, 2), next_sentence_label.view(-1))
    else:
        nsp_loss = torch.zeros(())                             # encoders.py:1101-1107
    return lm_loss.reshape(1), img_loss.reshape(1), nsp_loss.reshape(1)


def objective1_relabel(lm_label_ids, image_label, is_match):
    """train_concap.py:279-284: mismatched pairs carry no MLM / region labels."""
    keep = (is_match == 0).long()[:, None]
    il = image_label * keep
    il = torch.where(il == 0, torch.full_like(il, -1), il)
    ll = lm_label_ids * keep
    ll = torch.where(ll == 0, torch.full_like(ll, -1), ll)
    return ll, il


# --------------------------------------------------------------------------------------- optimizer
NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def decays(name):
    """train_concap.py:201,220-224: substring test on the parameter name."""
    return not any(nd in name for nd in NO_DECAY)


def warmup_linear(step, warmup_steps, t_total):
    """pytorch-transformers 1.1.0 WarmupLinearSchedule multiplier (PARITY UNPINNED, see header)."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (train_concap.py:307-308): scale by max_norm/(norm+1e-6) if <1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = float(max_norm) / (float(total) + 1e-6)
    if coef < 1:
        for g in grads:
            g.mul_(coef)
    return total


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0, correct_bias=True):
    """pytorch-transformers==1.1.0 AdamW.step for one tensor, in place (PARITY UNPINNED: the package is
    not vendored under /root/reference; formula restated from its published source):
      m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ; denom = sqrt(v) + eps
      step_size = lr * sqrt(1-b2^t)/(1-b1^t) ; p <- p - step_size * m/denom ; then p <- p - lr*wd*p."""
    m.mul_(beta1).add_(g, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    denom = v.sqrt().add_(eps)
    step_size = lr
    if correct_bias:
        step_size = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p.addcdiv_(m, denom, value=-step_size)
    if weight_decay > 0.0:
        p.add_(p, alpha=-lr * weight_decay)


# --------------------------------------------------------------------------------------- weights / data
def param_shapes(cfg):
    """Ordered {name: shape} of the reference state_dict for the supported (ctrl_*) families.
    `cls.predictions.decoder.weight` is tied to the word embeddings and listed last as an alias."""
    H, Hv, I, Iv = cfg.hidden_size, cfg.v_hidden_size, cfg.intermediate_size, cfg.v_intermediate_size
    V, F_, P = cfg.vocab_size, cfg.v_feature_size, cfg.max_position_embeddings
    s = {}

    def lin(name, out, inp):
        s[name + ".weight"] = (out, inp)
        s[name + ".bias"] = (out,)

    def ln(name, n):
        s[name + ".weight"] = (n,)
        s[name + ".bias"] = (n,)

    e = "bert.embeddings."
    kind = cfg.image_embeddings
    if kind in ("vilbert", "lxmert"):
        s[e + "word_embeddings.weight"] = (V, H)
        s[e + "position_embeddings.weight"] = (P, H)
        s[e + "token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
        ln(e + "LayerNorm", H)
        ve = "bert.v_embeddings."
        lin(ve + "image_embeddings", Hv, F_)
        lin(ve + "image_location_embeddings", Hv, cfg.num_locs)
        if kind == "vilbert":
            ln(ve + "LayerNorm", Hv)
        else:
            ln(ve + "ImgLayerNorm", Hv)
            ln(ve + "LocLayerNorm", Hv)
    elif kind == "uniter":
        s[e + "word_embeddings.weight"] = (V, H)
        s[e + "position_embeddings.weight"] = (P, H)
        s[e + "token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
        ln(e + "LayerNorm", H)
        lin(e + "image_embeddings", Hv, F_)
        lin(e + "image_location_embeddings", Hv, cfg.num_locs)
        ln(e + "image_layer_norm", H)
        ln(e + "image_location_layer_norm", H)
        ln(e + "v_LayerNorm", H)
    elif kind == "visualbert":
        s[e + "word_embeddings.weight"] = (V, H)
        s[e + "position_embeddings.weight"] = (P, H)
        s[e + "token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
        ln(e + "LayerNorm", H)
        lin(e + "projection", H, F_)
        s[e + "token_type_embeddings_visual.weight"] = (cfg.type_vocab_size, H)
        s[e + "position_embeddings_visual.weight"] = (P, H)
    elif kind == "vl-bert":
        lin(e + "obj_downsample.1", Hv, 2 * F_)
        s[e + "object_linguistic_embeddings.weight"] = (1, H)
        if cfg.visual_target_weights.get("6", 0) > 0:
            s[e + "object_mask_word_embedding.weight"] = (1, H)
        s[e + "object_mask_visual_embedding.weight"] = (1, F_)
        s[e + "end_embedding.weight"] = (1, H)
        s[e + "word_embeddings.weight"] = (V, H)
        s[e + "position_embeddings.weight"] = (P, H)
        s[e + "token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
        ln(e + "visual_ln_text", H)
        ln(e + "visual_ln_object", H)
        ln(e + "LayerNorm", H)
    else:
        raise ValueError(kind)
    for n, typ in sublayer_schedule(cfg):
        p = "bert.encoder.layer.%d." % n
        if typ == "attn":
            has_t = n in cfg.tt_attn_sublayers or n in cfg.tv_attn_sublayers
            has_v = n in cfg.vv_attn_sublayers or n in cfg.vt_attn_sublayers
            shared = n in cfg.shared_sublayers and has_t and has_v
            # per-sub-layer attention widths (encoders.py:189-206,364-366; config/vilbert_base.json: the co-attention sub-layers project both
            # streams to 1024 = 8 heads of 128)
            Ha = cfg.sublayer2attn_hidden_size.get(str(n), H)
            Hva = cfg.sublayer2v_attn_hidden_size.get(str(n), Hv)
            if has_t:
                for k in ("query", "key", "value"):
                    lin(p + "attention_self." + k, Ha, H)
            if has_v and not shared:
                for k in ("v_query", "v_key", "v_value"):
                    lin(p + "attention_self." + k, Hva, Hv)
            if has_t:
                lin(p + "attention_output.dense", H, Ha)
                ln(p + "attention_output.LayerNorm", H)
            if has_v and not shared:
                lin(p + "attention_output.v_dense", Hv, Hva)
                ln(p + "attention_output.v_LayerNorm", Hv)
        else:
            has_t, has_v = n in cfg.t_ff_sublayers, n in cfg.v_ff_sublayers
            shared = n in cfg.shared_sublayers and has_t and has_v
            In = cfg.sublayer2intermediate_size.get(str(n), I)
            Ivn = cfg.sublayer2v_intermediate_size.get(str(n), Iv)
            if has_t:
                lin(p + "intermediate.dense", In, H)
            if has_v and not shared:
                lin(p + "intermediate.v_dense", Ivn, Hv)
            if has_t:
                lin(p + "output.dense", H, In)
                ln(p + "output.LayerNorm", H)
            if has_v and not shared:
                lin(p + "output.v_dense", Hv, Ivn)
                ln(p + "output.v_LayerNorm", Hv)
    fm = cfg.fusion_method
    if fm != "none":
        lin("bert.t_pooler.dense", cfg.pooler_size, H)
    if fm not in ("none", "text", "vl-bert_vqa"):
        lin("bert.v_pooler.dense", cfg.v_pooler_size, Hv)
    s["cls.predictions.bias"] = (V,)
    lin("cls.predictions.transform.dense", H, H)
    ln("cls.predictions.transform.LayerNorm", H)
    if fm not in ("none", "vl-bert_vqa"):
        lin("cls.bi_seq_relationship", 2, cfg.pooler_size)
    lin("cls.imagePredictions.transform.dense", Hv, Hv)
    if cfg.image_head_ln:
        ln("cls.imagePredictions.transform.LayerNorm", Hv)
    for ix, width in VIS_TARGET_WIDTH.items():               # encoders.py:725-729: one decoder per target with a positive weight, in
        if cfg.visual_target_weights.get(ix, 0) > 0:          # the order of losses.py's table (not the config's)
            lin("cls.imagePredictions.decoder_dict." + ix, width, Hv)
    return s


def param_aliases(cfg):
    """{alias key: owning key}: extra state_dict names that point at an already listed parameter --
    the tied LM decoder (encoders.py:1038-1042) and the v_* names of shared sub-layers, which the
    reference registers as the same module objects (encoders.py:208-213,384-388,471-475,526-531)."""
    al = {}
    for n, typ in sublayer_schedule(cfg):
        if n not in cfg.shared_sublayers:
            continue
        p = "bert.encoder.layer.%d." % n
        if typ == "attn":
            has_t = n in cfg.tt_attn_sublayers or n in cfg.tv_attn_sublayers
            has_v = n in cfg.vv_attn_sublayers or n in cfg.vt_attn_sublayers
            if has_t and has_v:
                for k in ("query", "key", "value"):
                    for wb in ("weight", "bias"):
                        al[p + "attention_self.v_%s.%s" % (k, wb)] = p + "attention_self.%s.%s" % (k, wb)
                for k in ("dense", "LayerNorm"):
                    for wb in ("weight", "bias"):
                        al[p + "attention_output.v_%s.%s" % (k, wb)] = p + "attention_output.%s.%s" % (k, wb)
        elif n in cfg.t_ff_sublayers and n in cfg.v_ff_sublayers:
            for wb in ("weight", "bias"):
                al[p + "intermediate.v_dense." + wb] = p + "intermediate.dense." + wb
                al[p + "output.v_dense." + wb] = p + "output.dense." + wb
                al[p + "output.v_LayerNorm." + wb] = p + "output.LayerNorm." + wb
    al["cls.predictions.decoder.weight"] = "bert.embeddings.word_embeddings.weight"
    return al


def make_weights(cfg, seed=0, std=0.05):
    """Deterministic "trained-like" weights: matrices ~ N(0,std), biases ~ N(0,std/2), LN weight
    1 + N(0,0.1), LN bias N(0,0.05).  Counter-based per tensor, so a fixture only stores the seed."""
    sd = {}
    for i, (name, shape) in enumerate(param_shapes(cfg).items()):
        g = torch.Generator().manual_seed(seed * 100003 + i)
        is_ln = "LayerNorm" in name or "layer_norm" in name or "visual_ln" in name
        if is_ln and name.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif is_ln:
            t = 0.05 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            t = (std / 2) * torch.randn(shape, generator=g)
        else:
            t = std * torch.randn(shape, generator=g)
        sd[name] = t
    for alias, target in param_aliases(cfg).items():
        sd[alias] = sd[target]
    return sd


def synthetic_batch(cfg, B, T=20, R=36, seed=1234, device="cpu", pad=False):
    """Counterpart of ConceptCapLoaderTrain's 15-tensor batch on random data.
    Returns a dict with the reference driver's tensor names (train_concap.py:275-277)."""
    g = torch.Generator().manual_seed(seed)
    V = cfg.vocab_size
    lo = min(1000, V // 4)
    ids = torch.randint(lo, V, (B, T), generator=g)
    ids[:, 0], ids[:, T - 1] = min(101, V - 2), min(102, V - 1)
    input_mask = torch.ones(B, T, dtype=torch.long)
    if pad:  # ragged captions: zero-padded tails
        lens = torch.randint(max(3, T // 2), T + 1, (B,), generator=g)
        ar = torch.arange(T)[None]
        input_mask = (ar < lens[:, None]).long()
        ids = ids * input_mask
    lm = torch.full((B, T), -1, dtype=torch.long)
    sel = (torch.rand(B, T, generator=g) < 0.15) & (input_mask == 1)
    sel[:, 0] = False
    sel[:, T - 1] = False
    lm[sel] = ids[sel]
    ids = torch.where(sel, torch.full_like(ids, min(103, V - 3)), ids)
    is_match = (torch.rand(B, generator=g) < 0.5).long()
    feat = torch.rand(B, R, cfg.v_feature_size, generator=g)
    image_label = torch.where(torch.rand(B, R, generator=g) < 0.15, 1, -1)
    zero = (image_label == 1) & (torch.rand(B, R, generator=g) < 0.9)
    feat = feat * (~zero)[..., None]
    xy = torch.rand(B, R, 2, generator=g) * 0.6
    wh = torch.rand(B, R, 2, generator=g) * 0.3 + 0.1
    loc = torch.cat([xy, xy + wh, (wh[..., :1] * wh[..., 1:])], -1)
    image_mask = torch.ones(B, R, dtype=torch.long)
    if pad:
        nreg = torch.randint(max(2, R // 2), R + 1, (B,), generator=g)
        image_mask = (torch.arange(R)[None] < nreg[:, None]).long()
        image_label = torch.where(image_mask == 1, image_label, torch.full_like(image_label, -1))
    cls = torch.softmax(torch.randn(B, R, 1601, generator=g), -1)
    if cfg.add_global_imgfeat is not None:
        gfeat = (feat * image_mask[..., None]).sum(1, keepdim=True) / image_mask.sum(1)[:, None, None]
        gloc = torch.tensor([0.0, 0.0, 1.0, 1.0, 1.0]).expand(B, 1, 5)
        one = torch.ones(B, 1, dtype=torch.long)
        if cfg.add_global_imgfeat == "first":
            feat, loc, image_mask = torch.cat([gfeat, feat], 1), torch.cat([gloc, loc], 1), torch.cat([one, image_mask], 1)
        else:
            feat, loc, image_mask = torch.cat([feat, gfeat], 1), torch.cat([loc, gloc], 1), torch.cat([image_mask, one], 1)
    if cfg.objective == 1 or True:  # the ctrl_* launch scripts all run objective 1
        lm, image_label = objective1_relabel(lm, image_label, is_match)
    batch = dict(input_ids=ids, input_mask=input_mask, segment_ids=torch.zeros(B, T, dtype=torch.long),
                 lm_label_ids=lm, is_match=is_match, image_feat=feat.contiguous(), image_loc=loc.contiguous(),
                 image_cls=cls, image_label=image_label, image_mask=image_mask)
    if set(cfg.visual_target_weights) - {"0"}:                 # detector outputs for the hard-label targets (drawn last: the
        g2 = torch.Generator().manual_seed(seed + 77)          # tensors above do not depend on the configured targets)
        batch.update(obj_labels=torch.randint(0, 1600, (B, R), generator=g2), obj_confs=torch.rand(B, R, generator=g2),
                     attr_labels=torch.randint(0, 400, (B, R), generator=g2), attr_confs=torch.rand(B, R, generator=g2))
    if cfg.num_locs != 5:
        batch["image_loc"] = batch["image_loc"][..., :cfg.num_locs].contiguous()
    return {k: v.to(device) for k, v in batch.items()}


def forward_from_batch(sd, cfg, b, **kw):
    return pretrain_forward(sd, cfg, b["input_ids"], b["image_feat"], b["image_loc"], b["segment_ids"],
                            b["input_mask"], b["image_mask"], b["lm_label_ids"], b["image_label"],
                            b["image_cls"], b["is_match"], obj_labels=b.get("obj_labels"), obj_confs=b.get("obj_confs"),
                            attr_labels=b.get("attr_labels"), attr_confs=b.get("attr_confs"), **kw)


def hf_style_bert_state_dict(cfg, n_layers, seed=0, with_prefix=True):
    """A BERT checkpoint in the HuggingFace key layout (`bert.encoder.layer.N.attention.self.query.weight`, old-style
    `LayerNorm.gamma/beta` in the embeddings, a pooler the VOLTA models do not have), filled by a counter-based
    generator: the input of the `from_hf=True` branch of the reference loader (volta/utils.py:458-498)."""
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    pre = "bert." if with_prefix else ""
    shapes = {
        pre + "embeddings.word_embeddings.weight": (V, H),
        pre + "embeddings.position_embeddings.weight": (cfg.max_position_embeddings, H),
        pre + "embeddings.token_type_embeddings.weight": (cfg.type_vocab_size, H),
        pre + "embeddings.LayerNorm.gamma": (H,),
        pre + "embeddings.LayerNorm.beta": (H,),
        pre + "pooler.dense.weight": (H, H),
        pre + "pooler.dense.bias": (H,),
    }
    for n in range(n_layers):
        b = pre + "encoder.layer.%d." % n
        for nm in ("query", "key", "value"):
            shapes[b + "attention.self.%s.weight" % nm] = (H, H)
            shapes[b + "attention.self.%s.bias" % nm] = (H,)
        shapes[b + "attention.output.dense.weight"] = (H, H)
        shapes[b + "attention.output.dense.bias"] = (H,)
        shapes[b + "attention.output.LayerNorm.weight"] = (H,)
        shapes[b + "attention.output.LayerNorm.bias"] = (H,)
        shapes[b + "intermediate.dense.weight"] = (I, H)
        shapes[b + "intermediate.dense.bias"] = (I,)
        shapes[b + "output.dense.weight"] = (H, I)
        shapes[b + "output.dense.bias"] = (H,)
        shapes[b + "output.LayerNorm.weight"] = (H,)
        shapes[b + "output.LayerNorm.bias"] = (H,)
    if with_prefix:
        shapes["cls.predictions.bias"] = (V,)
        shapes["cls.predictions.transform.dense.weight"] = (H, H)
        shapes["cls.predictions.transform.dense.bias"] = (H,)
        shapes["cls.predictions.transform.LayerNorm.gamma"] = (H,)
        shapes["cls.predictions.transform.LayerNorm.beta"] = (H,)
        shapes["cls.predictions.decoder.weight"] = (V, H)
        shapes["cls.seq_relationship.weight"] = (2, H)
    sd = {}
    for i, (name, shape) in enumerate(shapes.items()):
        g = torch.Generator().manual_seed(seed * 7919 + i)
        sd[name] = 0.1 * torch.randn(shape, generator=g) + 0.01 * i
    if with_prefix:
        sd["cls.predictions.decoder.weight"] = sd[pre + "embeddings.word_embeddings.weight"]      # tied in BERT
    return sd


# ---------------------------------------------------------------------------------------- downstream tasks
def task_head_shapes(cfg, task_cfg, task_ids):
    """Parameter names / shapes of BertForVLTasks.clfs_dict (volta/encoders.py:1128-1152, SimpleClassifier :787-797)."""
    P, Hc, Hv = cfg.pooler_size, cfg.clf_hidden_size, cfg.v_hidden_size
    out = {}
    for t in task_ids:
        typ, pre = task_cfg[t]["type"], "clfs_dict.%s." % t
        if typ in ("VL-classifier", "VL-classifier-GQA", "VL-binary-classifier"):
            din = P * 2 if typ == "VL-binary-classifier" else P
            dout = 2 if typ == "VL-binary-classifier" else task_cfg[t]["num_labels"]
            out.update({pre + "logit_fc.0.weight": (Hc, din), pre + "logit_fc.0.bias": (Hc,), pre + "logit_fc.2.weight": (Hc,),
                        pre + "logit_fc.2.bias": (Hc,), pre + "logit_fc.3.weight": (dout, Hc), pre + "logit_fc.3.bias": (dout,)})
        elif typ == "VL-tri-classifier":
            out.update({pre + "weight": (3, P), pre + "bias": (3,)})
        elif typ == "VL-logit":
            out.update({pre + "weight": (1, P), pre + "bias": (1,)})
        elif typ.startswith("V-logit"):
            if task_cfg[t].get("num_clf_layers", 1) == 2:
                out.update({pre + "0.weight": (Hv, Hv), pre + "0.bias": (Hv,), pre + "3.weight": (1, Hv), pre + "3.bias": (1,)})
            else:
                out.update({pre + "weight": (1, Hv), pre + "bias": (1,)})
        else:
            raise ValueError(typ)
    return out


def make_task_weights(cfg, task_cfg, task_ids, seed=0, std=0.05):
    """Encoder weights of make_weights (without the pre-training heads `cls.*`) + generated task heads."""
    sd = {k: v for k, v in make_weights(cfg, seed=seed, std=std).items() if not k.startswith("cls.")}
    for i, (name, shape) in enumerate(task_head_shapes(cfg, task_cfg, task_ids).items()):
        g = torch.Generator().manual_seed(seed * 100003 + 50000 + i)
        if name.endswith("logit_fc.2.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            t = (std / 2) * torch.randn(shape, generator=g)
        else:
            t = std * torch.randn(shape, generator=g)
        sd[name] = t
    return sd


def tasks_forward(sd, cfg, task_cfg, task_id, input_ids, image_feat, image_loc, token_type_ids=None, attention_mask=None,
                  image_attention_mask=None, taps=None):
    """BertForVLTasks.forward in eval mode (volta/encoders.py:1159-1206): vil_prediction."""
    seq_t, seq_v, pooled_t, pooled_v = bert_model(sd, cfg, input_ids, image_feat, image_loc, token_type_ids, attention_mask,
                                                   image_attention_mask, taps=taps)
    pooled = fuse_pooled(cfg, pooled_t, pooled_v)
    typ, pre = task_cfg[task_id]["type"], "clfs_dict.%s." % task_id

    def simple(x):
        h = x @ sd[pre + "logit_fc.0.weight"].t() + sd[pre + "logit_fc.0.bias"]
        h = layer_norm(gelu(h), sd[pre + "logit_fc.2.weight"], sd[pre + "logit_fc.2.bias"])
        return h @ sd[pre + "logit_fc.3.weight"].t() + sd[pre + "logit_fc.3.bias"]

    if typ.startswith("V-logit"):
        if image_attention_mask is None:
            image_attention_mask = torch.ones(image_feat.shape[:2], dtype=torch.long)
        if task_cfg[task_id].get("num_clf_layers", 1) == 2:
            h = gelu(seq_v @ sd[pre + "0.weight"].t() + sd[pre + "0.bias"])
            out = h @ sd[pre + "3.weight"].t() + sd[pre + "3.bias"]
        else:
            out = seq_v @ sd[pre + "weight"].t() + sd[pre + "bias"]
        return out + ((1.0 - image_attention_mask.float()) * -10000.0).unsqueeze(2)
    if typ == "VL-binary-classifier":
        return simple(pooled.view(-1, pooled.size(1) * 2))
    if typ in ("VL-classifier", "VL-classifier-GQA"):
        return simple(pooled)
    return pooled @ sd[pre + "weight"].t() + sd[pre + "bias"]


# ---------------------------------------------------------------------------------------- ConceptCap batch producer
# The reference draws from Python's `random` / numpy; here every decision is a function of an explicit 32-bit word per
# (pair, slot, stream) so that the policy can be replayed: the product's HIP kernels take the words from Philox
# (stream = site, counter = (slot, pair, site, 0)), the fixtures feed the reference code the same words as word / 2^32.
CC_T15 = int(0.15 * 2 ** 32)            # "prob < 0.15"
CC_SITE_TOKEN, CC_SITE_RANDTOK, CC_SITE_REGION, CC_SITE_CAPTION = 0, 1, 2, 3


def concap_words(seed, site, n_pairs, n_slots):
    """uint32 [n_pairs, n_slots]: word 0 of philox(counter = (slot, pair, site, 0), key = seed)."""
    import numpy as np
    pair = np.repeat(np.arange(n_pairs, dtype=np.uint32), n_slots)
    slot = np.tile(np.arange(n_slots, dtype=np.uint32), n_pairs)
    w = philox_raw(slot, pair, np.full_like(pair, site), np.zeros_like(pair), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return w[:, 0].reshape(n_pairs, n_slots)


def concap_iou(boxes):
    """Pairwise IoU with the +1 pixel convention (volta/datasets/concept_cap_dataset.py:31-68); boxes [N, 4] float."""
    import numpy as np
    b = np.asarray(boxes, np.float32)
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    iw = np.minimum(b[:, None, 2], b[None, :, 2]) - np.maximum(b[:, None, 0], b[None, :, 0]) + 1
    ih = np.minimum(b[:, None, 3], b[None, :, 3]) - np.maximum(b[:, None, 1], b[None, :, 1]) + 1
    iw, ih = np.maximum(iw, 0), np.maximum(ih, 0)
    return iw * ih / (area[:, None] + area[None, :] - iw * ih)


def concap_random_word(tokens, words, rand_words, vocab_size, mask_id):
    """random_word (concept_cap_dataset.py:612-641): 15 % of the tokens are selected; of those 80 % -> [MASK], 10 % -> a random
    id, 10 % kept; the label is the original id, -1 elsewhere.  `prob / 0.15 < 0.8` is evaluated on the same draw."""
    out, lab = list(tokens), []
    for i, tok in enumerate(tokens):
        prob = float(words[i]) / 2.0 ** 32
        if prob < 0.15:
            prob /= 0.15
            if prob < 0.8:
                out[i] = mask_id
            elif prob < 0.9:
                out[i] = int(rand_words[i]) % vocab_size
            lab.append(tok)
        else:
            lab.append(-1)
    return out, lab


def concap_random_region(image_feat, num_boxes, overlaps, words):
    """random_region (concept_cap_dataset.py:643-668): 15 % of the boxes get label 1, 90 % of those have their feature row
    zeroed; `masked_label` ORs the IoU > 0.4 rows of every selected box (it only feeds the global-feature count)."""
    import numpy as np
    feat = np.array(image_feat, np.float32, copy=True)
    label, masked = [], np.zeros(feat.shape[0], bool)
    for i in range(num_boxes):
        prob = float(words[i]) / 2.0 ** 32
        if prob < 0.15:
            prob /= 0.15
            if prob < 0.9:
                feat[i] = 0
            masked = np.logical_or(masked, overlaps[i] > 0.4)
            label.append(1)
        else:
            label.append(-1)
    return feat, label, masked


def concap_make_batch(records, captions, seed, seq_len, region_len, vocab_size, add_global="first", num_locs=5, objective=1,
                      cls_id=101, sep_id=102, mask_id=103):
    """BertPreprocessBatch.__call__ + convert_example_to_features + ConceptCapLoaderTrain.__iter__ (concept_cap_dataset.py:
    429-500, 546-610, 229-286) + the objective-1 relabel of train_concap.py:279-284, on a list of raw records
    {caption_index, feat [n, F], cls [n, C], boxes [n, 4] (pixels), w, h}.  Returns the dict of the model's input tensors."""
    import numpy as np
    B, T, R = len(records), seq_len, region_len
    F, C = records[0]["feat"].shape[1], records[0]["cls"].shape[1]
    w_tok = concap_words(seed, CC_SITE_TOKEN, B, T)
    w_rnd = concap_words(seed, CC_SITE_RANDTOK, B, T)
    w_reg = concap_words(seed, CC_SITE_REGION, B, R)
    w_cap = concap_words(seed, CC_SITE_CAPTION, B, 2)
    out = dict(input_ids=np.zeros((B, T), np.int64), input_mask=np.zeros((B, T), np.int64), segment_ids=np.zeros((B, T), np.int64),
               lm_label_ids=np.full((B, T), -1, np.int64), is_match=np.zeros(B, np.int64), image_label=np.full((B, R), -1, np.int64),
               image_cls=np.zeros((B, R, C), np.float32))
    feats, locs, masks, masked_all = [], [], [], []
    for b, rec in enumerate(records):
        n = int(rec["feat"].shape[0])
        # random_cap (:505-522): objective 2 never swaps
        cap_idx, label = rec["caption_index"], 0
        if objective != 2 and float(w_cap[b, 0]) / 2.0 ** 32 > 0.5:
            cap_idx, label = int(w_cap[b, 1]) % len(captions), 1
        tokens = list(captions[cap_idx])[:T - 2]
        tokens, tok_lab = concap_random_word(tokens, w_tok[b], w_rnd[b], vocab_size, mask_id)
        ids = [cls_id] + tokens + [sep_id]
        out["input_ids"][b, :len(ids)] = ids
        out["input_mask"][b, :len(ids)] = 1
        out["lm_label_ids"][b, 1:1 + len(tok_lab)] = tok_lab
        out["is_match"][b] = label
        feat = np.zeros((R, F), np.float32)
        feat[:n] = rec["feat"]
        out["image_cls"][b, :n] = rec["cls"]
        loc = np.zeros((R, num_locs), np.float32)
        loc[:n, :4] = rec["boxes"]
        if num_locs == 5:
            loc[:, 4] = (loc[:, 3] - loc[:, 1]) * (loc[:, 2] - loc[:, 0]) / (float(rec["w"]) * float(rec["h"]))
        loc[:, 0] /= float(rec["w"]); loc[:, 2] /= float(rec["w"]); loc[:, 1] /= float(rec["h"]); loc[:, 3] /= float(rec["h"])
        ov = np.zeros((R, R), np.float32)
        ov[:n, :n] = concap_iou(rec["boxes"])
        feat, lab, masked = concap_random_region(feat, n, ov, w_reg[b])
        out["image_label"][b, :n] = lab
        feats.append(feat); locs.append(loc); masked_all.append(masked)
        masks.append(np.concatenate([np.ones(n, np.int64), np.zeros(R - n, np.int64)]))
    feat, loc, mask, masked = np.stack(feats), np.stack(locs), np.stack(masks), np.stack(masked_all)
    if add_global is not None:
        cnt = np.sum(masked == 0, axis=1, keepdims=True)
        cnt[cnt == 0] = 1
        g = (np.sum(feat, axis=1) / cnt).astype(np.float32)[:, None]
        gloc = np.tile(np.array([[0, 0, 1, 1] + [1] * (num_locs - 4)], np.float32), (B, 1))[:, None]
        one = np.ones((B, 1), np.int64)
        if add_global == "first":
            feat, loc, mask = np.concatenate([g, feat], 1), np.concatenate([gloc, loc], 1), np.concatenate([one, mask], 1)
        else:
            feat, loc, mask = np.concatenate([feat, g], 1), np.concatenate([loc, gloc], 1), np.concatenate([mask, one], 1)
    out.update(image_feat=feat.astype(np.float32), image_loc=loc.astype(np.float32), image_mask=mask)
    if objective == 1:      # train_concap.py:279-284: mismatched pairs carry no MLM / region targets
        keep = (out["is_match"] == 0).astype(np.int64)
        for k in ("image_label", "lm_label_ids"):
            v = out[k] * keep[:, None]
            v[v == 0] = -1
            out[k] = v
    return out
