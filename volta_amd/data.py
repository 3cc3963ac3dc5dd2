"""Synthetic ConceptualCaptions-shaped batches generated on the device (counterpart of
ConceptCapLoaderTrain.__iter__, volta/datasets/concept_cap_dataset.py:229-286, plus the objective-1 relabel
of train_concap.py:279-284).  Random region features / token ids / soft class targets with the reference's
masking rates; there is no network or dataset in the benchmark environment."""
import torch


def synthetic_batch(config, batch_size, seq_len=20, num_regions=36, seed=1234, device="cuda", objective=1):
    g = torch.Generator(device=device).manual_seed(seed)
    B, T, R, V = batch_size, seq_len, num_regions, config.vocab_size
    kw = dict(generator=g, device=device)
    ids = torch.randint(min(1000, V // 4), V, (B, T), **kw)
    ids[:, 0], ids[:, T - 1] = min(101, V - 2), min(102, V - 1)
    input_mask = torch.ones(B, T, dtype=torch.long, device=device)
    lm = torch.full((B, T), -1, dtype=torch.long, device=device)
    sel = torch.rand(B, T, **kw) < 0.15
    sel[:, 0] = False
    sel[:, T - 1] = False
    lm[sel] = ids[sel]
    ids = torch.where(sel, torch.full_like(ids, min(103, V - 3)), ids)
    is_match = (torch.rand(B, **kw) < 0.5).long()
    feat = torch.rand(B, R, config.v_feature_size, **kw)
    image_label = torch.where(torch.rand(B, R, **kw) < 0.15, 1, -1)
    zero = (image_label == 1) & (torch.rand(B, R, **kw) < 0.9)
    feat = feat * (~zero)[..., None]
    xy = torch.rand(B, R, 2, **kw) * 0.6
    wh = torch.rand(B, R, 2, **kw) * 0.3 + 0.1
    loc = torch.cat([xy, xy + wh, wh[..., :1] * wh[..., 1:]], -1)
    image_mask = torch.ones(B, R, dtype=torch.long, device=device)
    cls = torch.softmax(torch.randn(B, R, 1601, **kw), -1)
    if config.add_global_imgfeat is not None:
        gfeat = feat.mean(1, keepdim=True)
        gloc = torch.tensor([0.0, 0.0, 1.0, 1.0, 1.0], device=device).expand(B, 1, 5)
        one = torch.ones(B, 1, dtype=torch.long, device=device)
        parts = ([gfeat, feat], [gloc, loc], [one, image_mask]) if config.add_global_imgfeat == "first" else \
                ([feat, gfeat], [loc, gloc], [image_mask, one])
        feat, loc, image_mask = (torch.cat(p, 1) for p in parts)
    if objective == 1:      # mismatched pairs carry no MLM / region labels
        keep = (is_match == 0).long()[:, None]
        image_label = image_label * keep
        image_label = torch.where(image_label == 0, torch.full_like(image_label, -1), image_label)
        lm = lm * keep
        lm = torch.where(lm == 0, torch.full_like(lm, -1), lm)
    out = dict(input_ids=ids, input_mask=input_mask, segment_ids=torch.zeros(B, T, dtype=torch.long, device=device),
               lm_label_ids=lm, is_match=is_match, image_feat=feat.contiguous(), image_loc=loc[..., :config.num_locs].contiguous(), image_cls=cls,
               image_label=image_label, image_mask=image_mask)
    if set(config.visual_target_weights) - {"0"}:          # detector outputs for the hard-label targets (volta/losses.py:83-124)
        out.update(obj_labels=torch.randint(0, 1600, (B, R), **kw), obj_confs=torch.rand(B, R, **kw),
                   attr_labels=torch.randint(0, 400, (B, R), **kw), attr_confs=torch.rand(B, R, **kw))
    return out


def model_args(b):
    """Positional arguments in the order of the reference driver's call (train_concap.py:286-289)."""
    return (b["input_ids"], b["image_feat"], b["image_loc"], b["segment_ids"], b["input_mask"], b["image_mask"], b["lm_label_ids"],
            b["image_label"], b["image_cls"], b.get("obj_labels"), b.get("obj_confs"), b.get("attr_labels"), b.get("attr_confs"), None, b["is_match"])


class ConceptCapBatchProducer:
    """Device-side counterpart of the reference's BertPreprocessBatch + ConceptCapLoaderTrain batch assembly
    (volta/datasets/concept_cap_dataset.py:229-286, 403-668): raw per-pair records already on the GPU -> the tensors of
    `model_args`.  `captions`: list of token-id lists (no [CLS] / [SEP]); records: `feat [B, R, F]`, `cls [B, R, C]`, `boxes [B, R, 4]`
    (pixels), `num_boxes [B]`, `img_wh [B, 2]`, `cap_index [B]`.  One call = three HIP launches (csrc/concap.hip)."""

    def __init__(self, captions, seq_len, region_len, vocab_size, add_global_imgfeat="first", objective=1, cls_id=101, sep_id=102, mask_id=103,
                 device="cuda", extra_rows=0, min_ld=0, n_random=None, visualization=False):
        """`extra_rows` empty rows behind the corpus (a loader writes each batch's own captions there and points `cap_index` at them);
        `n_random`: replacement captions are drawn from rows [0, n_random) (default: all of `captions`)."""
        from . import _lib as L
        self.L = L
        self.T, self.R, self.V = int(seq_len), int(region_len), int(vocab_size)
        self.add_global = {None: 0, "first": 1, "last": 2}[add_global_imgfeat]
        self.objective, self.ids = int(objective), (int(cls_id), int(sep_id), int(mask_id))
        ld = max(1, int(min_ld), max((len(c) for c in captions), default=1))
        tok = torch.zeros(len(captions) + int(extra_rows), ld, dtype=torch.int32)
        for i, c in enumerate(captions):
            tok[i, :len(c)] = torch.tensor(c, dtype=torch.int32)
        self.cap_tokens = tok.to(device)
        self.cap_len = torch.tensor([len(c) for c in captions] + [0] * int(extra_rows), dtype=torch.int32, device=device)
        self.n_random = len(captions) if n_random is None else int(n_random)
        self.visualization = int(bool(visualization))        # no swap, no masking (the reference's validation loader can ask for it)
        self.device = device

    def __call__(self, feat, cls, boxes, num_boxes, img_wh, cap_index, seed):
        L, C = self.L, __import__("ctypes")
        B, R, F = feat.shape
        assert R == self.R and boxes.shape == (B, R, 4) and cls.shape[:2] == (B, R)
        Cn = cls.shape[2]
        Rv = R + (1 if self.add_global else 0)
        dev = feat.device
        i64 = dict(dtype=torch.int64, device=dev)
        out = dict(input_ids=torch.empty(B, self.T, **i64), input_mask=torch.empty(B, self.T, **i64), segment_ids=torch.empty(B, self.T, **i64),
                   lm_label_ids=torch.empty(B, self.T, **i64), is_match=torch.empty(B, **i64), image_feat=torch.empty(B, Rv, F, device=dev),
                   image_loc=torch.empty(B, Rv, 5, device=dev), image_cls=torch.empty(B, R, Cn, device=dev), image_label=torch.empty(B, R, **i64),
                   image_mask=torch.empty(B, Rv, **i64))
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous()
        keep = [f32(feat), f32(cls), f32(boxes), i32(num_boxes), f32(img_wh), i32(cap_index)]
        a = L.ConcapArgs(L.ptr(self.cap_tokens), L.ptr(self.cap_len), L.ptr(keep[5]), L.ptr(keep[0]), L.ptr(keep[1]), L.ptr(keep[2]), L.ptr(keep[3]),
                         L.ptr(keep[4]), *[L.ptr(out[k]) for k in ("input_ids", "input_mask", "segment_ids", "lm_label_ids", "is_match", "image_feat",
                                                                    "image_loc", "image_cls", "image_label", "image_mask")],
                         C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), B, self.T, R, F, Cn, self.n_random, self.cap_tokens.shape[1], self.V,
                         self.ids[0], self.ids[1], self.ids[2], self.add_global, self.objective, self.visualization)
        L.check(L.lib.vk_concap_batch(C.byref(a), L.stream_ptr()))
        out["_keep"] = keep          # inputs stay alive until the stream has consumed them
        return out
