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
    return dict(input_ids=ids, input_mask=input_mask, segment_ids=torch.zeros(B, T, dtype=torch.long, device=device),
                lm_label_ids=lm, is_match=is_match, image_feat=feat.contiguous(), image_loc=loc.contiguous(), image_cls=cls,
                image_label=image_label, image_mask=image_mask)


def model_args(b):
    """Positional arguments in the order of the reference driver's call (train_concap.py:286-289)."""
    return (b["input_ids"], b["image_feat"], b["image_loc"], b["segment_ids"], b["input_mask"], b["image_mask"], b["lm_label_ids"],
            b["image_label"], b["image_cls"], None, None, None, None, None, b["is_match"])
