"""ConceptCap batch producer kernels (vk_concap_batch) against the oracle's restatement of the reference pipeline on the same raw
records and the same Philox words: integer tensors bit-exact, float tensors to 1e-6 (the global feature row sums 36 fp32 values in a
different order).  Ragged box counts, both global-feature positions, all three objectives; then the B=256 / 2048-wide production shape
against size-independent properties.  GPU only."""
import numpy as np
import pytest
import torch

from oracle import volta_ref as R
from oracle.make_golden import concap_records

pytestmark = pytest.mark.gpu


def _run(recs, caps, seed, T, Rl, V, add_global, objective):
    from volta_amd.data import ConceptCapBatchProducer
    B, F, C = len(recs), recs[0]["feat"].shape[1], recs[0]["cls"].shape[1]
    feat, cls, boxes = np.zeros((B, Rl, F), np.float32), np.zeros((B, Rl, C), np.float32), np.zeros((B, Rl, 4), np.float32)
    for b, r in enumerate(recs):
        n = r["feat"].shape[0]
        feat[b, :n], cls[b, :n], boxes[b, :n] = r["feat"], r["cls"], r["boxes"]
        feat[b, n:] = 7.0                                  # garbage beyond num_boxes must not leak
    prod = ConceptCapBatchProducer(caps, T, Rl, V, add_global_imgfeat=add_global, objective=objective)
    out = prod(torch.from_numpy(feat).cuda(), torch.from_numpy(cls).cuda(), torch.from_numpy(boxes).cuda(),
               torch.tensor([r["feat"].shape[0] for r in recs]), torch.tensor([[r["w"], r["h"]] for r in recs], dtype=torch.float32),
               torch.tensor([r["caption_index"] for r in recs]), seed)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items() if k != "_keep"}


@pytest.mark.parametrize("add_global,objective,ragged", [("first", 1, True), ("last", 0, True), (None, 2, False), ("first", 0, False)])
def test_producer_matches_oracle(add_global, objective, ragged):
    T, Rl, V, seed = 14, 10, 3000, 1234567
    recs, caps = concap_records(24, Rl, seed=5, F=96, C=33, ragged=ragged)
    got = _run(recs, caps, seed, T, Rl, V, add_global, objective)
    want = R.concap_make_batch(recs, caps, seed=seed, seq_len=T, region_len=Rl, vocab_size=V, add_global=add_global, num_locs=5, objective=objective)
    for k in ("input_ids", "input_mask", "segment_ids", "lm_label_ids", "is_match", "image_label", "image_mask"):
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    np.testing.assert_allclose(got["image_loc"], want["image_loc"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(got["image_feat"], want["image_feat"], rtol=2e-6, atol=1e-6)
    np.testing.assert_array_equal(got["image_cls"], want["image_cls"])
    assert (got["lm_label_ids"] != -1).sum() > 5 or objective == 1


def test_producer_full_size_properties():
    """B=256, 36 regions x 2048 features, T=20 (the benchmark shape)."""
    T, Rl, V, B = 20, 36, 30522, 256
    recs, caps = concap_records(B, Rl, seed=11, F=2048, C=1601)
    got = _run(recs, caps, 99, T, Rl, V, "first", 0)
    lab, il = got["lm_label_ids"], got["image_label"]
    ntok = got["input_mask"].sum() - 2 * B
    assert abs((lab != -1).sum() / ntok - 0.15) < 0.02 and abs((il == 1).mean() - 0.15) < 0.02
    assert 0.4 < got["is_match"].mean() < 0.6
    assert (got["input_ids"][:, 0] == 101).all() and (got["input_ids"][lab != -1] != 0).all()
    masked_sel = got["input_ids"][lab != -1]
    assert 0.7 < (masked_sel == 103).mean() < 0.9                       # 80 % [MASK]
    # zeroed rows: ~90 % of the selected regions; every other row is an exact copy; the global row is the mean-like combination
    feat = np.stack([r["feat"] for r in recs])
    rows = got["image_feat"][:, 1:]
    zeroed = (rows == 0).all(2)
    assert (zeroed <= (il == 1)).all() and 0.8 < zeroed.sum() / (il == 1).sum() <= 1.0
    np.testing.assert_array_equal(rows[~zeroed], feat[~zeroed])
    cnt = got["image_feat"][:, 1:].sum(1) / np.maximum(got["image_feat"][:, 0], 1e-30)
    k = np.round(np.median(cnt, axis=1))
    assert ((k >= 1) & (k <= Rl)).all()                                 # divisor = number of rows not co-masked
    np.testing.assert_allclose(got["image_loc"][:, 0], np.tile([0, 0, 1, 1, 1], (B, 1)))
    assert (got["image_loc"][:, 1:, :4] >= 0).all() and (got["image_loc"][:, 1:, :4] <= 1.0 + 1e-6).all()
