"""ConceptCap batch producer: the oracle's restatement of the reference's per-batch policy -- caption swap,
15 % / 80-10-10 token masking, 15 % / 90 % region masking with IoU > 0.4 co-masking, box normalisation, global feature row --
against a fixture written by the REAL reference code driven with the same draws (oracle/make_golden.py concap).  CPU only."""
import os

import numpy as np

from oracle import volta_ref as R
from oracle.make_golden import concap_records

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_oracle_batch_matches_reference_pipeline():
    z = np.load(os.path.join(GOLD, "concap_batch.npz"))
    T, Rl, V, seed, B, rseed = (int(x) for x in z["params"])
    recs, caps = concap_records(B, Rl, seed=rseed)
    o = R.concap_make_batch(recs, caps, seed=seed, seq_len=T, region_len=Rl, vocab_size=V, add_global="first", num_locs=5, objective=0)
    for k in ("input_ids", "input_mask", "segment_ids", "lm_label_ids", "image_label", "image_mask"):
        np.testing.assert_array_equal(o[k], z[k], err_msg=k)
    np.testing.assert_array_equal(o["is_match"], z["is_next"])
    np.testing.assert_allclose(o["image_loc"], z["image_loc"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(o["image_feat"].astype(np.float64).sum(2), z["image_feat_rowsum"], rtol=1e-6)
    np.testing.assert_allclose(o["image_feat"][:, 0, :128], z["image_feat_global_head"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(o["image_cls"].astype(np.float64).sum(2), z["image_cls_rowsum"], rtol=1e-6)
    assert int((z["image_label"] == 1).sum()) > 5 and int((z["lm_label_ids"] != -1).sum()) > 5 and 0 < int(z["is_next"].sum()) < B


def test_policy_rates_and_objective1():
    recs, caps = concap_records(64, 10, seed=9, F=8, C=5, ragged=True)
    o = R.concap_make_batch(recs, caps, seed=123, seq_len=14, region_len=10, vocab_size=3000, objective=1)
    swapped = o["is_match"] == 1
    assert 0.3 < swapped.mean() < 0.7
    assert (o["lm_label_ids"][swapped] == -1).all() and (o["image_label"][swapped] == -1).all()      # train_concap.py:279-284
    w = R.concap_words(5, R.CC_SITE_REGION, 2000, 10)
    assert abs((w < R.CC_T15).mean() - 0.15) < 0.01
