"""LMDB records -> ConceptCapLoaderTrain -> the pre-training step, on the GPU: the loader's batch equals the batch producer fed with the same
arrays by hand (bit-exact: same kernels, same seed), and a model trains on it."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.lmdb_writer import pack_datapoint, write_lmdb  # noqa: E402
from tests.test_readers_cpu import _datapoint  # noqa: E402

pytestmark = pytest.mark.gpu


class Tok:
    vocab_size, cls_token_id, sep_token_id, mask_token_id = 30522, 101, 102, 103

    def encode(self, text, add_special_tokens=False):
        return [1000 + (sum(w.encode()) % 20000) for w in text.split()]


def _store(tmp_path, n, Rl):
    rng = np.random.default_rng(21)
    dps = []
    for i in range(n):
        dp = _datapoint(rng, Rl, 2048, 1601, 401)
        dp[1] = dp[1] / dp[1].sum(1, keepdims=True)
        dp[7][:, 2:] += dp[7][:, :2]                     # x2 > x1, y2 > y1
        dp[12] = " ".join("w%d" % int(x) for x in rng.integers(0, 500, int(rng.integers(3, 14))))
        dps.append(dp)
    write_lmdb(str(tmp_path / "training_feat_all.lmdb"), {b"%08d" % i: pack_datapoint(dp) for i, dp in enumerate(dps)})
    (tmp_path / "caption_train.json").write_text(json.dumps({dp[11]: dp[12] for dp in dps}))
    return dps


def test_loader_batches_equal_the_producer_fed_by_hand(tmp_path):
    from volta_amd import readers as R
    from volta_amd.data import ConceptCapBatchProducer
    Rl, B, T = 36, 4, 16
    dps = _store(tmp_path, 10, Rl)
    tok = Tok()
    ld = R.ConceptCapLoaderTrain(str(tmp_path), str(tmp_path), tok, seq_len=T, batch_size=B, region_len=Rl, add_global_imgfeat="first", objective=1,
                                 seed=5)
    assert len(ld) == 10
    corpus = [tok.encode(dp[12]) for dp in dps]
    names = ["input_ids", "input_mask", "segment_ids", "lm_label_ids", "is_match", "image_feat", "image_loc", "image_cls", None, None, None, None, None,
             "image_label", "image_mask"]
    seen = 0
    for step, batch in enumerate(ld):
        ids = batch[15]
        nb = len(ids)
        assert ids == [dp[11] for dp in dps[seen:seen + nb]] and len(batch) == 16
        mine = dps[seen:seen + nb]
        prod = ConceptCapBatchProducer(corpus + [tok.encode(dp[12]) for dp in mine], T, Rl, tok.vocab_size, add_global_imgfeat="first", objective=1,
                                       min_ld=T, n_random=10)
        st = lambda i, dt: torch.tensor(np.stack([dp[i] for dp in mine]), dtype=dt, device="cuda")
        want = prod(st(0, torch.float32), st(1, torch.float32), st(7, torch.float32), torch.full((nb,), Rl, dtype=torch.int32, device="cuda"),
                    torch.tensor([[float(dp[10]), float(dp[9])] for dp in mine], device="cuda"),
                    torch.arange(10, 10 + nb, dtype=torch.int32, device="cuda"), 5 * 1000003 + step)
        for i, name in enumerate(names):
            if name is not None:
                assert torch.equal(batch[i], want[name]), (step, name)
        for i, src in ((8, 2), (9, 3), (10, 4), (11, 5), (12, 6)):        # detector labels / confidences / attribute scores pass through
            assert np.array_equal(batch[i].cpu().numpy(), np.stack([dp[src] for dp in mine]))
        assert batch[5].shape == (nb, Rl + 1, 2048) and batch[6].shape == (nb, Rl + 1, 5) and batch[4].dtype == torch.int64
        seen += nb
    assert seen == 10 and step == 2
    # a shuffle window changes the order of whole batches, not their content
    ld2 = R.ConceptCapLoaderTrain(str(tmp_path), str(tmp_path), tok, seq_len=T, batch_size=B, region_len=Rl, add_global_imgfeat="first", seed=5, cache=3 * B)
    got = sorted(i for b in ld2 for i in b[15])
    assert got == sorted(dp[11] for dp in dps)


def test_model_trains_on_loader_batches(tmp_path):
    """The loader's tuple goes into BertForVLPreTraining in the order of the reference driver (train_concap.py:262-289)."""
    from volta_amd import readers as R
    from volta_amd.config import BertConfig
    from volta_amd.modeling import BertForVLPreTraining
    cfg = BertConfig.from_json_file(os.path.join(os.path.dirname(__file__), "..", "config", "ctrl_vilbert_base.json"))
    Rl, B, T = 36, 8, 20
    _store(tmp_path, B, Rl)
    ld = R.ConceptCapLoaderTrain(str(tmp_path), str(tmp_path), Tok(), seq_len=T, batch_size=B, region_len=Rl, add_global_imgfeat=cfg.add_global_imgfeat,
                                 objective=1, num_locs=cfg.num_locs, seed=5)
    torch.manual_seed(0)
    model = BertForVLPreTraining(cfg).cuda().train()
    batch = next(iter(ld))
    (input_ids, input_mask, segment_ids, lm_label_ids, is_match, image_feat, image_loc, image_cls, obj_labels, obj_confs, attr_labels, attr_confs,
     image_attrs, image_label, image_mask) = batch[:15]
    assert 0 < int(is_match.sum()) < B and int((lm_label_ids != -1).sum()) > 0 and int((image_label == 1).sum()) > 0
    lm, img, nsp = model(input_ids, image_feat, image_loc, segment_ids, input_mask, image_mask, lm_label_ids, image_label, image_cls, obj_labels,
                         obj_confs, attr_labels, attr_confs, image_attrs, is_match)
    (lm + img + nsp).backward()
    for x in (lm, img, nsp):
        assert torch.isfinite(x).all()
    assert float(lm) > 5.0 and 0.3 < float(nsp) < 1.5
    g = model.bert.embeddings.word_embeddings.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0


def test_validation_loader_visualization_mode_masks_nothing(tmp_path):
    """ConceptCapLoaderVal(visualization=True): every pair keeps its caption, no token and no region is masked, features arrive unchanged
    behind the global row (concept_cap_dataset.py:514,622,652); without the flag the usual 15 % policy applies to the same records."""
    from volta_amd import readers as R
    Rl, B, T = 36, 4, 16
    dps = _store(tmp_path, 8, Rl)
    os.rename(str(tmp_path / "training_feat_all.lmdb"), str(tmp_path / "validation_feat_all.lmdb"))
    os.rename(str(tmp_path / "caption_train.json"), str(tmp_path / "caption_valid.json"))
    tok = Tok()
    seen = 0
    for batch in R.ConceptCapLoaderVal(str(tmp_path), str(tmp_path), tok, seq_len=T, batch_size=B, region_len=Rl, visualization=True, seed=3):
        nb = len(batch[15])
        mine = dps[seen:seen + nb]
        assert int(batch[4].sum()) == 0 and int((batch[3] != -1).sum()) == 0 and int((batch[13] != -1).sum()) == 0
        for b, dp in enumerate(mine):
            ids = tok.encode(dp[12])[:T - 2]
            assert batch[0][b, :len(ids) + 2].tolist() == [101] + ids + [102]
            assert torch.equal(batch[5][b, 1:].cpu(), torch.tensor(dp[0]))
            assert torch.allclose(batch[5][b, 0].cpu(), torch.tensor(dp[0]).sum(0) / Rl, rtol=1e-5, atol=1e-6)
        seen += nb
    assert seen == 8
    masked = 0
    for batch in R.ConceptCapLoaderVal(str(tmp_path), str(tmp_path), tok, seq_len=T, batch_size=B, region_len=Rl, seed=3):
        masked += int((batch[3] != -1).sum()) + int((batch[13] == 1).sum())
    assert masked > 0


def test_loader_with_the_native_tokenizer(tmp_path):
    """WordPieceTokenizer in the loader: the batch's captions are tokenised by one native call; the text rows are [CLS] ids [SEP] of each
    record's own caption wherever the pair was not swapped and no token was masked."""
    from volta_amd import readers as R
    Rl, B, T = 36, 4, 16
    dps = _store(tmp_path, 8, Rl)
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + ["w", "##%d" % 0] + ["##%d" % i for i in range(1, 10)] + [str(i) for i in range(10)]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    tok = R.WordPieceTokenizer(str(tmp_path / "vocab.txt"))
    assert tok.encode("w12") == [5, vocab.index("##1"), vocab.index("##2")]
    ld = R.ConceptCapLoaderVal(str(_as_val(tmp_path)), str(tmp_path), tok, seq_len=T, batch_size=B, region_len=Rl, visualization=True, seed=3)
    seen = 0
    for batch in ld:
        for b in range(len(batch[15])):
            ids = tok.encode(dps[seen + b][12])[:T - 2]
            assert batch[0][b, :len(ids) + 2].tolist() == [2] + ids + [3] and int(batch[1][b].sum()) == len(ids) + 2
        seen += len(batch[15])
    assert seen == 8


def _as_val(tmp_path):
    os.rename(str(tmp_path / "training_feat_all.lmdb"), str(tmp_path / "validation_feat_all.lmdb"))
    os.rename(str(tmp_path / "caption_train.json"), str(tmp_path / "caption_valid.json"))
    return tmp_path
