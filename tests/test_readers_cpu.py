"""Record readers: `vk_lmdb_*`, `vk_concap_record_decode`, `vk_b64_decode` and the Python classes over them.  Host code:
runs without a GPU.  The LMDB files are laid out by tests/lmdb_writer.py (no `lmdb` package here: the container format is parity-unpinned
against real files); the arithmetic of ImageFeaturesH5Reader is pinned by a fixture the REAL reference class wrote."""
import base64
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.lmdb_writer import pack_datapoint, write_lmdb  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _readers():
    from volta_amd import readers
    return readers


def _records(n, seed=0, big_every=5):
    rng = np.random.default_rng(seed)
    recs = {}
    for i in range(n):
        size = int(rng.integers(0, 200)) if i % big_every else int(rng.integers(3000, 30000))
        recs[b"%08d" % (i * 7)] = rng.integers(0, 256, size, dtype=np.uint8).tobytes()
    return recs


@pytest.mark.parametrize("n,max_keys,depth", [(1, None, 1), (400, None, 2), (60, 4, 3), (300, 3, 5)])
def test_lmdb_scan_and_lookup(tmp_path, n, max_keys, depth):
    """Every record comes back byte-exact, in key order, from trees of depth 1..5 with in-page and overflow values; point lookups
    find every key and reject absent ones (before the first, between two, after the last, a prefix, an extension)."""
    R = _readers()
    recs = _records(n)
    path = str(tmp_path / "db.lmdb")
    info = write_lmdb(path, recs, max_keys=max_keys)
    assert info["depth"] >= depth and info["overflow_pages"] > 0
    db = R.LMDBReader(path)
    assert len(db) == n
    got = [(k, bytes(v)) for k, v in db]
    assert got == sorted(recs.items())
    assert [(k, bytes(v)) for k, v in db] == got                     # the cursor rewinds
    for k, v in recs.items():
        assert bytes(db.get(k)) == v
    keys = sorted(recs)
    for absent in (b"", b"0", keys[0][:-1], keys[0] + b"0", keys[-1] + b"\xff", b"99999999", keys[len(keys) // 2][:-1] + b"\x00"):
        if absent not in recs:
            assert db.get(absent) is None
    db.close()
    with pytest.raises(ValueError):
        db.get(keys[0])


def test_lmdb_directory_layout_empty_db_and_bad_files(tmp_path):
    R = _readers()
    from volta_amd._lib import VoltaHipError
    d = tmp_path / "store"
    d.mkdir()
    write_lmdb(str(d / "data.mdb"), {b"a": b"1", b"keys": b"2"})
    db = R.LMDBReader(str(d))                                        # the directory that holds data.mdb
    assert bytes(db.get("a")) == b"1" and [k for k, _ in db] == [b"a", b"keys"]
    write_lmdb(str(tmp_path / "empty.lmdb"), {})
    e = R.LMDBReader(str(tmp_path / "empty.lmdb"))
    assert len(e) == 0 and list(e) == [] and e.get(b"x") is None
    (tmp_path / "junk").write_bytes(b"\x01" * 10000)
    with pytest.raises(VoltaHipError, match="meta page"):
        R.LMDBReader(str(tmp_path / "junk"))
    with pytest.raises(VoltaHipError, match="stat"):
        R.LMDBReader(str(tmp_path / "missing"))
    whole = (tmp_path / "empty.lmdb").read_bytes()
    good = str(tmp_path / "t.lmdb")
    write_lmdb(good, _records(40))
    (tmp_path / "cut.lmdb").write_bytes(open(good, "rb").read()[:3 * 4096])     # truncated copy: pages missing
    with pytest.raises(VoltaHipError):
        c = R.LMDBReader(str(tmp_path / "cut.lmdb"))
        list(c)
    assert len(whole) == 2 * 4096


def test_base64_decoder_matches_the_standard_library():
    R = _readers()
    from volta_amd._lib import VoltaHipError
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 3, 4, 5, 36 * 4, 36 * 2048):
        a = rng.standard_normal(n).astype(np.float32)
        text = base64.b64encode(a.tobytes())
        assert np.array_equal(R.b64_to_array(text), a)
        assert np.array_equal(R.b64_to_array(text.decode()), a)
        assert np.array_equal(R.b64_to_array(text.rstrip(b"=")), a)                    # padding optional
        assert np.array_equal(R.b64_to_array(base64.urlsafe_b64encode(a.tobytes())), a)
        assert np.array_equal(R.b64_to_array(base64.encodebytes(a.tobytes())), a)      # line breaks every 76 characters
    with pytest.raises(VoltaHipError, match="not base64"):
        R.b64_to_array(b"AAAA*AAA")
    with pytest.raises(ValueError):
        R.b64_to_array(base64.b64encode(b"12345"))                                     # 5 bytes are not whole float32s


def _datapoint(rng, n, F=24, Cn=11, A=7, as_text=False):
    w, h = int(rng.integers(300, 800)), int(rng.integers(300, 800))
    dp = [rng.standard_normal((n, F)).astype(np.float32), rng.random((n, Cn)).astype(np.float32), rng.integers(0, 1600, n).astype(np.int64),
          rng.random(n).astype(np.float32), rng.integers(0, 400, n).astype(np.int64), rng.random(n).astype(np.float32),
          rng.random((n, A)).astype(np.float32), (rng.random((n, 4)) * 300).astype(np.float32),
          str(n) if as_text else n, str(h) if as_text else h, str(w) if as_text else w, str(int(rng.integers(1, 10 ** 9))),
          "a caption with an accent: café #%d" % n]
    return dp


def test_concap_record_reader_decodes_every_field_into_its_slot(tmp_path):
    """tensorpack-style store (key %08d -> msgpack datapoint, plus `__keys__`): batches of 4 out of 10 ragged records; rows beyond
    num_boxes are zero; num_boxes / img_h / img_w arrive as python ints, as text (the TSV hands them over as strings) or as numpy scalars;
    float64 features and int32 labels are converted."""
    R = _readers()
    rng = np.random.default_rng(9)
    Rl, F, Cn, A = 9, 24, 11, 7
    dps = []
    for i in range(10):
        dp = _datapoint(rng, int(rng.integers(1, Rl + 1)) if i else Rl, F, Cn, A, as_text=bool(i % 2))
        if i == 3:
            dp[0], dp[2], dp[8], dp[9] = dp[0].astype(np.float64), dp[2].astype(np.int32), np.int64(dp[0].shape[0]), np.float32(int(dp[9]))
        dps.append(dp)
    recs = {b"%08d" % i: pack_datapoint(dp, str_keys=(i == 5)) for i, dp in enumerate(dps)}
    recs[b"__keys__"] = pack_datapoint([k for k in recs])
    path = str(tmp_path / "training_feat_all.lmdb")
    write_lmdb(path, recs)
    rd = R.ConceptCapRecordReader(path, 4, region_len=Rl, feature_size=F, num_classes=Cn, num_attrs=A, with_labels=True, pin_memory=False)
    assert rd.num_records == 10 and len(rd) == 3
    seen = 0
    for batch in rd:
        B = len(batch["image_id"])
        assert B == (4 if seen < 8 else 2)
        for b in range(B):
            dp = dps[seen + b]
            n = dp[0].shape[0]
            assert int(batch["num_boxes"][b]) == n
            assert batch["img_wh"][b].tolist() == [float(dp[10]), float(dp[9])]
            assert batch["image_id"][b] == dp[11] and batch["caption"][b] == dp[12]
            for name, src in (("feat", dp[0]), ("cls", dp[1]), ("attr_scores", dp[6]), ("boxes", dp[7]), ("obj_labels", dp[2]), ("obj_confs", dp[3]),
                              ("attr_labels", dp[4]), ("attr_confs", dp[5])):
                got = batch[name][b].numpy()
                assert np.array_equal(got[:n], src.astype(got.dtype)), name
                assert not got[n:].any(), name
        seen += B
    assert seen == 10
    assert len(R.ConceptCapRecordReader(path, 4, region_len=Rl, feature_size=F, num_classes=Cn, num_attrs=A, drop_last=True, pin_memory=False)) == 2


def test_concap_record_decode_rejects_what_the_reference_would_choke_on(tmp_path):
    R = _readers()
    from volta_amd._lib import VoltaHipError
    rng = np.random.default_rng(2)

    def first_batch(dp, **kw):
        path = str(tmp_path / ("r%d.lmdb" % int(rng.integers(1 << 30))))
        write_lmdb(path, {b"00000000": pack_datapoint(dp) if not isinstance(dp, bytes) else dp})
        args = dict(region_len=6, feature_size=24, num_classes=11, num_attrs=7, pin_memory=False)
        args.update(kw)
        return next(iter(R.ConceptCapRecordReader(path, 1, **args)))

    ok = _datapoint(rng, 5)
    assert first_batch(ok)["num_boxes"].tolist() == [5]
    with pytest.raises(VoltaHipError, match="do not fit"):
        first_batch(_datapoint(rng, 7))                                   # more boxes than region_len: the reference's assignment raises
    with pytest.raises(VoltaHipError, match="13 fields"):
        first_batch(ok[:8])                                               # the 8-field rows preprocess_cc_train.py itself yields
    with pytest.raises(VoltaHipError, match="features"):
        first_batch(ok, feature_size=32)                                  # width mismatch
    with pytest.raises(VoltaHipError, match="malformed|13 fields"):
        first_batch(pack_datapoint(ok)[:200])                             # truncated record
    bad = list(ok)
    bad[8] = "five"
    with pytest.raises(VoltaHipError, match="not numbers"):
        first_batch(bad)
    # a bad record inside a batch decoded on several threads: the error names the cause, whichever thread met it
    path = str(tmp_path / "mixed.lmdb")
    write_lmdb(path, {b"%08d" % i: pack_datapoint(_datapoint(rng, 7 if i == 3 else 4)) for i in range(6)})
    rd = R.ConceptCapRecordReader(path, 6, region_len=6, feature_size=24, num_classes=11, num_attrs=7, pin_memory=False, threads=4)
    with pytest.raises(VoltaHipError, match="7 boxes do not fit"):
        next(iter(rd))


def test_extraction_tsv_rows(tmp_path):
    """The 13-column TSV the detector writes (preprocess_cc_train.py:8-10): base64 fp32 columns come back as arrays."""
    R = _readers()
    rng = np.random.default_rng(4)
    rows, lines = [], []
    for i in range(3):
        n = int(rng.integers(1, 6))
        boxes, feats, cls = rng.random((n, 4)).astype(np.float32), rng.random((n, 16)).astype(np.float32), rng.random((n, 5)).astype(np.float32)
        rows.append((i, boxes, feats, cls))
        e = lambda a: base64.b64encode(a.tobytes()).decode()
        lines.append("\t".join([str(100 + i), "480", "640", "x", "x", "x", "x", str(n), e(boxes), e(feats), e(cls), "x", "x"]))
    p = tmp_path / "train_obj36-36.tsv"
    p.write_text("\n".join(lines) + "\n")
    got = list(R.read_extraction_tsv(str(p), feature_size=16, num_classes=5))
    assert len(got) == 3
    for (i, boxes, feats, cls), g in zip(rows, got):
        assert g["img_id"] == str(100 + i) and (g["img_h"], g["img_w"], g["num_boxes"]) == (480, 640, boxes.shape[0])
        assert np.array_equal(g["boxes"], boxes) and np.array_equal(g["features"], feats) and np.array_equal(g["cls_prob"], cls)


def test_image_features_reader_matches_the_reference_class(tmp_path):
    """ImageFeaturesH5Reader over `vk_lmdb_*` against tests/golden/feature_reader.npz, written by the REAL reference class
    (oracle/make_golden.py:write_feature_reader): values, row order and dtypes, for 4 / 5 box columns x no / first / last global row x
    in-memory or not."""
    R = _readers()
    from oracle.make_golden import feature_store_records
    recs = feature_store_records()
    store = {k.encode(): pickle.dumps(v) for k, v in recs.items()}
    store[b"keys"] = pickle.dumps([k.encode() for k in recs])
    path = str(tmp_path / "flickr30k_feat.lmdb")
    write_lmdb(path, store)
    gold = np.load(os.path.join(GOLD, "feature_reader.npz"))
    for nl in (4, 5):
        for glob in (None, "first", "last"):
            for mem in (False, True):
                cfg = types.SimpleNamespace(v_feature_size=16, num_locs=nl, add_global_imgfeat=glob)
                rd = R.ImageFeaturesH5Reader(path, cfg, in_memory=mem)
                assert len(rd) == len(recs)
                for k in recs:
                    for _ in range(2 if mem else 1):
                        f, n, loc, ori = rd[int(k)]                       # callers pass ints: str(image_id).encode()
                    tag = "l%d_%s_%d_%s" % (nl, glob, mem, k)
                    for name, got in (("features", f), ("num", np.array(n)), ("loc", loc), ("ori", ori)):
                        want = gold[tag + "_" + name]
                        assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), (tag, name)
                with pytest.raises(ValueError):
                    rd["no-such-image"]


def test_loader_file_rule_and_host_side_batching(tmp_path):
    """ConceptCapLoaderTrain's host half without a GPU: the per-rank file rule of the reference (:195-201), the corpus table with the batch's
    own captions behind it, and that the replacement range is the corpus only."""
    R = _readers()
    import json
    rng = np.random.default_rng(1)
    dps = [_datapoint(rng, 6, 2048, 1601, 401) for _ in range(3)]
    write_lmdb(str(tmp_path / "training_feat_part_1.lmdb"), {b"%08d" % i: pack_datapoint(dp) for i, dp in enumerate(dps)})
    (tmp_path / "caption_train.json").write_text(json.dumps({dp[11]: dp[12] for dp in dps} | {"extra": "one more caption"}))

    class Tok:
        vocab_size, cls_token_id, sep_token_id, mask_token_id = 3000, 101, 102, 103
        def encode(self, text, add_special_tokens=False):
            assert add_special_tokens is False
            return [1000 + len(w) for w in text.split()]

    ld = R.ConceptCapLoaderTrain(str(tmp_path), str(tmp_path), Tok(), seq_len=12, batch_size=2, region_len=6, add_global_imgfeat="first", rank=1,
                                 device="cpu")
    assert len(ld) == 3 and ld.n_corpus == 4 and ld.producer.n_random == 3
    assert ld.producer.cap_tokens.shape == (4 + 2, 12) and ld.producer.cap_len.tolist()[:4] == [len(dp[12].split()) for dp in dps] + [3]
    with pytest.raises(Exception):
        R.ConceptCapLoaderTrain(str(tmp_path), str(tmp_path), Tok(), seq_len=12, batch_size=2, region_len=6, device="cpu")   # no training_feat_all.lmdb
    # the validation loader: its own file names, global feature first for any truthy flag, the visualization switch reaches the producer
    write_lmdb(str(tmp_path / "validation_feat_all.lmdb"), {b"%08d" % i: pack_datapoint(dp) for i, dp in enumerate(dps[:2])})
    (tmp_path / "caption_valid.json").write_text(json.dumps({dp[11]: dp[12] for dp in dps[:2]}))
    val = R.ConceptCapLoaderVal(str(tmp_path), str(tmp_path), Tok(), seq_len=12, batch_size=2, region_len=6, add_global_imgfeat="last", visualization=True,
                                device="cpu")
    assert len(val) == 2 and val.add_global_imgfeat == "first" and val.producer.add_global == 1 and val.producer.visualization == 1 and val.window == 1
    assert R.ConceptCapLoaderVal(str(tmp_path), str(tmp_path), Tok(), seq_len=12, batch_size=2, region_len=6, add_global_imgfeat=None,
                                 device="cpu").producer.visualization == 0


def test_damaged_stores_raise_instead_of_crashing(tmp_path):
    """Byte flips in page headers / offset tables / nodes and truncated files: every call either works or raises VoltaHipError -- the walker
    never leaves the mapping (tools/fuzz/run.sh drives the same mutations under AddressSanitizer; an entry count read from a damaged
    header once sent it 60 KB past the page)."""
    import random
    R = _readers()
    from volta_amd._lib import VoltaHipError
    recs = _records(60)
    good = str(tmp_path / "good.lmdb")
    write_lmdb(good, recs, max_keys=4)
    orig = open(good, "rb").read()
    rnd = random.Random(3)
    outcomes = {"ok": 0, "error": 0}
    for it in range(250):
        buf = bytearray(orig)
        for _ in range(rnd.randint(1, 6)):
            pg = rnd.randrange(len(buf) // 4096)
            buf[pg * 4096 + (rnd.randrange(64) if rnd.random() < 0.7 else rnd.randrange(4096))] = rnd.randrange(256)
        if it % 10 == 0:
            buf = buf[:4096 * rnd.randrange(2, len(buf) // 4096)]
        bad = str(tmp_path / "bad.lmdb")
        with open(bad, "wb") as f:
            f.write(buf)
        try:
            db = R.LMDBReader(bad)
            for n, (k, v) in enumerate(db):
                bytes(v[-4:])
                if n > 500:
                    break
            for k in list(recs)[:8]:
                v = db.get(k)
                if v is not None:
                    bytes(v[-4:])
            db.close()
            outcomes["ok"] += 1
        except VoltaHipError:
            outcomes["error"] += 1
    assert outcomes["ok"] > 0 and outcomes["error"] > 0
