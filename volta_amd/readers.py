"""Record readers in front of the device-side batch producer: the LMDB feature stores and the extraction
TSV of the reference, read through `libvolta_hip.so`'s host entry points (`csrc/records.cpp`) -- memory-mapped, decoded straight
into pinned staging buffers, one host-to-device copy per batch.

  LMDBReader               `lmdb.open(path, readonly=True, lock=False)` + `txn.get` / cursor iteration
  ImageFeaturesH5Reader    volta/datasets/_image_features_reader.py:16-196 (same constructor, `len`, `reader[image_id]`)
  read_extraction_tsv      the rows data/conceptual_captions/preprocess_cc_train.py:57-72 yields
  ConceptCapRecordReader   tensorpack's `LMDBSerializer.load(lmdb_file, shuffle=False)` + batching: raw datapoints -> staging arrays
  ConceptCapLoaderTrain    volta/datasets/concept_cap_dataset.py:139-288: records -> the tensors the pre-training step takes
  ConceptCapLoaderVal      volta/datasets/concept_cap_dataset.py:291-400
  WordPieceTokenizer       the `tokenizer.encode(caption)` of :465 (pytorch-transformers' BertTokenizer) in native code

No `lmdb`, `tensorpack`, `msgpack_numpy` or `zmq` is needed."""
import contextlib
import ctypes as C
import json
import os
import pickle

import numpy as np
import torch

from . import _lib as L


class LMDBReader:
    """Read-only view of an LMDB data file (or of the directory that holds `data.mdb`).  Values are zero-copy memoryviews of the
    mapping, valid until `close()`."""

    def __init__(self, path):
        h = C.c_void_p()
        L.check(L.lib.vk_lmdb_open(os.fsencode(path), C.byref(h)))
        self._h = h
        self.path = path

    def __len__(self):
        return int(L.lib.vk_lmdb_entries(self._handle()))

    @staticmethod
    def _view(p, n):
        return memoryview((C.c_ubyte * n.value).from_address(p.value)) if n.value else memoryview(b"")

    def get(self, key, default=None):
        if isinstance(key, str):
            key = key.encode()
        p, n = C.c_void_p(), C.c_size_t()
        rc = L.lib.vk_lmdb_get(self._handle(), key, len(key), C.byref(p), C.byref(n))
        if rc < 0:
            L.check(rc)
        return self._view(p, n) if rc == 1 else default

    def _handle(self):
        if self._h is None:
            raise ValueError("LMDBReader is closed")
        return self._h

    def iter_raw(self):
        """(key bytes, value address, value length) in key order: what the native decoders take."""
        h = self._handle()
        L.check(L.lib.vk_lmdb_first(h))
        kp, kn, vp, vn = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        while True:
            rc = L.lib.vk_lmdb_next(h, C.byref(kp), C.byref(kn), C.byref(vp), C.byref(vn))
            if rc < 0:
                L.check(rc)
            if rc == 0:
                return
            yield C.string_at(kp.value, kn.value), vp.value, vn.value

    def __iter__(self):
        """(key bytes, value memoryview) in key order, like an `lmdb` cursor."""
        for key, addr, n in self.iter_raw():
            yield key, (memoryview((C.c_ubyte * n).from_address(addr)) if n else memoryview(b""))

    def close(self):
        if self._h is not None:
            L.lib.vk_lmdb_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def b64_to_array(text, dtype=np.float32):
    """base64 text (str / bytes) -> 1-D numpy array, decoded by `vk_b64_decode` into the array's own buffer."""
    if isinstance(text, str):
        text = text.encode("ascii")
    out = np.empty((len(text) // 4 + 1) * 3, dtype=np.uint8)
    n = C.c_size_t()
    L.check(L.lib.vk_b64_decode(text, len(text), C.c_void_p(out.ctypes.data), out.nbytes, C.byref(n)))
    item = np.dtype(dtype).itemsize
    if n.value % item:
        raise ValueError("base64 payload of %d bytes is not a whole number of %s" % (n.value, np.dtype(dtype)))
    return out[:n.value].view(dtype)


class ImageFeaturesH5Reader:
    """The reference's task-feature reader (volta/datasets/_image_features_reader.py:16-196) over `LMDBReader`: key `keys` holds
    the pickled list of image ids; each id maps to a pickled dict with `img_h`, `img_w` and base64 fp32 `features` [n, v_feature_size] /
    `boxes` [n, 4].  `reader[image_id]` -> (features, num_boxes, image_location, image_location_ori) with the global feature (mean of
    the regions) and its whole-image box added first / last as the config asks -- the dtypes follow the reference's numpy promotions
    (locations become float64 once the integer global box is concatenated)."""

    def __init__(self, features_path, config, in_memory=False):
        self.features_path = features_path
        self._in_memory = in_memory
        self.env = LMDBReader(features_path)
        keys = self.env.get(b"keys")
        if keys is None:
            raise KeyError("%s has no `keys` record" % features_path)
        self._image_ids = pickle.loads(keys)
        self._index = {k: i for i, k in reversed(list(enumerate(self._image_ids)))}       # first occurrence, as list.index
        self._cache = [None] * len(self._image_ids)
        self.feature_size = config.v_feature_size
        self.num_locs = config.num_locs
        self.add_global_imgfeat = config.add_global_imgfeat

    def __len__(self):
        return len(self._image_ids)

    def keys(self):
        return self._image_ids

    def __getitem__(self, image_id):
        image_id = str(image_id).encode()
        if image_id not in self._index:
            raise ValueError("%r is not in list" % image_id)
        index = self._index[image_id]
        if self._in_memory and self._cache[index] is not None:
            return self._cache[index]
        raw = self.env.get(image_id)
        if raw is None:
            raise KeyError(image_id)
        item = pickle.loads(raw)
        h, w = int(item["img_h"]), int(item["img_w"])
        features = b64_to_array(item["features"]).reshape(-1, self.feature_size)
        boxes = b64_to_array(item["boxes"]).reshape(-1, 4)
        n, nl = boxes.shape[0], self.num_locs
        loc = np.zeros((n, nl), dtype=np.float32)
        loc[:, :4] = boxes
        if nl == 5:
            loc[:, 4] = (loc[:, 3] - loc[:, 1]) * (loc[:, 2] - loc[:, 0]) / (float(w) * float(h))
        loc_ori = loc.copy()
        loc[:, 0] /= float(w)
        loc[:, 1] /= float(h)
        loc[:, 2] /= float(w)
        loc[:, 3] /= float(h)
        num_boxes = features.shape[0]
        if self.add_global_imgfeat in ("first", "last"):
            g_feat = (np.sum(features, axis=0) / num_boxes)[None]
            g_loc = np.array([[0, 0, 1, 1] + [1] * (nl - 4)])                    # integer rows: the concatenation promotes to float64
            g_ori = np.array([[0, 0, w, h] + [w * h] * (nl - 4)])
            first = self.add_global_imgfeat == "first"
            features = np.concatenate([g_feat, features] if first else [features, g_feat], axis=0)
            loc = np.concatenate([g_loc, loc] if first else [loc, g_loc], axis=0)
            loc_ori = np.concatenate([g_ori, loc_ori] if first else [loc_ori, g_ori], axis=0)
            num_boxes += 1
        out = (features, num_boxes, loc, loc_ori)
        if self._in_memory:
            self._cache[index] = out
        return out


TSV_FIELDNAMES = ["img_id", "img_h", "img_w", "objects_id", "objects_conf", "attrs_id", "attrs_conf", "num_boxes", "boxes", "features",
                  "cls_prob", "attrs", "classes"]


def read_extraction_tsv(path, feature_size=2048, num_classes=1601):
    """Rows of the detector's extraction TSV (data/conceptual_captions/preprocess_cc_train.py:8-10,57-72): yields dicts with
    `img_id`, `img_h`, `img_w`, `num_boxes` and the decoded fp32 arrays `boxes` [n, 4], `features` [n, F], `cls_prob` [n, C]."""
    with open(path, "rb") as f:
        for line in f:
            cols = line.rstrip(b"\r\n").split(b"\t")
            if len(cols) < 11:
                raise ValueError("%s: a row has %d columns, the layout has %d" % (path, len(cols), len(TSV_FIELDNAMES)))
            item = dict(zip(TSV_FIELDNAMES, cols))
            n = int(item["num_boxes"])
            yield dict(img_id=item["img_id"].decode(), img_h=int(item["img_h"]), img_w=int(item["img_w"]), num_boxes=n,
                       boxes=b64_to_array(item["boxes"]).reshape(n, 4), features=b64_to_array(item["features"]).reshape(n, feature_size),
                       cls_prob=b64_to_array(item["cls_prob"]).reshape(n, num_classes))


class ConceptCapRecordReader:
    """Batches of raw Conceptual Captions datapoints out of a tensorpack-serialised LMDB (`LMDBSerializer.save`: key = b"%08d" -> msgpack
    of the 13-field datapoint, plus a `__keys__` record that is skipped).  Every field is decoded by `vk_concap_record_decode` into its slot
    of pinned staging arrays shaped for `ConceptCapBatchProducer`: feat [B, R, F], cls [B, R, C], boxes [B, R, 4] (pixels), num_boxes [B],
    img_wh [B, 2], and -- `with_labels` -- obj / attr labels and confidences [B, R], attr scores [B, R, A].  A batch is decoded by ONE native call on
    `threads` host threads (`vk_concap_records_decode`).  `staging_sets` staging sets rotate so
    the copy of batch i can overlap the decode of batch i + 1.  The last batch may be smaller unless `drop_last`."""

    def __init__(self, path, batch_size, region_len=36, feature_size=2048, num_classes=1601, num_attrs=401, with_labels=False,
                 drop_last=False, pin_memory=None, threads=None, staging_sets=2):
        self.db = LMDBReader(path)
        self.B, self.R, self.F, self.Cn, self.A = int(batch_size), int(region_len), int(feature_size), int(num_classes), int(num_attrs)
        self.with_labels, self.drop_last = with_labels, drop_last
        self.threads = int(threads) if threads else max(1, min(8, len(os.sched_getaffinity(0)) // 2))
        self.num_records = len(self.db) - (1 if self.db.get(b"__keys__") is not None else 0)
        pin = torch.cuda.is_available() if pin_memory is None else pin_memory
        self._sets = [self._staging(pin) for _ in range(max(2, int(staging_sets)))]

    def _staging(self, pin):
        B, R = self.B, self.R
        mk = lambda shape, dt: torch.zeros(shape, dtype=dt, pin_memory=pin)
        s = dict(feat=mk((B, R, self.F), torch.float32), cls=mk((B, R, self.Cn), torch.float32), boxes=mk((B, R, 4), torch.float32),
                 num_boxes=mk((B,), torch.int32), img_wh=mk((B, 2), torch.float32))
        if self.with_labels:
            s.update(obj_labels=mk((B, R), torch.int64), obj_confs=mk((B, R), torch.float32), attr_labels=mk((B, R), torch.int64),
                     attr_confs=mk((B, R), torch.float32), attr_scores=mk((B, R, self.A), torch.float32))
        return s

    def __len__(self):
        return self.num_records // self.B if self.drop_last else -(-self.num_records // self.B)

    def _decode_batch(self, addrs, lens, s):
        """All records of a batch in one native call, on `self.threads` host threads."""
        n = len(addrs)
        at = lambda name, b: C.c_void_p(s[name][b].data_ptr()) if name in s else None
        slots = (L.ConcapRecord * n)()
        for b in range(n):
            slots[b] = L.ConcapRecord(at("feat", b), at("cls", b), at("attr_scores", b), at("boxes", b), at("obj_labels", b), at("obj_confs", b),
                                      at("attr_labels", b), at("attr_confs", b), self.R, self.F, self.Cn, self.A)
        recs, ln = (C.c_void_p * n)(*addrs), (C.c_size_t * n)(*lens)
        L.check(L.lib.vk_concap_records_decode(recs, ln, slots, n, self.threads, None))
        ids, caps = [], []
        for b in range(n):
            s["num_boxes"][b] = slots[b].num_boxes
            s["img_wh"][b, 0], s["img_wh"][b, 1] = slots[b].img_w, slots[b].img_h
            ids.append(slots[b].image_id.decode())
            caps.append(C.string_at(slots[b].caption, slots[b].caption_len).decode("utf-8"))
        return ids, caps

    def __iter__(self):
        which, addrs, lens = 0, [], []
        for key, addr, n in self.db.iter_raw():
            if key == b"__keys__":
                continue
            addrs.append(addr)
            lens.append(n)
            if len(addrs) == self.B:
                s = self._sets[which]
                ids, caps = self._decode_batch(addrs, lens, s)
                yield dict(s, image_id=ids, caption=caps)
                which, addrs, lens = (which + 1) % len(self._sets), [], []
        if addrs and not self.drop_last:
            s = self._sets[which]
            ids, caps = self._decode_batch(addrs, lens, s)
            yield dict({k: v[:len(ids)] for k, v in s.items()}, image_id=ids, caption=caps)


class WordPieceTokenizer:
    """BERT's tokenizer in native code (`csrc/wordpiece.cpp`: basic tokenizer + WordPiece over a `vocab.txt`), with the handful of attributes
    the loaders read from a tokenizer.  `encode(text)` -> ids without [CLS] / [SEP] (what pytorch-transformers 1.1's `BertTokenizer.encode`
    returned, concept_cap_dataset.py:465); `encode_batch(texts, ld)` -> (ids [n, ld] int32 zero-padded / truncated, counts [n]) in one call on
    several host threads."""

    def __init__(self, vocab_file, do_lower_case=True, threads=None):
        h = C.c_void_p()
        L.check(L.lib.vk_wordpiece_open(os.fsencode(vocab_file), int(bool(do_lower_case)), C.byref(h)))
        self._h = h
        self.vocab_file = vocab_file
        self.vocab_size = int(L.lib.vk_wordpiece_vocab_size(h))
        self.threads = int(threads) if threads else max(1, min(8, len(os.sched_getaffinity(0)) // 2))
        for name in ("unk", "cls", "sep", "mask", "pad"):
            setattr(self, name + "_token", "[%s]" % name.upper())
            i = self.convert_tokens_to_ids("[%s]" % name.upper())
            setattr(self, name + "_token_id", i)

    def __len__(self):
        return self.vocab_size

    def convert_tokens_to_ids(self, token):
        if isinstance(token, (list, tuple)):
            return [self.convert_tokens_to_ids(t) for t in token]
        i = int(L.lib.vk_wordpiece_token_id(self._h, token.encode("utf-8")))
        return i if i >= 0 else self.unk_token_id

    def encode(self, text, add_special_tokens=False):
        raw = text.encode("utf-8", "replace")
        cap = 2 * len(raw) + 8
        buf = (C.c_int32 * cap)()
        n = L.lib.vk_wordpiece_encode(self._h, raw, len(raw), buf, cap)
        if n < 0:
            L.check(n)
        ids = list(buf[:n])
        return [self.cls_token_id] + ids + [self.sep_token_id] if add_special_tokens else ids

    def encode_batch(self, texts, ld):
        n = len(texts)
        raws = [t.encode("utf-8", "replace") for t in texts]
        ids, counts = torch.zeros(n, ld, dtype=torch.int32), torch.zeros(n, dtype=torch.int32)
        if n:
            arr, lens = (C.c_char_p * n)(*raws), (C.c_size_t * n)(*[len(r) for r in raws])
            L.check(L.lib.vk_wordpiece_encode_batch(self._h, arr, lens, n, C.c_void_p(ids.data_ptr()), ld, C.c_void_p(counts.data_ptr()), self.threads))
        return ids, counts

    def __del__(self):
        try:
            if self._h is not None:
                L.lib.vk_wordpiece_close(self._h)
                self._h = None
        except Exception:
            pass


def _encoder(tokenizer):
    """caption -> token ids without [CLS] / [SEP] (what pytorch-transformers 1.1's `tokenizer.encode` returned, :465)."""
    def encode(text):
        try:
            return list(tokenizer.encode(text, add_special_tokens=False))
        except TypeError:
            return list(tokenizer.encode(text))
    return encode


class ConceptCapLoaderTrain:
    """Counterpart of the reference's ConceptCapLoaderTrain (volta/datasets/concept_cap_dataset.py:139-400): same constructor arguments
    where they still mean something, same per-rank file rule (`training_feat_part_<rank>.lmdb`, else `training_feat_all.lmdb`, :195-201),
    same batch contract -- iterating yields the 15 tensors of :270-286 (on the GPU) plus the list of image ids.

    Pipeline: `ConceptCapRecordReader` (host, native decode into pinned memory) -> one asynchronous copy per array -> `ConceptCapBatchProducer`
    (`vk_concap_batch`: caption swap, token / region masking with IoU co-masking, box normalisation, global feature, on the device).
    `tokenizer.encode(caption)` is the only per-record Python left (it runs on the prefetch thread); the corpus of `caption_train.json` is tokenised once.  Sampling decisions
    come from Philox streams of (`seed`, batch index) instead of Python's `random` -- same distribution, different draws.  tensorpack's
    `LocallyShuffleData(ds, cache)` becomes a shuffle of whole batches inside a window of `cache // batch_size` batches."""

    _caption_file, _visualization = "caption_train.json", False

    @staticmethod
    def _lmdb_name(rank):
        return "training_feat_part_%d.lmdb" % rank if rank is not None else "training_feat_all.lmdb"

    def __init__(self, annotations_path, features_path, tokenizer, bert_model=None, seq_len=36, batch_size=512, num_workers=0, cache=0,
                 local_rank=-1, objective=0, num_locs=5, add_global_imgfeat=None, region_len=36, vocab_size=None, seed=0, device="cuda",
                 rank=None, prefetch=2):
        import torch.distributed as dist
        from .data import ConceptCapBatchProducer
        if rank is None and local_rank != -1 and dist.is_available() and dist.is_initialized():
            rank = dist.get_rank()
        name = self._lmdb_name(rank)
        self.prefetch = max(0, int(prefetch))       # batches decoded ahead by a background thread (the native decode releases the GIL)
        self.records = ConceptCapRecordReader(os.path.join(features_path, name), batch_size, region_len=region_len, with_labels=True,
                                              threads=num_workers or None, staging_sets=self.prefetch + 2)
        self.num_dataset = self.records.num_records
        with open(os.path.join(annotations_path, self._caption_file)) as f:
            corpus = list(json.load(f).values())
        self.tokenizer, self.encode = tokenizer, _encoder(tokenizer)
        self.n_corpus = len(corpus)
        vocab = vocab_size if vocab_size is not None else getattr(tokenizer, "vocab_size", None) or len(tokenizer.vocab)
        ids = {n: getattr(tokenizer, n + "_token_id", None) for n in ("cls", "sep", "mask")}
        ids = {n: (v if v is not None else tokenizer.vocab["[%s]" % n.upper()]) for n, v in ids.items()}
        # table rows [0, n_corpus): the corpus; rows [n_corpus, n_corpus + B): this batch's own captions, rewritten every batch.
        # A replacement caption is captions[randint(0, num_caps - 1)] with num_caps = the number of records (:421,531-532)
        self.seq_len, self.batch_size, self.num_locs = seq_len, batch_size, num_locs
        self.producer = ConceptCapBatchProducer([self.encode(c) for c in corpus], seq_len, region_len, vocab, add_global_imgfeat=add_global_imgfeat,
                                                objective=objective, cls_id=ids["cls"], sep_id=ids["sep"], mask_id=ids["mask"], device=device,
                                                extra_rows=batch_size, min_ld=seq_len, n_random=max(1, min(self.num_dataset, self.n_corpus)),
                                                visualization=self._visualization)
        self.add_global_imgfeat, self.objective, self.seed, self.device = add_global_imgfeat, objective, int(seed), device
        self.window = max(1, int(cache) // max(1, batch_size))
        self.epoch = 0
        self._stream = None

    def __len__(self):
        return self.num_dataset

    def _produce(self, raw, step):
        """One batch: copies and the producer's kernels run on the loader's own stream, so they overlap the training step that is still running
        on the caller's stream; the host waits for THAT stream only (the pinned staging set is rewritten a few batches later), and the batch's
        tensors are handed to the caller's stream (`record_stream`: the allocator must not recycle them under the model's kernels)."""
        on_gpu = torch.cuda.is_available() and str(self.device).startswith("cuda")
        if on_gpu and self._stream is None:
            self._stream = torch.cuda.Stream()
        with (torch.cuda.stream(self._stream) if on_gpu else contextlib.nullcontext()):
            batch = self._produce_on_current_stream(raw, step)
        if on_gpu:
            self._stream.synchronize()
            user = torch.cuda.current_stream()
            for t in batch:
                t.record_stream(user)
        return batch + (raw["image_id"],)

    def _produce_on_current_stream(self, raw, step):
        dev, B = self.device, len(raw["image_id"])
        own, lens = raw["tokens"] if "tokens" in raw else self._tokenise(raw["caption"])
        self.producer.cap_tokens[self.n_corpus:self.n_corpus + B].copy_(own, non_blocking=True)
        self.producer.cap_len[self.n_corpus:self.n_corpus + B].copy_(lens, non_blocking=True)
        up = {k: raw[k].to(dev, non_blocking=True) for k in ("feat", "cls", "boxes", "num_boxes", "img_wh", "obj_labels", "obj_confs",
                                                             "attr_labels", "attr_confs", "attr_scores")}
        cap_index = torch.arange(self.n_corpus, self.n_corpus + B, dtype=torch.int32, device=dev)
        out = self.producer(up["feat"], up["cls"], up["boxes"], up["num_boxes"], up["img_wh"], cap_index, self.seed * 1000003 + step)
        if self.num_locs == 4:
            out["image_loc"] = out["image_loc"][..., :4].contiguous()
        return (out["input_ids"], out["input_mask"], out["segment_ids"], out["lm_label_ids"], out["is_match"], out["image_feat"], out["image_loc"],
                out["image_cls"], up["obj_labels"], up["obj_confs"], up["attr_labels"], up["attr_confs"], up["attr_scores"], out["image_label"],
                out["image_mask"])

    def _tokenise(self, captions):
        """This batch's captions -> (ids [B, ld] int32, lengths [B] int32): one native call with a `WordPieceTokenizer`, `tokenizer.encode` per
        caption otherwise."""
        ld = self.producer.cap_tokens.shape[1]
        if isinstance(self.tokenizer, WordPieceTokenizer):
            return self.tokenizer.encode_batch(captions, ld)
        toks = [self.encode(c)[:ld] for c in captions]
        own = torch.zeros(len(toks), ld, dtype=torch.int32)
        for i, t in enumerate(toks):
            own[i, :len(t)] = torch.tensor(t, dtype=torch.int32)
        return own, torch.tensor([len(t) for t in toks], dtype=torch.int32)

    def _raw_batches(self):
        """The record reader's batches, decoded up to `prefetch` batches ahead on a background thread.  A staging set is rewritten
        `prefetch + 2` batches after it was handed out: one batch is with the consumer, one is being decoded, `prefetch` wait in the queue."""
        if not self.prefetch:
            yield from self.records
            return
        import queue
        import threading
        q, stop = queue.Queue(maxsize=self.prefetch), threading.Event()

        def put(x):
            while not stop.is_set():
                try:
                    q.put(x, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def fill():
            try:
                for raw in self.records:
                    raw["tokens"] = self._tokenise(raw["caption"])      # tokenised here, off the thread that issues the training step
                    if not put(raw):
                        return
                put(None)
            except BaseException as e:          # handed to the consumer, raised there
                put(e)

        th = threading.Thread(target=fill, daemon=True)
        th.start()
        try:
            while True:
                raw = q.get()
                if raw is None:
                    return
                if isinstance(raw, BaseException):
                    raise raw
                yield raw
        finally:
            stop.set()
            th.join()

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + 7919 * self.epoch)
        self.epoch += 1
        step = 0
        if self.window <= 1:
            for raw in self._raw_batches():
                yield self._produce(raw, step)
                step += 1
            return
        pool = []
        for raw in self._raw_batches():
            pool.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in raw.items()})
            if len(pool) == self.window:
                yield self._produce(pool.pop(int(torch.randint(len(pool), (1,), generator=g))), step)
                step += 1
        while pool:
            yield self._produce(pool.pop(int(torch.randint(len(pool), (1,), generator=g))), step)
            step += 1


class ConceptCapLoaderVal(ConceptCapLoaderTrain):
    """Counterpart of the reference's ConceptCapLoaderVal (volta/datasets/concept_cap_dataset.py:291-400): `validation_feat_all.lmdb` and
    `caption_valid.json`, records in file order (no shuffle window, no per-rank shards), and -- `visualization=True` -- neither caption swaps nor
    token / region masking (:514,622,652).  As in the reference (:359) ANY truthy `add_global_imgfeat` puts the global feature FIRST, whatever the
    config says about "last"."""
    _caption_file = "caption_valid.json"

    @staticmethod
    def _lmdb_name(rank):
        return "validation_feat_all.lmdb"

    def __init__(self, annotations_path, features_path, tokenizer, bert_model=None, seq_len=36, batch_size=512, num_workers=0, cache=5000, objective=0,
                 num_locs=5, add_global_imgfeat=True, visualization=False, **kw):
        self._visualization = bool(visualization)
        super().__init__(annotations_path, features_path, tokenizer, bert_model, seq_len=seq_len, batch_size=batch_size, num_workers=num_workers, cache=0,
                         objective=objective, num_locs=num_locs, add_global_imgfeat="first" if add_global_imgfeat else None, **kw)
