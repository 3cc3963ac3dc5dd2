"""Input-side rate on the GPU box: LMDB records -> ConceptCapLoaderTrain (native decode into pinned memory, host-to-device copies, device-side batch
producer) -> optionally the training step.  Writes a synthetic store with tests/lmdb_writer.py first (one record repeated: the decode cost does not
depend on the values).  The PCIe-inclusive figure of DESIGN.md section 5; `bench.py`'s `value` never includes it.

    python tools/bench_loader.py [--records 2048] [--batch 256] [--threads 8] [--train]"""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.lmdb_writer import pack_datapoint, write_lmdb  # noqa: E402


def tokenizer(d):
    """The native WordPiece tokenizer over a synthetic vocabulary in which every caption word ("w<number>") splits into pieces."""
    from volta_amd.readers import WordPieceTokenizer
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "w"] + ["##%d" % i for i in range(10)] + ["filler%d" % i for i in range(30506)]
    path = os.path.join(d, "vocab.txt")
    with open(path, "w") as f:
        f.write("\n".join(vocab) + "\n")
    return WordPieceTokenizer(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--prefetch", type=int, default=2)
    ap.add_argument("--train", action="store_true")
    a = ap.parse_args()
    from volta_amd import readers as R
    from volta_amd.config import BertConfig
    rng = np.random.default_rng(0)
    n, Rl = a.records, 36
    d = tempfile.mkdtemp(dir="/tmp")
    boxes = (rng.random((Rl, 4)) * 300).astype(np.float32)
    boxes[:, 2:] += boxes[:, :2]
    cls = rng.random((Rl, 1601)).astype(np.float32)
    dp = [rng.random((Rl, 2048)).astype(np.float32), cls / cls.sum(1, keepdims=True), rng.integers(0, 1600, Rl), rng.random(Rl).astype(np.float32),
          rng.integers(0, 400, Rl), rng.random(Rl).astype(np.float32), rng.random((Rl, 401)).astype(np.float32), boxes, Rl, 600, 800, "1", "x"]
    recs, caps = {}, {}
    for i in range(n):
        dp[11] = str(i)
        dp[12] = " ".join("w%d" % int(x) for x in rng.integers(0, 500, int(rng.integers(5, 16))))
        caps[dp[11]] = dp[12]
        recs[b"%08d" % i] = pack_datapoint(dp)
    write_lmdb(os.path.join(d, "training_feat_all.lmdb"), recs)
    rec_bytes = len(next(iter(recs.values())))
    del recs
    with open(os.path.join(d, "caption_train.json"), "w") as f:
        json.dump(caps, f)
    cfg = BertConfig.from_json_file(os.path.join(os.path.dirname(__file__), "..", "config", "ctrl_vilbert_base.json"))
    ld = R.ConceptCapLoaderTrain(d, d, tokenizer(d), seq_len=20, batch_size=a.batch, region_len=Rl, add_global_imgfeat=cfg.add_global_imgfeat, objective=1,
                                 num_locs=cfg.num_locs, seed=3, num_workers=a.threads, prefetch=a.prefetch)
    out = {"tokenizer": "native WordPiece", "records": n, "record_bytes": rec_bytes, "batch": a.batch, "decode_threads": a.threads, "prefetch": a.prefetch, "host_cpus": len(os.sched_getaffinity(0))}
    for _ in ld:      # first epoch: page cache, allocator
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nb = 0
    for batch in ld:
        nb += len(batch[15])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["loader_only_pairs_per_s"] = nb / dt
    out["loader_only_GBps_of_records"] = nb * rec_bytes / dt / 1e9
    if a.train:
        from volta_amd.modeling import BertForVLPreTraining
        from volta_amd.optimization import AdamW, clip_grad_norm_
        torch.manual_seed(0)
        model = BertForVLPreTraining(cfg).cuda().train()
        opt = AdamW(model.parameters(), lr=1e-4)

        def step(b):
            lm, img, nsp = model(b[0], b[5], b[6], b[2], b[1], b[14], b[3], b[13], b[7], b[8], b[9], b[10], b[11], b[12], b[4])
            (lm + img + nsp).backward()
            clip_grad_norm_(model.parameters(), 5.0, defer_to_optimizer=True)
            opt.step()
            opt.zero_grad()

        for b in ld:          # warm-up epoch (plan compilation)
            if len(b[15]) == a.batch:
                step(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = 0
        for b in ld:
            if len(b[15]) == a.batch:
                step(b)
                nb += a.batch
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["train_from_lmdb_pairs_per_s"] = nb / dt
        out["train_from_lmdb_ms_per_step"] = dt * 1e3 / (nb / a.batch)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
