"""Native BERT tokenizer (csrc/wordpiece.cpp) against two implementations present in the image.  The reference's own tokenizer, pytorch-transformers
1.1's BertTokenizer, is absent; its Python classes live on as transformers.models.bert.tokenization_bert_legacy (BasicTokenizer + WordpieceTokenizer,
same lineage, later version): the closest witness, compared on every text incl. the order-sensitive ones.  The `tokenizers` package's
BertWordPieceTokenizer (Rust) is an independent second check; its normalizer strips accents BEFORE lower-casing where BERT's original (and the Python
lineage) does it after -- only characters whose lower-case form carries a new combining mark (U+0130) tell the two apart -- and it has no final-sigma
rule, so those characters are kept out of the texts compared with it."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORDS = ["a", "dog", "cat", "run", "runs", "running", "the", "grass", "cafe", "naive", "uber", "strasse", "man", "woman", "riding", "bike", "street",
         "photo", "of", "in", "on", "with", "and", "people", "table", "sitting", "2", "19", "2019", "x", "y", "z", "中", "国", "人", "山", "привет", "мир",
         "ελλας", "한", "ᄒ", "ᅡ", "ᆫ"]
PIECES = ["##s", "##ing", "##ning", "##ed", "##er", "##e", "##n", "##t", "##a", "##o", "##1", "##9", "##x", "##ир", "##ας"]
PUNCT = list(".,!?;:'\"()-[]{}/\\@#$%^&*_+=<>|~`") + ["’", "“", "”", "—", "。", "¿", "¡"]


@pytest.fixture(scope="module")
def toks(tmp_path_factory):
    from tokenizers import BertWordPieceTokenizer
    from volta_amd.readers import WordPieceTokenizer
    d = tmp_path_factory.mktemp("vocab")
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + WORDS + PIECES + PUNCT
    path = str(d / "vocab.txt")
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(vocab) + "\n")
    return WordPieceTokenizer(path), BertWordPieceTokenizer(path, lowercase=True), vocab


def test_special_ids_and_known_cases(toks):
    mine, ref, vocab = toks
    assert len(mine) == len(vocab) and (mine.pad_token_id, mine.unk_token_id, mine.cls_token_id, mine.sep_token_id, mine.mask_token_id) == (0, 1, 2, 3, 4)
    assert mine.convert_tokens_to_ids("dog") == vocab.index("dog") and mine.convert_tokens_to_ids("no-such-token") == 1
    cases = ["A dog running on the grass.", "Cafés!  naïve\tdog", "中国x dogs", "unknownword the", "a" * 120 + " dog", "dog​ cat\x00 run�",
             "", "   ", "ÜBER Straße", "2019's photo-of (man)", "한 dog", "dog cat　run the", "PRIVET Привет мир",
             "x" * 100, "x" * 101, "runningx", "dog,cat;run"]
    for s in cases:
        assert mine.encode(s) == ref.encode(s, add_special_tokens=False).ids, repr(s)
    assert mine.encode("dog cat", add_special_tokens=True) == [2, vocab.index("dog"), vocab.index("cat"), 3]


def test_random_texts_match_the_rust_tokenizer(toks):
    mine, ref, vocab = toks
    rng = np.random.default_rng(0)
    accents = ["é", "è", "ï", "ü", "ñ", "ç", "É", "Ö", "́", "̈", "ß", "Ł", "ø"]
    spaces = [" ", "  ", "\t", "\n", "\r\n", " ", " ", "　"]
    junk = ["\x00", "\x07", "\x7f", "​", "­", "�", "﻿"]
    texts = []
    for _ in range(3000):
        parts = []
        for _ in range(int(rng.integers(0, 14))):
            r = rng.random()
            if r < 0.55:
                w = WORDS[int(rng.integers(len(WORDS)))]
                if rng.random() < 0.3:
                    w = w.upper() if rng.random() < 0.5 else w.capitalize()
                if rng.random() < 0.35:
                    w += PIECES[int(rng.integers(len(PIECES)))][2:]
                if rng.random() < 0.15:
                    k = int(rng.integers(0, len(w) + 1))
                    w = w[:k] + accents[int(rng.integers(len(accents)))] + w[k:]
                if rng.random() < 0.05:
                    k = int(rng.integers(0, len(w) + 1))
                    w = w[:k] + junk[int(rng.integers(len(junk)))] + w[k:]
                parts.append(w)
            elif r < 0.75:
                parts.append(PUNCT[int(rng.integers(len(PUNCT)))])
            elif r < 0.85:
                parts.append("".join(chr(int(c)) for c in rng.integers(97, 123, int(rng.integers(1, 12)))))
            else:
                parts.append("")
            if rng.random() < 0.8:
                parts.append(spaces[int(rng.integers(len(spaces)))])
        t = "".join(parts)
        if "Σ" in t or "İ" in t:          # final-sigma context rule / the order-sensitive dotted capital I
            continue
        texts.append(t)
    bad = [(t, mine.encode(t), ref.encode(t, add_special_tokens=False).ids) for t in texts if mine.encode(t) != ref.encode(t, add_special_tokens=False).ids]
    assert not bad, (len(bad), bad[:3])
    # the batch entry point: same ids, zero padding, truncation at the row length, counts
    ids, counts = mine.encode_batch(texts[:257], 12)
    for i, t in enumerate(texts[:257]):
        want = mine.encode(t)[:12]
        assert int(counts[i]) == len(want) and ids[i, :len(want)].tolist() == want and not ids[i, len(want):].any()


def _random_texts(n, seed, with_order_sensitive=False):
    rng = np.random.default_rng(seed)
    accents = ["é", "è", "ï", "ü", "ñ", "ç", "É", "Ö", "́", "̈", "ß", "Ł", "ø"] + (["İ", "Σ", "ς"] if with_order_sensitive else [])
    spaces = [" ", "  ", "\t", "\n", "\r\n", "\u00a0", "\u2003", "\u3000"]
    junk = ["\x00", "\x07", "\x7f", "\u200b", "\u00ad", "\ufffd", "\ufeff"]
    texts = []
    for _ in range(n):
        parts = []
        for _ in range(int(rng.integers(0, 14))):
            r = rng.random()
            if r < 0.55:
                w = WORDS[int(rng.integers(len(WORDS)))]
                if rng.random() < 0.3:
                    w = w.upper() if rng.random() < 0.5 else w.capitalize()
                if rng.random() < 0.35:
                    w += PIECES[int(rng.integers(len(PIECES)))][2:]
                if rng.random() < 0.15:
                    k = int(rng.integers(0, len(w) + 1))
                    w = w[:k] + accents[int(rng.integers(len(accents)))] + w[k:]
                if rng.random() < 0.05:
                    k = int(rng.integers(0, len(w) + 1))
                    w = w[:k] + junk[int(rng.integers(len(junk)))] + w[k:]
                parts.append(w)
            elif r < 0.75:
                parts.append(PUNCT[int(rng.integers(len(PUNCT)))])
            elif r < 0.85:
                parts.append("".join(chr(int(c)) for c in rng.integers(97, 123, int(rng.integers(1, 12)))))
            else:
                parts.append("")
            if rng.random() < 0.8:
                parts.append(spaces[int(rng.integers(len(spaces)))])
        texts.append("".join(parts))
    return texts


def test_random_texts_match_the_python_lineage_of_the_reference_tokenizer(toks):
    """The reference tokenises with pytorch-transformers 1.1's BertTokenizer = BasicTokenizer + WordpieceTokenizer in Python.  That package is absent, but
    its classes live on, under the package's later name, as transformers.models.bert.tokenization_bert_legacy (same code path: clean -> CJK spacing ->
    whitespace split -> lower -> NFD + strip Mn -> punctuation split -> greedy longest-match pieces).  Same lineage, later version: the closest witness the image
    holds, and it covers what the Rust comparison had to leave out -- the dotted capital I and the capital sigma, where lower-casing BEFORE accent
    stripping (this lineage, and csrc/wordpiece.cpp) and after it differ."""
    legacy = pytest.importorskip("transformers.models.bert.tokenization_bert_legacy")
    mine, _, vocab = toks
    index = {t: i for i, t in enumerate(vocab)}
    basic = legacy.BasicTokenizer(do_lower_case=True)
    pieces = legacy.WordpieceTokenizer(vocab=index, unk_token="[UNK]")

    def ref_ids(text):
        return [index[p] for w in basic.tokenize(text) for p in pieces.tokenize(w)]
    fixed = ["İstanbul dog", "ΣΟΦΟΣ the ΟΔΟΣ", "Σ", "dogΣ cat", "A dog running on the grass.", "Cafés!  naïve\tdog", "中国x dogs", "x" * 100, "x" * 101,
             "dog\u200b cat\x00 run\ufffd", "2019's photo-of (man)", "ÜBER Straße", ""]
    texts = fixed + _random_texts(3000, 1, with_order_sensitive=True)
    bad = [(t, mine.encode(t), ref_ids(t)) for t in texts if mine.encode(t) != ref_ids(t)]
    assert not bad, (len(bad), bad[:3])


def test_unusable_vocabularies_and_bytes(toks, tmp_path):
    from volta_amd._lib import VoltaHipError
    from volta_amd.readers import WordPieceTokenizer
    mine, ref, vocab = toks
    with pytest.raises(VoltaHipError, match="cannot open"):
        WordPieceTokenizer(str(tmp_path / "missing.txt"))
    (tmp_path / "nounk.txt").write_text("a\nb\n")
    with pytest.raises(VoltaHipError, match="UNK"):
        WordPieceTokenizer(str(tmp_path / "nounk.txt"))
    (tmp_path / "crlf.txt").write_bytes(b"[PAD]\r\n[UNK]\r\n[CLS]\r\n[SEP]\r\n[MASK]\r\ndog\r\n##s")          # CRLF, no final newline
    t = WordPieceTokenizer(str(tmp_path / "crlf.txt"), threads=2)
    assert len(t) == 7 and t.encode("Dogs dog cat") == [5, 6, 5, 1]
    # bytes that are not UTF-8 are dropped like U+FFFD, they do not derail the rest of the text
    import ctypes as C
    from volta_amd import _lib as L
    raw = b"dog \xff\xfe cat\xc3"
    buf = (C.c_int32 * 16)()
    n = L.lib.vk_wordpiece_encode(mine._h, raw, len(raw), buf, 16)
    assert list(buf[:n]) == [vocab.index("dog"), vocab.index("cat")]
    assert L.lib.vk_wordpiece_encode(mine._h, raw, len(raw), buf, 1) == 2 and buf[0] == vocab.index("dog")     # the count is returned even when truncated
