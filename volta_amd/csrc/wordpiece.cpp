// BERT tokenisation of captions in native code (host side of the record loader): what `self.tokenizer.encode(caption)` does in
// BertPreprocessBatch.__call__ (volta/datasets/concept_cap_dataset.py:465) with the reference's tokenizer, pytorch-transformers 1.1's
// BertTokenizer (`requirements.txt:39`; not under /root/reference -- restated from its published algorithm, Devlin et al.'s tokenization.py):
//   basic tokenizer  drop NUL / U+FFFD / control characters, every whitespace character -> ' ', spaces around CJK ideographs, split on
//                    whitespace, then per word: lower-case, NFD, drop combining marks (Mn), split at every punctuation character;
//   WordPiece        greedy longest-match-first against the vocabulary, continuation pieces prefixed "##", words of more than 100 characters
//                    or without a full cover -> [UNK].
// `encode` returns ids without [CLS] / [SEP], as that tokenizer's `encode(text)` did.  Text that spells a special token ("[MASK]") is
// tokenised like any other text (the reference keeps such words whole: captions do not contain them).
// Unicode data: unicode_tables.h, generated from Python's unicodedata (tools/gen_unicode_tables.py).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "unicode_tables.h"
#include "util.h"
#include "volta_hip.h"

using vk::set_error;

namespace {

using namespace vk_unicode;

bool in_ranges(const Range* r, int n, uint32_t c) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (c < r[mid].lo) hi = mid - 1;
        else if (c > r[mid].hi) lo = mid + 1;
        else return true;
    }
    return false;
}

inline bool is_whitespace(uint32_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || (c >= 0x80 && in_ranges(WS, N_WS, c)); }
inline bool is_control(uint32_t c) {
    if (c == '\t' || c == '\n' || c == '\r') return false;
    if (c < 0x80) return c < 0x20 || c == 0x7F;
    return in_ranges(CTRL, N_CTRL, c);
}
inline bool is_punct(uint32_t c) {
    if (c < 0x80) return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
    return in_ranges(PUNCT, N_PUNCT, c);
}
inline bool is_cjk(uint32_t c) {
    return (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0x3400 && c <= 0x4DBF) || (c >= 0x20000 && c <= 0x2A6DF) || (c >= 0x2A700 && c <= 0x2B73F) ||
           (c >= 0x2B740 && c <= 0x2B81F) || (c >= 0x2B820 && c <= 0x2CEAF) || (c >= 0xF900 && c <= 0xFAFF) || (c >= 0x2F800 && c <= 0x2FA1F);
}

// UTF-8 -> code points; malformed bytes become U+FFFD (which the cleaning step then drops, as Python's decoder + BasicTokenizer would)
void decode_utf8(const char* s, size_t n, std::vector<uint32_t>& out) {
    const unsigned char* p = (const unsigned char*)s;
    size_t i = 0;
    while (i < n) {
        const unsigned char b = p[i];
        uint32_t c = 0xFFFD;
        int len = 1;
        if (b < 0x80) c = b;
        else if ((b & 0xE0) == 0xC0 && i + 1 < n && (p[i + 1] & 0xC0) == 0x80) { c = ((b & 0x1Fu) << 6) | (p[i + 1] & 0x3Fu); len = 2; if (c < 0x80) c = 0xFFFD; }
        else if ((b & 0xF0) == 0xE0 && i + 2 < n && (p[i + 1] & 0xC0) == 0x80 && (p[i + 2] & 0xC0) == 0x80) {
            c = ((b & 0x0Fu) << 12) | ((p[i + 1] & 0x3Fu) << 6) | (p[i + 2] & 0x3Fu); len = 3;
            if (c < 0x800 || (c >= 0xD800 && c <= 0xDFFF)) c = 0xFFFD;
        } else if ((b & 0xF8) == 0xF0 && i + 3 < n && (p[i + 1] & 0xC0) == 0x80 && (p[i + 2] & 0xC0) == 0x80 && (p[i + 3] & 0xC0) == 0x80) {
            c = ((b & 0x07u) << 18) | ((p[i + 1] & 0x3Fu) << 12) | ((p[i + 2] & 0x3Fu) << 6) | (p[i + 3] & 0x3Fu); len = 4;
            if (c < 0x10000 || c > 0x10FFFF) c = 0xFFFD;
        }
        out.push_back(c);
        i += len;
    }
}

void append_utf8(std::string& s, uint32_t c) {
    if (c < 0x80) s.push_back((char)c);
    else if (c < 0x800) { s.push_back((char)(0xC0 | (c >> 6))); s.push_back((char)(0x80 | (c & 0x3F))); }
    else if (c < 0x10000) { s.push_back((char)(0xE0 | (c >> 12))); s.push_back((char)(0x80 | ((c >> 6) & 0x3F))); s.push_back((char)(0x80 | (c & 0x3F))); }
    else { s.push_back((char)(0xF0 | (c >> 18))); s.push_back((char)(0x80 | ((c >> 12) & 0x3F))); s.push_back((char)(0x80 | ((c >> 6) & 0x3F))); s.push_back((char)(0x80 | (c & 0x3F))); }
}

// lower-case + NFD + drop Mn of one code point, appended to `out`
void fold(uint32_t c, std::vector<uint32_t>& out) {
    if (c < 0x80) { out.push_back(c >= 'A' && c <= 'Z' ? c + 32 : c); return; }
    if (c >= 0xAC00 && c <= 0xD7A3) {               // Hangul syllable: algorithmic canonical decomposition (no case, no Mn)
        const uint32_t s = c - 0xAC00, l = 0x1100 + s / 588, v = 0x1161 + (s % 588) / 28, t = 0x11A7 + s % 28;
        out.push_back(l);
        out.push_back(v);
        if (t != 0x11A7) out.push_back(t);
        return;
    }
    int lo = 0, hi = N_FOLD - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        if (c < FOLD[mid].cp) hi = mid - 1;
        else if (c > FOLD[mid].cp) lo = mid + 1;
        else {
            for (uint32_t k = 0; k < FOLD[mid].len; ++k) out.push_back(FOLD_POOL[FOLD[mid].off + k]);
            return;
        }
    }
    out.push_back(c);
}

// str.lower()'s one context rule (CPython handle_capital_sigma; the reference lower-cases each whitespace-separated token with it): U+03A3 at
// position k of the token [b, e) takes the final form U+03C2 when the nearest character before it that is not case-ignorable is cased and the
// nearest one after it that is not case-ignorable is not (or there is none).
bool final_sigma(const std::vector<uint32_t>& cp, size_t b, size_t e, size_t k) {
    size_t j = k;
    while (j > b && in_ranges(CASE_IGNORABLE, N_CASE_IGNORABLE, cp[j - 1])) --j;
    if (j == b || !in_ranges(CASED, N_CASED, cp[j - 1])) return false;
    j = k + 1;
    while (j < e && in_ranges(CASE_IGNORABLE, N_CASE_IGNORABLE, cp[j])) ++j;
    return j == e || !in_ranges(CASED, N_CASED, cp[j]);
}

}  // namespace

struct vk_wordpiece {
    std::unordered_map<std::string, int32_t> vocab;
    int32_t unk = -1;
    bool lowercase = true;
    size_t max_piece_bytes = 0;

    // one word (already folded, no whitespace / punctuation inside unless it IS a single punctuation character) -> ids
    void word(const std::vector<uint32_t>& w, size_t b, size_t e, std::vector<int32_t>& ids, std::string& scratch) const {
        if (e - b > 100) { ids.push_back(unk); return; }
        // byte offsets of the characters
        std::string text;
        std::vector<uint32_t> at;
        for (size_t i = b; i < e; ++i) { at.push_back((uint32_t)text.size()); append_utf8(text, w[i]); }
        at.push_back((uint32_t)text.size());
        const size_t n = e - b, first = ids.size();
        size_t start = 0;
        while (start < n) {
            size_t end = n;
            int32_t found = -1;
            while (start < end) {
                scratch.assign(start ? "##" : "");
                scratch.append(text, at[start], at[end] - at[start]);
                if (scratch.size() <= max_piece_bytes) {
                    auto it = vocab.find(scratch);
                    if (it != vocab.end()) { found = it->second; break; }
                }
                --end;
            }
            if (found < 0) { ids.resize(first); ids.push_back(unk); return; }
            ids.push_back(found);
            start = end;
        }
    }

    void encode(const char* s, size_t len, std::vector<int32_t>& ids) const {
        std::vector<uint32_t> raw, cp, w;
        decode_utf8(s, len, raw);
        cp.reserve(raw.size() + 8);
        for (uint32_t c : raw) {                                   // clean + CJK spacing
            if (c == 0 || c == 0xFFFD || is_control(c)) continue;
            if (is_whitespace(c)) { cp.push_back(' '); continue; }
            if (is_cjk(c)) { cp.push_back(' '); cp.push_back(c); cp.push_back(' '); continue; }
            cp.push_back(c);
        }
        std::string scratch;
        size_t i = 0;
        while (i < cp.size()) {
            while (i < cp.size() && cp[i] == ' ') ++i;
            size_t j = i;
            while (j < cp.size() && cp[j] != ' ') ++j;
            if (j == i) break;
            w.clear();
            if (lowercase) for (size_t k = i; k < j; ++k) {
                if (cp[k] == 0x3A3 && final_sigma(cp, i, j, k)) w.push_back(0x3C2);
                else fold(cp[k], w);
            }
            else w.assign(cp.begin() + i, cp.begin() + j);
            // split at punctuation: every punctuation character is a word of its own
            size_t b = 0;
            for (size_t k = 0; k <= w.size(); ++k) {
                const bool p = k < w.size() && is_punct(w[k]);
                if (k == w.size() || p) {
                    if (k > b) word(w, b, k, ids, scratch);
                    if (p) word(w, k, k + 1, ids, scratch);
                    b = k + 1;
                }
            }
            i = j;
        }
    }
};

extern "C" int vk_wordpiece_open(const char* vocab_path, int lowercase, vk_wordpiece** out) {
    if (!vocab_path || !out) return set_error("vk_wordpiece_open: null argument");
    FILE* f = fopen(vocab_path, "rb");
    if (!f) return set_error("vk_wordpiece_open: cannot open %s", vocab_path);
    vk_wordpiece* t = new vk_wordpiece;
    t->lowercase = lowercase != 0;
    std::string line;
    int c;
    int32_t id = 0;
    auto flush = [&]() {
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
        t->vocab[line] = id;                 // a repeated token keeps its LAST line number, as load_vocab's dict assignment does
        t->max_piece_bytes = std::max(t->max_piece_bytes, line.size());
        ++id;
        line.clear();
    };
    bool any = false;
    while ((c = fgetc(f)) != EOF) {
        any = true;
        if (c == '\n') flush(); else line.push_back((char)c);
    }
    if (any && !line.empty()) flush();
    fclose(f);
    auto it = t->vocab.find("[UNK]");
    if (it == t->vocab.end()) {
        delete t;
        return set_error("vk_wordpiece_open: %s has no [UNK] entry", vocab_path);
    }
    t->unk = it->second;
    *out = t;
    return 0;
}

extern "C" void vk_wordpiece_close(vk_wordpiece* t) { delete t; }

extern "C" int vk_wordpiece_vocab_size(const vk_wordpiece* t) { return t ? (int)t->vocab.size() : -1; }

extern "C" int vk_wordpiece_token_id(const vk_wordpiece* t, const char* token) {
    if (!t || !token) return -1;
    auto it = t->vocab.find(token);
    return it == t->vocab.end() ? -1 : it->second;
}

// ids of `text` (UTF-8, `len` bytes) into ids[0 .. cap); returns the number of ids the text has (ids beyond `cap` are not written)
extern "C" int vk_wordpiece_encode(const vk_wordpiece* t, const char* text, size_t len, int32_t* ids, int cap) {
    if (!t || (!text && len) || (!ids && cap > 0)) return set_error("vk_wordpiece_encode: null argument");
    std::vector<int32_t> v;
    t->encode(text, len, v);
    const int n = (int)v.size();
    if (n && cap > 0) memcpy(ids, v.data(), sizeof(int32_t) * (size_t)std::min(n, cap));
    return n;
}

// n texts into rows of ids [n, ld] (zero-padded, truncated at ld), counts[i] = min(number of ids, ld); `threads` host threads
extern "C" int vk_wordpiece_encode_batch(const vk_wordpiece* t, const char* const* texts, const size_t* lens, int n, int32_t* ids, int ld, int32_t* counts, int threads) {
    if (n <= 0) return 0;
    if (!t || !texts || !lens || !ids || !counts || ld <= 0) return set_error("vk_wordpiece_encode_batch: null argument");
    threads = std::max(1, std::min(threads, n));
    std::atomic<int> next{0};
    auto work = [&]() {
        std::vector<int32_t> v;
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            v.clear();
            t->encode(texts[i], lens[i], v);
            const int k = std::min((int)v.size(), ld);
            int32_t* row = ids + (size_t)i * ld;
            if (k) memcpy(row, v.data(), sizeof(int32_t) * (size_t)k);
            if (k < ld) memset(row + k, 0, sizeof(int32_t) * (size_t)(ld - k));
            counts[i] = k;
        }
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < threads; ++i) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return 0;
}
