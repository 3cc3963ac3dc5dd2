// Record readers in front of the ConceptCap batch producer.  Host code: the files are memory-mapped and every field is
// decoded straight into the caller's (pinned) staging slot, one copy per byte.
//
//  * vk_lmdb_*          read-only walk of an LMDB data file (the container of the reference's feature stores: tensorpack's LMDBSerializer
//                       for Conceptual Captions, volta/datasets/concept_cap_dataset.py:117-121,305-309; `lmdb.open(..., readonly=True)` in
//                       volta/datasets/_image_features_reader.py:46-56).  Follows LMDB 0.9's on-disk layout (data version 1): two meta pages,
//                       a B+tree of branch / leaf pages, values larger than a page on overflow pages.  Main database only, no DUPSORT.
//  * vk_concap_record_decode   one Conceptual Captions datapoint: a msgpack array of 13 fields with msgpack_numpy-encoded ndarrays, the
//                       order BertPreprocessBatch.__call__ unpacks (concept_cap_dataset.py:430-431).
//  * vk_b64_decode      the base64 fields of the extraction TSV (data/conceptual_captions/preprocess_cc_train.py:66-68) and of the
//                       pickled per-image dicts of the task feature stores (_image_features_reader.py:87-88).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "util.h"
#include "volta_hip.h"

using vk::set_error;

// ------------------------------------------------------------------------------------------------ LMDB
namespace {

constexpr uint32_t MDB_MAGIC = 0xBEEFC0DEu;
constexpr size_t PAGE_HDR = 16;                    // pgno (8) | pad (2) | flags (2) | lower (2), upper (2)  -- or a 4-byte page count
constexpr uint16_t P_BRANCH = 0x01, P_LEAF = 0x02, P_OVERFLOW = 0x04, P_META = 0x08, P_LEAF2 = 0x20;
constexpr uint16_t F_BIGDATA = 0x01, F_SUBDATA = 0x02, F_DUPDATA = 0x04;
constexpr uint64_t P_INVALID = ~(uint64_t)0;
constexpr int MAX_DEPTH = 32;

template <typename T>
inline T rd(const uint8_t* p) {
    T v;
    memcpy(&v, p, sizeof(T));
    return v;
}

struct Meta {
    uint32_t psize = 0;
    uint16_t depth = 0, flags = 0;
    uint64_t entries = 0, root = P_INVALID, last_pg = 0, txnid = 0;
    bool ok = false;
};

}  // namespace

struct vk_lmdb {
    const uint8_t* base = nullptr;
    size_t size = 0;
    int fd = -1;
    Meta meta;
    // cursor: path from the root to the current leaf
    uint64_t pg[MAX_DEPTH];
    int idx[MAX_DEPTH];
    int top = -1;

    const uint8_t* page(uint64_t pgno) const {
        if (pgno > meta.last_pg || (pgno + 1) * (uint64_t)meta.psize > size) return nullptr;
        return base + pgno * meta.psize;
    }
};

namespace {

Meta read_meta(const uint8_t* p, size_t avail) {
    Meta m;
    if (avail < PAGE_HDR + 136) return m;
    if (!(rd<uint16_t>(p + 10) & P_META)) return m;
    const uint8_t* q = p + PAGE_HDR;
    if (rd<uint32_t>(q) != MDB_MAGIC) return m;
    const uint32_t version = rd<uint32_t>(q + 4);
    if (version != 1) return m;
    m.psize = rd<uint32_t>(q + 24);                 // mm_dbs[FREE_DBI].md_pad holds the page size
    const uint8_t* db = q + 72;                     // mm_dbs[MAIN_DBI]
    m.flags = rd<uint16_t>(db + 4);
    m.depth = rd<uint16_t>(db + 6);
    m.entries = rd<uint64_t>(db + 32);
    m.root = rd<uint64_t>(db + 40);
    m.last_pg = rd<uint64_t>(q + 120);
    m.txnid = rd<uint64_t>(q + 128);
    m.ok = m.psize >= 512 && (m.psize & (m.psize - 1)) == 0;
    return m;
}

// entries of a branch / leaf page; 0 for a page whose `lower` bound is not a whole offset table inside the page (a damaged file must not
// send the walker outside the mapping)
inline int num_keys(const vk_lmdb* db, const uint8_t* pg) {
    const uint32_t lower = rd<uint16_t>(pg + 12), upper = rd<uint16_t>(pg + 14);
    if (lower < PAGE_HDR || (lower & 1) || lower > upper || upper > db->meta.psize) return 0;
    return (int)((lower - PAGE_HDR) >> 1);
}

// node i of a page: lo (2) | hi (2) | flags (2) | ksize (2) | key | data
inline const uint8_t* node_at(const vk_lmdb* db, const uint8_t* pg, int i) {
    const int n = num_keys(db, pg);
    if (i < 0 || i >= n) return nullptr;
    const uint16_t off = rd<uint16_t>(pg + PAGE_HDR + 2 * (size_t)i);
    if (off < PAGE_HDR + 2 * (size_t)n || (size_t)off + 8 > db->meta.psize) return nullptr;
    return pg + off;
}

inline uint64_t branch_child(const uint8_t* n) { return (uint64_t)rd<uint16_t>(n) | ((uint64_t)rd<uint16_t>(n + 2) << 16) | ((uint64_t)rd<uint16_t>(n + 4) << 32); }

int leaf_value(const vk_lmdb* db, const uint8_t* pg, const uint8_t* n, const void** key, size_t* klen, const void** val, size_t* vlen) {
    const uint16_t flags = rd<uint16_t>(n + 4), ks = rd<uint16_t>(n + 6);
    const size_t dsize = (size_t)rd<uint16_t>(n) | ((size_t)rd<uint16_t>(n + 2) << 16);
    if (flags & (F_SUBDATA | F_DUPDATA)) return set_error("vk_lmdb: named sub-databases / DUPSORT records are not supported");
    const size_t in_page = (size_t)(n - pg) + 8 + ks;
    if (in_page > db->meta.psize) return set_error("vk_lmdb: a key runs past its page");
    if (key) *key = n + 8;
    if (klen) *klen = ks;
    if (flags & F_BIGDATA) {
        if (in_page + 8 > db->meta.psize) return set_error("vk_lmdb: truncated overflow reference");
        const uint64_t opg = rd<uint64_t>(n + 8 + ks);
        const uint8_t* o = db->page(opg);
        if (!o || !(rd<uint16_t>(o + 10) & P_OVERFLOW)) return set_error("vk_lmdb: bad overflow page %llu", (unsigned long long)opg);
        const uint64_t npages = rd<uint32_t>(o + 12);
        if (PAGE_HDR + dsize > npages * db->meta.psize || (opg + npages) * db->meta.psize > db->size) return set_error("vk_lmdb: overflow value runs past its pages");
        *val = o + PAGE_HDR;
    } else {
        if (in_page + dsize > db->meta.psize) return set_error("vk_lmdb: a value runs past its page");
        *val = n + 8 + ks;
    }
    *vlen = dsize;
    return 0;
}

inline int key_cmp(const void* a, size_t al, const void* b, size_t bl) {
    const int c = memcmp(a, b, al < bl ? al : bl);
    return c ? c : (al < bl ? -1 : (al > bl ? 1 : 0));
}

// descend to the left-most leaf below the page on top of the cursor stack
int descend_left(vk_lmdb* db) {
    for (;;) {
        const uint8_t* pg = db->page(db->pg[db->top]);
        if (!pg) return set_error("vk_lmdb: page %llu outside the file", (unsigned long long)db->pg[db->top]);
        const uint16_t fl = rd<uint16_t>(pg + 10);
        if (fl & P_LEAF) return (fl & P_LEAF2) ? set_error("vk_lmdb: DUPFIXED leaf pages are not supported") : 0;
        if (!(fl & P_BRANCH) || num_keys(db, pg) < 1) return set_error("vk_lmdb: page %llu is neither branch nor leaf", (unsigned long long)db->pg[db->top]);
        if (db->top + 1 >= MAX_DEPTH) return set_error("vk_lmdb: tree deeper than %d", MAX_DEPTH);
        const uint8_t* n = node_at(db, pg, db->idx[db->top]);
        if (!n) return set_error("vk_lmdb: bad node offset");
        db->pg[db->top + 1] = branch_child(n);
        db->idx[db->top + 1] = 0;
        ++db->top;
    }
}

}  // namespace

extern "C" int vk_lmdb_open(const char* path, vk_lmdb** out) {
    if (!path || !out) return set_error("vk_lmdb_open: null argument");
    std::string p(path);
    struct stat st;
    if (stat(p.c_str(), &st) != 0) return set_error("vk_lmdb_open: cannot stat %s", path);
    if (S_ISDIR(st.st_mode)) {
        p += "/data.mdb";
        if (stat(p.c_str(), &st) != 0) return set_error("vk_lmdb_open: no data.mdb under %s", path);
    }
    const int fd = open(p.c_str(), O_RDONLY);
    if (fd < 0) return set_error("vk_lmdb_open: cannot open %s", p.c_str());
    if ((size_t)st.st_size < 2 * 512) {
        close(fd);
        return set_error("vk_lmdb_open: %s is too short to hold the meta pages", p.c_str());
    }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) {
        close(fd);
        return set_error("vk_lmdb_open: mmap of %s failed", p.c_str());
    }
    vk_lmdb* db = new vk_lmdb;
    db->base = (const uint8_t*)m;
    db->size = (size_t)st.st_size;
    db->fd = fd;
    const Meta m0 = read_meta(db->base, db->size);
    Meta m1;
    if (m0.ok && (size_t)m0.psize + PAGE_HDR + 136 <= db->size) m1 = read_meta(db->base + m0.psize, db->size - m0.psize);
    if (!m0.ok) {
        vk_lmdb_close(db);
        return set_error("vk_lmdb_open: %s has no LMDB meta page (magic / data version 1)", p.c_str());
    }
    db->meta = (m1.ok && m1.txnid > m0.txnid) ? m1 : m0;
    db->meta.psize = m0.psize;
    if (db->meta.root != P_INVALID && !db->page(db->meta.root)) {
        vk_lmdb_close(db);
        return set_error("vk_lmdb_open: root page outside the file (truncated copy?)");
    }
    *out = db;
    return 0;
}

extern "C" void vk_lmdb_close(vk_lmdb* db) {
    if (!db) return;
    if (db->base) munmap((void*)db->base, db->size);
    if (db->fd >= 0) close(db->fd);
    delete db;
}

extern "C" int64_t vk_lmdb_entries(const vk_lmdb* db) { return db ? (int64_t)db->meta.entries : -1; }

extern "C" int vk_lmdb_first(vk_lmdb* db) {
    if (!db) return set_error("vk_lmdb_first: null handle");
    db->top = -1;
    if (db->meta.root == P_INVALID) return 0;
    db->top = 0;
    db->pg[0] = db->meta.root;
    db->idx[0] = 0;
    return descend_left(db);
}

// 1 = a record was returned and the cursor advanced, 0 = end of the database, -1 = error
extern "C" int vk_lmdb_next(vk_lmdb* db, const void** key, size_t* klen, const void** val, size_t* vlen) {
    if (!db || !val || !vlen) return set_error("vk_lmdb_next: null argument");
    while (db->top >= 0) {
        const uint8_t* pg = db->page(db->pg[db->top]);
        if (!pg) return set_error("vk_lmdb_next: page outside the file");
        if (db->idx[db->top] < num_keys(db, pg)) {
            if (rd<uint16_t>(pg + 10) & P_LEAF) {
                const uint8_t* n = node_at(db, pg, db->idx[db->top]);
                if (!n) return set_error("vk_lmdb_next: bad node offset");
                if (leaf_value(db, pg, n, key, klen, val, vlen)) return -1;
                ++db->idx[db->top];
                return 1;
            }
            if (descend_left(db)) return -1;      // a branch entry not yet visited
            continue;
        }
        --db->top;                                // page exhausted: next entry of the parent
        if (db->top >= 0) ++db->idx[db->top];
    }
    return 0;
}

// 1 = found, 0 = no such key, -1 = error.  Does not move the cursor.
extern "C" int vk_lmdb_get(const vk_lmdb* db, const void* key, size_t klen, const void** val, size_t* vlen) {
    if (!db || !key || !val || !vlen) return set_error("vk_lmdb_get: null argument");
    if (db->meta.root == P_INVALID) return 0;
    uint64_t pgno = db->meta.root;
    for (int level = 0; level < MAX_DEPTH; ++level) {
        const uint8_t* pg = db->page(pgno);
        if (!pg) return set_error("vk_lmdb_get: page %llu outside the file", (unsigned long long)pgno);
        const uint16_t fl = rd<uint16_t>(pg + 10);
        const int n = num_keys(db, pg);
        if (fl & P_LEAF) {
            if (fl & P_LEAF2) return set_error("vk_lmdb_get: DUPFIXED leaf pages are not supported");
            int lo = 0, hi = n - 1;
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1;
                const uint8_t* nd = node_at(db, pg, mid);
                if (!nd) return set_error("vk_lmdb_get: bad node offset");
                const int c = key_cmp(key, klen, nd + 8, rd<uint16_t>(nd + 6));
                if (c == 0) return leaf_value(db, pg, nd, nullptr, nullptr, val, vlen) ? -1 : 1;
                if (c < 0) hi = mid - 1; else lo = mid + 1;
            }
            return 0;
        }
        if (!(fl & P_BRANCH) || n < 1) return set_error("vk_lmdb_get: page %llu is neither branch nor leaf", (unsigned long long)pgno);
        // the first branch key is implicit (-inf): the child is the last entry whose key <= the search key
        int lo = 1, hi = n - 1, pick = 0;
        while (lo <= hi) {
            const int mid = (lo + hi) >> 1;
            const uint8_t* nd = node_at(db, pg, mid);
            if (!nd) return set_error("vk_lmdb_get: bad node offset");
            if (key_cmp(key, klen, nd + 8, rd<uint16_t>(nd + 6)) >= 0) { pick = mid; lo = mid + 1; } else hi = mid - 1;
        }
        const uint8_t* nd = node_at(db, pg, pick);
        if (!nd) return set_error("vk_lmdb_get: bad node offset");
        pgno = branch_child(nd);
    }
    return set_error("vk_lmdb_get: tree deeper than %d", MAX_DEPTH);
}

// ------------------------------------------------------------------------------------------------ base64
extern "C" int vk_b64_decode(const char* src, size_t n, void* dst, size_t cap, size_t* out_len) {
    static int8_t table[256];
    static bool init = false;
    if (!init) {
        memset(table, -1, sizeof(table));
        const char* abc = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
        for (int i = 0; i < 64; ++i) table[(uint8_t)abc[i]] = (int8_t)i;
        table[(uint8_t)'-'] = 62;                   // url-safe alphabet accepted as well
        table[(uint8_t)'_'] = 63;
        init = true;
    }
    if (!src || !dst || !out_len) return set_error("vk_b64_decode: null argument");
    uint8_t* d = (uint8_t*)dst;
    size_t o = 0;
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t c = (uint8_t)src[i];
        if (c == '=') break;
        const int8_t v = table[c];
        if (v < 0) {
            if (c == '\n' || c == '\r' || c == ' ') continue;
            return set_error("vk_b64_decode: byte 0x%02x at %zu is not base64", c, i);
        }
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            if (o >= cap) return set_error("vk_b64_decode: output needs more than %zu bytes", cap);
            d[o++] = (uint8_t)(acc >> bits);
        }
    }
    *out_len = o;
    return 0;
}

// ------------------------------------------------------------------------------------------------ msgpack (the subset a datapoint uses)
namespace {

struct Cur {
    const uint8_t* p;
    const uint8_t* end;
    bool bad = false;
    bool need(size_t n) {
        if ((size_t)(end - p) < n) bad = true;
        return !bad;
    }
    template <typename T>
    T be() {                                        // big-endian scalar
        if (!need(sizeof(T))) return T(0);
        uint64_t v = 0;
        for (size_t i = 0; i < sizeof(T); ++i) v = (v << 8) | p[i];
        p += sizeof(T);
        return (T)v;
    }
};

enum Kind { K_NIL, K_BOOL, K_INT, K_FLOAT, K_STR, K_BIN, K_ARRAY, K_MAP, K_EXT };
struct Val {
    Kind kind = K_NIL;
    int64_t i = 0;
    double f = 0;
    const uint8_t* s = nullptr;
    size_t n = 0;                                   // bytes of str / bin, members of array / map
};

// reads the head of the next value; strings / blobs are consumed, containers are entered (the caller reads `n` members / pairs)
bool next(Cur& c, Val& v) {
    if (!c.need(1)) return false;
    const uint8_t t = *c.p++;
    v = Val();
    auto blob = [&](Kind k, size_t n) {
        v.kind = k;
        v.n = n;
        if (!c.need(n)) return false;
        v.s = c.p;
        c.p += n;
        return true;
    };
    if (t <= 0x7f) { v.kind = K_INT; v.i = t; return true; }
    if (t >= 0xe0) { v.kind = K_INT; v.i = (int8_t)t; return true; }
    if (t >= 0x80 && t <= 0x8f) { v.kind = K_MAP; v.n = t & 15; return true; }
    if (t >= 0x90 && t <= 0x9f) { v.kind = K_ARRAY; v.n = t & 15; return true; }
    if (t >= 0xa0 && t <= 0xbf) return blob(K_STR, t & 31);
    switch (t) {
        case 0xc0: v.kind = K_NIL; return true;
        case 0xc2: case 0xc3: v.kind = K_BOOL; v.i = t & 1; return true;
        case 0xc4: return blob(K_BIN, c.be<uint8_t>());
        case 0xc5: return blob(K_BIN, c.be<uint16_t>());
        case 0xc6: return blob(K_BIN, c.be<uint32_t>());
        case 0xc7: { const size_t n = c.be<uint8_t>(); return blob(K_EXT, n + 1); }
        case 0xc8: { const size_t n = c.be<uint16_t>(); return blob(K_EXT, n + 1); }
        case 0xc9: { const size_t n = c.be<uint32_t>(); return blob(K_EXT, n + 1); }
        case 0xca: { const uint32_t u = c.be<uint32_t>(); float f; memcpy(&f, &u, 4); v.kind = K_FLOAT; v.f = f; return !c.bad; }
        case 0xcb: { const uint64_t u = c.be<uint64_t>(); memcpy(&v.f, &u, 8); v.kind = K_FLOAT; return !c.bad; }
        case 0xcc: v.kind = K_INT; v.i = c.be<uint8_t>(); return !c.bad;
        case 0xcd: v.kind = K_INT; v.i = c.be<uint16_t>(); return !c.bad;
        case 0xce: v.kind = K_INT; v.i = c.be<uint32_t>(); return !c.bad;
        case 0xcf: v.kind = K_INT; v.i = (int64_t)c.be<uint64_t>(); return !c.bad;
        case 0xd0: v.kind = K_INT; v.i = (int8_t)c.be<uint8_t>(); return !c.bad;
        case 0xd1: v.kind = K_INT; v.i = (int16_t)c.be<uint16_t>(); return !c.bad;
        case 0xd2: v.kind = K_INT; v.i = (int32_t)c.be<uint32_t>(); return !c.bad;
        case 0xd3: v.kind = K_INT; v.i = (int64_t)c.be<uint64_t>(); return !c.bad;
        case 0xd4: return blob(K_EXT, 2);
        case 0xd5: return blob(K_EXT, 3);
        case 0xd6: return blob(K_EXT, 5);
        case 0xd7: return blob(K_EXT, 9);
        case 0xd8: return blob(K_EXT, 17);
        case 0xd9: return blob(K_STR, c.be<uint8_t>());
        case 0xda: return blob(K_STR, c.be<uint16_t>());
        case 0xdb: return blob(K_STR, c.be<uint32_t>());
        case 0xdc: v.kind = K_ARRAY; v.n = c.be<uint16_t>(); return !c.bad;
        case 0xdd: v.kind = K_ARRAY; v.n = c.be<uint32_t>(); return !c.bad;
        case 0xde: v.kind = K_MAP; v.n = c.be<uint16_t>(); return !c.bad;
        case 0xdf: v.kind = K_MAP; v.n = c.be<uint32_t>(); return !c.bad;
    }
    c.bad = true;                                   // 0xc1: never used
    return false;
}

bool skip(Cur& c, const Val& v, int depth = 0) {
    if (depth > 16) return !(c.bad = true);
    size_t members = v.kind == K_ARRAY ? v.n : (v.kind == K_MAP ? 2 * v.n : 0);
    for (size_t i = 0; i < members; ++i) {
        Val m;
        if (!next(c, m) || !skip(c, m, depth + 1)) return false;
    }
    return true;
}

// one field of a datapoint: an ndarray ({nd: True, type, kind, shape, data}), a numpy scalar ({nd: False, type, data}) or a plain scalar / string
struct Field {
    bool is_array = false;
    char dtype[8] = {0};
    int ndim = 0;
    int64_t shape[4] = {0, 0, 0, 0};
    const uint8_t* data = nullptr;
    size_t nbytes = 0;
    Val scalar;                                     // when not an array
    int64_t count() const {
        int64_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= shape[i];
        return n;
    }
};

bool key_is(const Val& k, const char* name) { return (k.kind == K_STR || k.kind == K_BIN) && k.n == strlen(name) && memcmp(k.s, name, k.n) == 0; }

bool read_field(Cur& c, Field& f) {
    Val v;
    if (!next(c, v)) return false;
    if (v.kind != K_MAP) {
        f.scalar = v;
        return skip(c, v);
    }
    bool nd = false, have_nd = false;
    for (size_t i = 0; i < v.n; ++i) {
        Val k, x;
        if (!next(c, k) || !next(c, x)) return false;
        if (key_is(k, "nd")) { have_nd = true; nd = x.kind == K_BOOL && x.i; }
        else if (key_is(k, "type") && (x.kind == K_STR || x.kind == K_BIN) && x.n < sizeof(f.dtype)) memcpy(f.dtype, x.s, x.n);
        else if (key_is(k, "data") && (x.kind == K_BIN || x.kind == K_STR)) { f.data = x.s; f.nbytes = x.n; }
        else if (key_is(k, "shape") && x.kind == K_ARRAY) {
            if (x.n > 4) return !(c.bad = true);
            f.ndim = (int)x.n;
            for (size_t d = 0; d < x.n; ++d) {
                Val s;
                if (!next(c, s) || s.kind != K_INT || s.i < 0) return !(c.bad = true);
                f.shape[d] = s.i;
            }
            continue;
        }
        if (!skip(c, x)) return false;
    }
    if (!have_nd || !f.data) return !(c.bad = true);
    f.is_array = true;
    if (!nd) f.ndim = 0;                            // numpy scalar: one element
    return true;
}

int elem_size(const char* dt) {
    if (dt[0] != '<' && dt[0] != '|' && dt[0] != '=') return 0;
    const int n = atoi(dt + 2);
    return (dt[1] == 'f' && (n == 4 || n == 8)) || ((dt[1] == 'i' || dt[1] == 'u') && (n == 1 || n == 2 || n == 4 || n == 8)) ? n : 0;
}

double elem_at(const Field& f, int64_t i) {
    const uint8_t* p = f.data + (size_t)i * elem_size(f.dtype);
    const char k = f.dtype[1];
    switch (elem_size(f.dtype)) {
        case 1: return k == 'i' ? (double)rd<int8_t>(p) : (double)rd<uint8_t>(p);
        case 2: return k == 'i' ? (double)rd<int16_t>(p) : (double)rd<uint16_t>(p);
        case 4: return k == 'f' ? (double)rd<float>(p) : (k == 'i' ? (double)rd<int32_t>(p) : (double)rd<uint32_t>(p));
        default: return k == 'f' ? rd<double>(p) : (k == 'i' ? (double)rd<int64_t>(p) : (double)rd<uint64_t>(p));
    }
}

// a field that holds one number: python int / float, a decimal string (the TSV hands num_boxes / img_h / img_w over as text), a numpy scalar
bool field_number(const Field& f, double* out) {
    if (f.is_array) {
        if (!elem_size(f.dtype) || f.count() != 1 || f.nbytes < (size_t)elem_size(f.dtype)) return false;
        *out = elem_at(f, 0);
        return true;
    }
    const Val& v = f.scalar;
    if (v.kind == K_INT || v.kind == K_BOOL) { *out = (double)v.i; return true; }
    if (v.kind == K_FLOAT) { *out = v.f; return true; }
    if ((v.kind == K_STR || v.kind == K_BIN) && v.n && v.n < 40) {
        char buf[40];
        memcpy(buf, v.s, v.n);
        buf[v.n] = 0;
        char* e = nullptr;
        *out = strtod(buf, &e);
        return e && *e == 0;
    }
    return false;
}

// rows [0, nb) of an [nb, width] (or [nb]) array into dst [R, width] as T; rows >= nb are zero, as in the reference's np.zeros staging
template <typename T>
int fill_rows(const Field& f, const char* name, int64_t nb, int64_t width, int64_t R, T* dst) {
    if (!dst) return 0;
    const int es = elem_size(f.dtype);
    if (!f.is_array || !es) return set_error("vk_concap_record_decode: `%s` is not a numeric ndarray (dtype '%s')", name, f.dtype);
    const int64_t rows = f.ndim >= 1 ? f.shape[0] : 0, w = f.ndim == 2 ? f.shape[1] : (f.ndim == 1 ? 1 : -1);
    if (rows < nb || w != width || (size_t)(rows * w) * es > f.nbytes)
        return set_error("vk_concap_record_decode: `%s` has shape [%lld, %lld], need at least [%lld, %lld]", name, (long long)rows, (long long)w, (long long)nb, (long long)width);
    const bool same = (sizeof(T) == 4 && f.dtype[1] == 'f' && es == 4 && std::is_floating_point<T>::value) || (sizeof(T) == 8 && f.dtype[1] == 'i' && es == 8 && !std::is_floating_point<T>::value);
    if (same) memcpy(dst, f.data, (size_t)(nb * width) * sizeof(T));
    else
        for (int64_t i = 0; i < nb * width; ++i) dst[i] = (T)elem_at(f, i);
    memset(dst + nb * width, 0, (size_t)((R - nb) * width) * sizeof(T));
    return 0;
}

}  // namespace

extern "C" int vk_concap_record_decode(const void* rec, size_t len, vk_concap_record* r) {
    if (!rec || !r || r->R <= 0) return set_error("vk_concap_record_decode: null argument");
    Cur c{(const uint8_t*)rec, (const uint8_t*)rec + len};
    Val top;
    if (!next(c, top) || top.kind != K_ARRAY || top.n != 13)
        return set_error("vk_concap_record_decode: a datapoint is a msgpack array of 13 fields (concept_cap_dataset.py:430-431)");
    Field f[13];
    for (int i = 0; i < 13; ++i)
        if (!read_field(c, f[i]) || c.bad) return set_error("vk_concap_record_decode: malformed msgpack in field %d", i);
    // order: features, cls_prob, obj_labels, obj_confs, attr_labels, attr_confs, attr_scores, boxes, num_boxes, img_h, img_w, img_id, caption
    double nbd, h, w;
    if (!field_number(f[8], &nbd) || !field_number(f[9], &h) || !field_number(f[10], &w)) return set_error("vk_concap_record_decode: num_boxes / img_h / img_w are not numbers");
    const int64_t nb = (int64_t)nbd;
    if (nb < 0 || nb > r->R) return set_error("vk_concap_record_decode: %lld boxes do not fit region_len %d", (long long)nb, r->R);
    if (fill_rows(f[0], "features", nb, r->F, r->R, r->feat) || fill_rows(f[1], "cls_prob", nb, r->C, r->R, r->cls) ||
        fill_rows(f[7], "boxes", nb, 4, r->R, r->boxes) || fill_rows(f[6], "attr_scores", nb, r->A, r->R, r->attr) ||
        fill_rows(f[2], "obj_labels", nb, 1, r->R, r->obj_labels) || fill_rows(f[3], "obj_confs", nb, 1, r->R, r->obj_confs) ||
        fill_rows(f[4], "attr_labels", nb, 1, r->R, r->attr_labels) || fill_rows(f[5], "attr_confs", nb, 1, r->R, r->attr_confs))
        return -1;
    r->num_boxes = (int32_t)nb;
    r->img_h = (float)h;
    r->img_w = (float)w;
    // image id: text or a number
    r->image_id[0] = 0;
    double idn;
    const Val& iv = f[11].scalar;
    if (!f[11].is_array && (iv.kind == K_STR || iv.kind == K_BIN)) {
        const size_t n = iv.n < sizeof(r->image_id) - 1 ? iv.n : sizeof(r->image_id) - 1;
        memcpy(r->image_id, iv.s, n);
        r->image_id[n] = 0;
    } else if (field_number(f[11], &idn)) {
        snprintf(r->image_id, sizeof(r->image_id), "%lld", (long long)idn);
    }
    const Val& cv = f[12].scalar;
    if (f[12].is_array || (cv.kind != K_STR && cv.kind != K_BIN)) return set_error("vk_concap_record_decode: the caption is not a string");
    r->caption = (const char*)cv.s;
    r->caption_len = (int32_t)cv.n;
    return 0;
}

// A batch of datapoints on `threads` host threads (the decoder shares nothing: every record has its own slot).  Returns 0 or -1 with the
// message of the first record that failed; `failed` (optional) receives that record's index.
extern "C" int vk_concap_records_decode(const void* const* recs, const size_t* lens, vk_concap_record* slots, int n, int threads, int* failed) {
    if (n <= 0) return 0;
    if (!recs || !lens || !slots) return set_error("vk_concap_records_decode: null argument");
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    if (failed) *failed = -1;
    if (threads == 1) {
        for (int i = 0; i < n; ++i)
            if (vk_concap_record_decode(recs[i], lens[i], &slots[i])) {
                if (failed) *failed = i;
                return -1;
            }
        return 0;
    }
    std::atomic<int> next{0}, bad{n};
    std::vector<std::string> msg((size_t)threads);
    auto work = [&](int t) {
        for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
            if (vk_concap_record_decode(recs[i], lens[i], &slots[i]) == 0) continue;
            int cur = bad.load();
            while (i < cur && !bad.compare_exchange_weak(cur, i)) {}
            if (i <= bad.load()) msg[(size_t)t] = vk_last_error();      // the error text is thread-local: carry it to the caller's thread
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    if (bad.load() == n) return 0;
    if (failed) *failed = bad.load();
    // re-run the failing record on this thread: deterministic, and leaves its message in this thread's vk_last_error()
    return vk_concap_record_decode(recs[bad.load()], lens[bad.load()], &slots[bad.load()]) ? -1 : set_error("vk_concap_records_decode: record %d failed", bad.load());
}
