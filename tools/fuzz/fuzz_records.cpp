#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include "volta_hip.h"
static thread_local char buf[512];
namespace vk { int set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vsnprintf(buf, sizeof buf, fmt, a); va_end(a); return -1; } }
extern "C" const char* vk_last_error(void) { return buf; }
static std::vector<unsigned char> slurp(const char* p) { FILE* f = fopen(p, "rb"); std::vector<unsigned char> v; if (!f) return v; fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); v.resize(n); if (n) fread(v.data(), 1, n, f); fclose(f); return v; }
int main(int argc, char** argv) {
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    std::vector<unsigned char> orig = slurp(argv[1]);
    int iters = atoi(argv[3]);
    long ok = 0, err = 0, recs = 0;
    for (int it = 0; it < iters; ++it) {
        std::vector<unsigned char> b = orig;
        int nm = 1 + rnd() % 6;
        for (int m = 0; m < nm; ++m) { size_t pg = rnd() % (b.size() / 4096); size_t off = pg * 4096 + ((rnd() % 10 < 7) ? rnd() % 64 : rnd() % 4096); b[off] = rnd() & 255; }
        if (rnd() % 10 == 0) b.resize(4096 * (2 + rnd() % (b.size() / 4096 - 1)));
        FILE* f = fopen("/tmp/asan/m.lmdb", "wb"); fwrite(b.data(), 1, b.size(), f); fclose(f);
        vk_lmdb* db = nullptr;
        if (vk_lmdb_open("/tmp/asan/m.lmdb", &db)) { ++err; continue; }
        if (vk_lmdb_first(db) == 0) {
            const void *k, *v; size_t kl, vl; int rc, cnt = 0; unsigned long sum = 0;
            while ((rc = vk_lmdb_next(db, &k, &kl, &v, &vl)) == 1 && cnt++ < 2000) { for (size_t i = 0; i < kl; ++i) sum += ((const unsigned char*)k)[i]; for (size_t i = 0; i < vl; i += 97) sum += ((const unsigned char*)v)[i]; if (vl) sum += ((const unsigned char*)v)[vl - 1]; ++recs; }
            if (rc < 0) ++err; else ++ok;
            char key[16];
            for (int q = 0; q < 20; ++q) { snprintf(key, sizeof key, "%08d", (int)(rnd() % 500)); if (vk_lmdb_get(db, key, 8, &v, &vl) == 1 && vl) sum += ((const unsigned char*)v)[vl - 1]; }
            if (sum == 42) printf("!");
        } else ++err;
        vk_lmdb_close(db);
    }
    printf("lmdb: ok %ld err %ld records read %ld\n", ok, err, recs);
    std::vector<unsigned char> rec = slurp(argv[2]);
    std::vector<float> feat(6 * 24), cls(6 * 11), attr(6 * 7), box(24), oc(6), ac(6);
    std::vector<int64_t> ol(6), al(6);
    ok = err = 0;
    for (int it = 0; it < iters * 20; ++it) {
        std::vector<unsigned char> b = rec;
        int nm = 1 + rnd() % 4;
        for (int m = 0; m < nm; ++m) { size_t off = (rnd() % 2) ? rnd() % b.size() : rnd() % 64; b[off] = rnd() & 255; }
        if (rnd() % 5 == 0) b.resize(rnd() % b.size());
        // exact-size heap copy so that ASan sees any over-read
        unsigned char* h = (unsigned char*)malloc(b.size() ? b.size() : 1); memcpy(h, b.data(), b.size());
        vk_concap_record r; memset(&r, 0, sizeof r);
        r.feat = feat.data(); r.cls = cls.data(); r.attr = attr.data(); r.boxes = box.data(); r.obj_labels = ol.data(); r.obj_confs = oc.data(); r.attr_labels = al.data(); r.attr_confs = ac.data();
        r.R = 6; r.F = 24; r.C = 11; r.A = 7;
        if (vk_concap_record_decode(h, b.size(), &r) == 0) { ++ok; volatile char c = 0; for (int i = 0; i < r.caption_len; ++i) c += r.caption[i]; } else ++err;
        free(h);
    }
    printf("record: ok %ld err %ld\n", ok, err);
    // base64
    for (int it = 0; it < iters * 20; ++it) {
        size_t n = rnd() % 200; std::vector<char> s(n ? n : 1); for (size_t i = 0; i < n; ++i) s[i] = "ABCDabcd0123+/=-_\n *"[rnd() % 20];
        size_t cap = rnd() % 160; std::vector<unsigned char> out(cap ? cap : 1); size_t ol2 = 0;
        char* h = (char*)malloc(n ? n : 1); memcpy(h, s.data(), n);
        unsigned char* o = (unsigned char*)malloc(cap ? cap : 1);
        vk_b64_decode(h, n, o, cap, &ol2);
        free(h); free(o);
    }
    printf("b64 done\n");
    // tokenizer: random bytes (valid and broken UTF-8, long runs without spaces), exact-size heap copies
    if (argc > 4) {
        vk_wordpiece* t = nullptr;
        if (vk_wordpiece_open(argv[4], 1, &t)) { printf("vocab: %s\n", vk_last_error()); return 1; }
        long total = 0;
        for (int it = 0; it < iters * 20; ++it) {
            size_t n = rnd() % 300;
            char* h = (char*)malloc(n ? n : 1);
            const int mode = rnd() % 4;
            for (size_t i = 0; i < n; ++i) h[i] = mode == 0 ? (char)(rnd() & 255) : mode == 1 ? "dog cat.,!  \t\xc3\xa9\xe4\xb8\xad\xf0\x9f\x98\x80" "ab"[rnd() % 24] : mode == 2 ? (char)('a' + rnd() % 3) : (char)(0x80 | (rnd() & 0x7f));
            int cap = rnd() % 64;
            int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * (cap ? cap : 1));
            int k = vk_wordpiece_encode(t, h, n, ids, cap);
            if (k < 0) { printf("encode failed\n"); return 1; }
            total += k;
            free(ids); free(h);
        }
        vk_wordpiece_close(t);
        printf("wordpiece: %ld ids\n", total);
    }
    return 0;
}
