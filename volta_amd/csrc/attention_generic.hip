// Gated bimodal attention for head sizes other than 64 (config/vilbert_base.json: 8 heads of 128 in the vision stream and in the
// co-attention sub-layers).  Same contract as attention.hip -- one workgroup per (batch element, head), every gate
// block of volta's BertGatedSelfAttention (encoders.py:258-340) in one launch, joint softmax of a query row over the key sets its
// modality attends, per-block dropout on the counter-based stream (element (query row, key) of block (mq, mk): word key & 3 of
// Philox-4x32-7(counter = (key >> 2, (b * nh + h) * Lq + q, site, 0))), log-sum-exp saved for the backward -- but written for generality, not
// for speed: plain fp32 FMAs on LDS-resident operands, one query row (forward, dQ) or one key row (dK, dV) per wave at a time, lanes over
// the other index.  The 64-wide heads of every ctrl_* config stay on the MFMA kernels of attention.hip while their rows fit its tiles
// (64 text tokens, 128 regions); LONGER rows -- the task configs' 80-token VCR captions, 200 regions of Visual7W / FlickrGrounding, 256 /
// 306 regions of GuessWhatPointing (config_tasks/all_tasks.yml:59-70,95-105,306-335; the reference's attention has no length limit,
// encoders.py:258-340) -- come here at any head size: a query row's keys are held NS x 64 at a time in registers (NS <= 8: 512 keys of
// both modalities together), and the backward runs in two phases that each keep only two row images in LDS (K, V for dQ; Q, dO for
// dK, dV), so that 64-wide heads fit up to ~540 rows in total in the forward and ~491 in the backward (vk_gated_attn_lds_bytes).
#include "common.h"
#include "../../include/volta_hip.h"
#include "util.h"

namespace vk {

struct AttnG {
    const uint16_t* q[2]; const uint16_t* k[2]; const uint16_t* v[2];
    int32_t ld[2], L[2];
    const float* mask[2];
    uint16_t* ctx[2]; int32_t ldo[2];
    float* lse[2];
    int32_t B, nh;
    int32_t gate[2][2];
    vk_dropout drop[2][2];
    float scale;
    const uint16_t* dctx[2];
    uint16_t* dq[2]; uint16_t* dk[2]; uint16_t* dv[2];
    int32_t ldg[2];
    float* probs[2][2];       // forward only: attention probabilities (after dropout) of block (mq, mk), or NULL
};

constexpr int GW = 4;                      // waves per workgroup
constexpr int GMAXNS = 8;                  // key groups of 64 per query row: at most 512 keys of both modalities together

template <int D> struct GLayout {
    static constexpr int RS = D * 2 + 16;  // LDS row stride in bytes: the 16-byte pad spreads lane-strided rows over the banks
};

// cooperative staging of rows [0, L) of one head (D bf16 each, 16-byte pieces) into an LDS image
template <int D>
__device__ __forceinline__ void g_stage(char* img, const uint16_t* base, int ld, int L, int tid, int nthr) {
    constexpr int CH = D / 8;
    for (int idx = tid; idx < L * CH; idx += nthr) {
        const int row = idx / CH, ch = idx - row * CH;
        *(u32x4*)(img + row * GLayout<D>::RS + ch * 16) = *(const u32x4*)(base + (size_t)row * ld + ch * 8);
    }
}
// dot of the wave's fp32 vector (LDS, broadcast reads) with one bf16 LDS row
template <int D>
__device__ __forceinline__ float g_dot(const float* vec, const char* row) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < D / 8; ++c) {
        const u32x4 w = *(const u32x4*)(row + c * 16);
        const f32x4 a = *(const f32x4*)(vec + c * 8), b = *(const f32x4*)(vec + c * 8 + 4);
        s += a[0] * bf2f(w[0] & 0xFFFF) + a[1] * bf2f(w[0] >> 16) + a[2] * bf2f(w[1] & 0xFFFF) + a[3] * bf2f(w[1] >> 16) +
             b[0] * bf2f(w[2] & 0xFFFF) + b[1] * bf2f(w[2] >> 16) + b[2] * bf2f(w[3] & 0xFFFF) + b[3] * bf2f(w[3] >> 16);
    }
    return s;
}
__device__ __forceinline__ float g_keep(const vk_dropout& dc, uint64_t seed, uint32_t drow, int key) {
    if (!dc.threshold) return 1.f;
    const u32x4 w = philox4((uint32_t)(key >> 2), drow, dc.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    return w[key & 3] >= dc.threshold ? dc.scale : 0.f;
}

// ------------------------------------------------------------------------------------------------ forward
template <int D, int NS>
__global__ __launch_bounds__(64 * GW) void attn_generic_fwd_kernel(const AttnG a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RS = GLayout<D>::RS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.nh, h = blockIdx.x - b * a.nh;
    char* kimg[2]; char* vimg[2];
    char* p = smem;
    for (int m = 0; m < 2; ++m) {
        const bool ka = a.gate[0][m] || a.gate[1][m];
        kimg[m] = p; p += ka ? a.L[m] * RS : 0;
        vimg[m] = p; p += ka ? a.L[m] * RS : 0;
        if (ka) {
            g_stage<D>(kimg[m], a.k[m] + (size_t)b * a.L[m] * a.ld[m] + h * D, a.ld[m], a.L[m], tid, 64 * GW);
            g_stage<D>(vimg[m], a.v[m] + (size_t)b * a.L[m] * a.ld[m] + h * D, a.ld[m], a.L[m], tid, 64 * GW);
        }
    }
    float* qbuf = (float*)p + wave * D;                 p += GW * D * 4;
    float* pbuf = (float*)p + wave * (NS * 64);
    __syncthreads();
    for (int mq = 0; mq < 2; ++mq) {
        if (!(a.gate[mq][0] || a.gate[mq][1])) continue;
        const int Lq = a.L[mq];
        const int n0 = a.gate[mq][0] ? a.L[0] : 0, n1 = a.gate[mq][1] ? a.L[1] : 0, ntot = n0 + n1;
        const uint64_t seed0 = a.drop[mq][0].threshold ? *a.drop[mq][0].seed : 0, seed1 = a.drop[mq][1].threshold ? *a.drop[mq][1].seed : 0;
        for (int q = wave; q < Lq; q += GW) {
            const uint16_t* qrow = a.q[mq] + ((size_t)b * Lq + q) * a.ld[mq] + h * D;
            if (2 * lane < D) {
                const uint32_t w = *(const uint32_t*)(qrow + 2 * lane);
                qbuf[2 * lane] = bf2f(w & 0xFFFF) * a.scale;
                qbuf[2 * lane + 1] = bf2f(w >> 16) * a.scale;
            }
            float s[NS], mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int j = lane + 64 * i;
                s[i] = -INFINITY;
                if (j < ntot) {
                    const int mk = j < n0 ? 0 : 1, k = j < n0 ? j : j - n0;
                    s[i] = g_dot<D>(qbuf, kimg[mk] + k * RS) + a.mask[mk][(size_t)b * a.L[mk] + k];
                }
                mx = fmaxf(mx, s[i]);
            }
            mx = wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) { s[i] = lane + 64 * i < ntot ? __expf(s[i] - mx) : 0.f; sum += s[i]; }
            sum = wave_sum(sum);
            const float inv = 1.0f / sum;
            if (lane == 0) a.lse[mq][((size_t)b * a.nh + h) * Lq + q] = mx + __logf(sum);
            const uint32_t drow = (uint32_t)(((size_t)b * a.nh + h) * Lq + q);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int j = lane + 64 * i;
                if (j < ntot) {
                    const int mk = j < n0 ? 0 : 1, k = j < n0 ? j : j - n0;
                    pbuf[j] = s[i] * inv * g_keep(a.drop[mq][mk], mk ? seed1 : seed0, drow, k);
                    if (a.probs[mq][mk]) a.probs[mq][mk][(((size_t)b * a.nh + h) * Lq + q) * a.L[mk] + k] = pbuf[j];
                }
            }
            float o0 = 0.f, o1 = 0.f;
            if (2 * lane < D) {
                for (int j = 0; j < ntot; ++j) {
                    const int mk = j < n0 ? 0 : 1, k = j < n0 ? j : j - n0;
                    const uint32_t w = *(const uint32_t*)(vimg[mk] + k * RS + 4 * lane);
                    const float pj = pbuf[j];
                    o0 += pj * bf2f(w & 0xFFFF);
                    o1 += pj * bf2f(w >> 16);
                }
                *(uint32_t*)(a.ctx[mq] + ((size_t)b * Lq + q) * a.ldo[mq] + h * D + 2 * lane) = pack2bf(o0, o1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// Two phases, each with two row images in LDS.  Phase 1 (K, V images): one query row per wave -- its q, dO and O rows come straight
// from global memory -- delta = dO . O, dS over the row's keys, dQ.  Phase 2 (Q, dO images, staged over the same LDS after a barrier;
// every query's log-sum-exp and delta = dO . O beside them): one key row per wave, dK and dV over the queries that attend it.
template <int D, int NS>
__global__ __launch_bounds__(64 * GW) void attn_generic_bwd_kernel(const AttnG a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RS = GLayout<D>::RS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / a.nh, h = blockIdx.x - b * a.nh;
    int Lsum = 0;
    for (int m = 0; m < 2; ++m) Lsum += a.L[m];
    // fixed carve: [image A | image B] rows of both modalities, then per-row floats, then the waves' vectors
    char* imgA[2]; char* imgB[2];
    float* lse_s[2]; float* del_s[2];
    char* p = smem;
    imgA[0] = p; imgA[1] = p + (size_t)a.L[0] * RS; p += (size_t)Lsum * RS;
    imgB[0] = p; imgB[1] = p + (size_t)a.L[0] * RS; p += (size_t)Lsum * RS;
    lse_s[0] = (float*)p; lse_s[1] = lse_s[0] + a.L[0]; p += ((size_t)Lsum * 4 + 15) & ~(size_t)15;
    del_s[0] = (float*)p; del_s[1] = del_s[0] + a.L[0]; p += ((size_t)Lsum * 4 + 15) & ~(size_t)15;
    float* vec0 = (float*)p + wave * D;                 p += GW * D * 4;       // the wave's fp32 row (q x scale, or k x scale)
    float* vec1 = (float*)p + wave * D;                 p += GW * D * 4;       // (dO, or v)
    float* pb0 = (float*)p + wave * (NS * 64);          p += GW * NS * 64 * 4; // dS
    float* pb1 = (float*)p + wave * (NS * 64);                                 // dropped probabilities
    uint64_t seed[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) seed[i][j] = a.drop[i][j].threshold ? *a.drop[i][j].seed : 0;

    // ---- phase 1: K, V images
    for (int m = 0; m < 2; ++m)
        if (a.gate[0][m] || a.gate[1][m]) {
            g_stage<D>(imgA[m], a.k[m] + (size_t)b * a.L[m] * a.ld[m] + h * D, a.ld[m], a.L[m], tid, 64 * GW);
            g_stage<D>(imgB[m], a.v[m] + (size_t)b * a.L[m] * a.ld[m] + h * D, a.ld[m], a.L[m], tid, 64 * GW);
        }
    __syncthreads();
    for (int mq = 0; mq < 2; ++mq) {
        if (!(a.gate[mq][0] || a.gate[mq][1])) continue;
        const int Lq = a.L[mq];
        const int n0 = a.gate[mq][0] ? a.L[0] : 0, n1 = a.gate[mq][1] ? a.L[1] : 0, ntot = n0 + n1;
        for (int q = wave; q < Lq; q += GW) {
            if (2 * lane < D) {
                const uint32_t wq = *(const uint32_t*)(a.q[mq] + ((size_t)b * Lq + q) * a.ld[mq] + h * D + 2 * lane);
                const uint32_t wg = *(const uint32_t*)(a.dctx[mq] + ((size_t)b * Lq + q) * a.ldo[mq] + h * D + 2 * lane);
                vec0[2 * lane] = bf2f(wq & 0xFFFF) * a.scale; vec0[2 * lane + 1] = bf2f(wq >> 16) * a.scale;
                vec1[2 * lane] = bf2f(wg & 0xFFFF);           vec1[2 * lane + 1] = bf2f(wg >> 16);
            }
            const uint32_t drow = (uint32_t)(((size_t)b * a.nh + h) * Lq + q);
            const float lse = a.lse[mq][((size_t)b * a.nh + h) * Lq + q];
            float pr[NS], dp[NS], kp[NS], delta = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int j = lane + 64 * i;
                pr[i] = dp[i] = kp[i] = 0.f;
                if (j < ntot) {
                    const int mk = j < n0 ? 0 : 1, k = j < n0 ? j : j - n0;
                    const float sc = g_dot<D>(vec0, imgA[mk] + k * RS) + a.mask[mk][(size_t)b * a.L[mk] + k];
                    pr[i] = __expf(sc - lse);
                    kp[i] = g_keep(a.drop[mq][mk], seed[mq][mk], drow, k);
                    dp[i] = g_dot<D>(vec1, imgB[mk] + k * RS);
                    delta += pr[i] * kp[i] * dp[i];
                }
            }
            delta = wave_sum(delta);
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int j = lane + 64 * i;
                if (j < ntot) pb0[j] = pr[i] * (kp[i] * dp[i] - delta) * a.scale;
            }
            if (2 * lane < D) {
                float o0 = 0.f, o1 = 0.f;
                for (int j = 0; j < ntot; ++j) {
                    const int mk = j < n0 ? 0 : 1, k = j < n0 ? j : j - n0;
                    const uint32_t w = *(const uint32_t*)(imgA[mk] + k * RS + 4 * lane);
                    const float ds = pb0[j];
                    o0 += ds * bf2f(w & 0xFFFF);
                    o1 += ds * bf2f(w >> 16);
                }
                *(uint32_t*)(a.dq[mq] + ((size_t)b * Lq + q) * a.ldg[mq] + h * D + 2 * lane) = pack2bf(o0, o1);
            }
        }
    }
    __syncthreads();
    // ---- phase 2: Q, dO images over the same LDS; log-sum-exp and delta = dO . O (= sum_j P keep dP, the value phase 1 used) per query
    for (int m = 0; m < 2; ++m)
        if (a.gate[m][0] || a.gate[m][1]) {
            g_stage<D>(imgA[m], a.q[m] + (size_t)b * a.L[m] * a.ld[m] + h * D, a.ld[m], a.L[m], tid, 64 * GW);
            g_stage<D>(imgB[m], a.dctx[m] + (size_t)b * a.L[m] * a.ldo[m] + h * D, a.ldo[m], a.L[m], tid, 64 * GW);
            for (int i = tid; i < a.L[m]; i += 64 * GW) lse_s[m][i] = a.lse[m][((size_t)b * a.nh + h) * a.L[m] + i];
            for (int q = wave; q < a.L[m]; q += GW) {
                float d = 0.f;
                if (2 * lane < D) {
                    const uint32_t wo = *(const uint32_t*)(a.ctx[m] + ((size_t)b * a.L[m] + q) * a.ldo[m] + h * D + 2 * lane);
                    const uint32_t wg = *(const uint32_t*)(a.dctx[m] + ((size_t)b * a.L[m] + q) * a.ldo[m] + h * D + 2 * lane);
                    d = bf2f(wo & 0xFFFF) * bf2f(wg & 0xFFFF) + bf2f(wo >> 16) * bf2f(wg >> 16);
                }
                d = wave_sum(d);
                if (lane == 0) del_s[m][q] = d;
            }
        }
    __syncthreads();
    for (int mk = 0; mk < 2; ++mk) {
        if (!(a.gate[0][mk] || a.gate[1][mk])) continue;
        const int Lk = a.L[mk];
        const int n0 = a.gate[0][mk] ? a.L[0] : 0, n1 = a.gate[1][mk] ? a.L[1] : 0, ntot = n0 + n1;
        for (int k = wave; k < Lk; k += GW) {
            if (2 * lane < D) {
                const uint32_t wk = *(const uint32_t*)(a.k[mk] + ((size_t)b * Lk + k) * a.ld[mk] + h * D + 2 * lane);
                const uint32_t wv = *(const uint32_t*)(a.v[mk] + ((size_t)b * Lk + k) * a.ld[mk] + h * D + 2 * lane);
                vec0[2 * lane] = bf2f(wk & 0xFFFF) * a.scale; vec0[2 * lane + 1] = bf2f(wk >> 16) * a.scale;
                vec1[2 * lane] = bf2f(wv & 0xFFFF);           vec1[2 * lane + 1] = bf2f(wv >> 16);
            }
            const float maskv = a.mask[mk][(size_t)b * Lk + k];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int j = lane + 64 * i;
                if (j < ntot) {
                    const int mq = j < n0 ? 0 : 1, q = j < n0 ? j : j - n0;
                    const float sc = g_dot<D>(vec0, imgA[mq] + q * RS) + maskv;
                    const float pr = __expf(sc - lse_s[mq][q]);
                    const float kp = g_keep(a.drop[mq][mk], seed[mq][mk], (uint32_t)(((size_t)b * a.nh + h) * a.L[mq] + q), k);
                    const float dp = g_dot<D>(vec1, imgB[mq] + q * RS);
                    pb0[j] = pr * (kp * dp - del_s[mq][q]) * a.scale;
                    pb1[j] = pr * kp;
                }
            }
            if (2 * lane < D) {
                float k0 = 0.f, k1 = 0.f, v0 = 0.f, v1 = 0.f;
                for (int j = 0; j < ntot; ++j) {
                    const int mq = j < n0 ? 0 : 1, q = j < n0 ? j : j - n0;
                    const uint32_t wq = *(const uint32_t*)(imgA[mq] + q * RS + 4 * lane), wg = *(const uint32_t*)(imgB[mq] + q * RS + 4 * lane);
                    const float ds = pb0[j], pd = pb1[j];
                    k0 += ds * bf2f(wq & 0xFFFF); k1 += ds * bf2f(wq >> 16);
                    v0 += pd * bf2f(wg & 0xFFFF); v1 += pd * bf2f(wg >> 16);
                }
                *(uint32_t*)(a.dk[mk] + ((size_t)b * Lk + k) * a.ldg[mk] + h * D + 2 * lane) = pack2bf(k0, k1);
                *(uint32_t*)(a.dv[mk] + ((size_t)b * Lk + k) * a.ldg[mk] + h * D + 2 * lane) = pack2bf(v0, v1);
            }
        }
    }
}

static size_t g_lds_bytes(const AttnG& k, int D, int NS, bool bwd) {
    const size_t RS = (size_t)D * 2 + 16;
    size_t n = 0;
    if (bwd) {
        const size_t Lsum = (size_t)k.L[0] + (size_t)k.L[1];
        n = 2 * Lsum * RS + 2 * ((Lsum * 4 + 15) & ~(size_t)15);
    } else {
        for (int m = 0; m < 2; ++m)
            if (k.gate[0][m] || k.gate[1][m]) n += 2 * (size_t)k.L[m] * RS;
    }
    n += (size_t)GW * D * 4 * (bwd ? 2 : 1) + (size_t)GW * NS * 64 * 4 * (bwd ? 2 : 1);
    return (n + 15) & ~(size_t)15;
}

template <int D, int NS>
static int g_launch_ns(const AttnG& k, bool bwd, hipStream_t s) {
    const size_t lds = g_lds_bytes(k, D, NS, bwd);
    if (lds > 160 * 1024) return set_error("vk_gated_attn: head size %d with lengths (%d, %d) needs %zu bytes of LDS (> 160 KiB)", D, k.L[0], k.L[1], lds);
    if (bwd) {
        auto kern = attn_generic_bwd_kernel<D, NS>;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return set_error("vk_gated_attn_bwd: cannot reserve %zu bytes of LDS", lds);
        hipLaunchKernelGGL(kern, dim3(k.B * k.nh), dim3(64 * GW), lds, s, k);
        return check_launch("vk_gated_attn_bwd (generic kernel)");
    }
    auto kern = attn_generic_fwd_kernel<D, NS>;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return set_error("vk_gated_attn_fwd: cannot reserve %zu bytes of LDS", lds);
    hipLaunchKernelGGL(kern, dim3(k.B * k.nh), dim3(64 * GW), lds, s, k);
    return check_launch("vk_gated_attn_fwd (generic kernel)");
}

template <int D>
static int g_launch(const AttnG& k, bool bwd, hipStream_t s) {
    // keys a query row can see (both modalities when both gates of a query modality are open); the backward's key rows see queries alike
    int need = 0;
    for (int m = 0; m < 2; ++m) {
        const int nq = (k.gate[m][0] ? k.L[0] : 0) + (k.gate[m][1] ? k.L[1] : 0), nk = (k.gate[0][m] ? k.L[0] : 0) + (k.gate[1][m] ? k.L[1] : 0);
        need = need > nq ? need : nq;
        need = need > nk ? need : nk;
    }
    if (need > GMAXNS * 64) return set_error("vk_gated_attn: %d keys per query row (at most %d on the generic kernels)", need, GMAXNS * 64);
    if (need <= 192) return g_launch_ns<D, 3>(k, bwd, s);
    if (need <= 320) return g_launch_ns<D, 5>(k, bwd, s);
    return g_launch_ns<D, 8>(k, bwd, s);
}

// LDS bytes the generic kernels would ask for (host arithmetic for vk_gated_attn_lds_bytes); 0 when no key is visible
size_t attn_generic_lds(const vk_attn_args* a, bool bwd) {
    AttnG k;
    for (int m = 0; m < 2; ++m) {
        const bool qa = a->gate[m][0] || a->gate[m][1], ka = a->gate[0][m] || a->gate[1][m];
        k.L[m] = (qa || ka) ? a->L[m] : 0;
        for (int j = 0; j < 2; ++j) k.gate[m][j] = a->gate[m][j];
    }
    int need = 0;
    for (int m = 0; m < 2; ++m) {
        const int nq = (k.gate[m][0] ? k.L[0] : 0) + (k.gate[m][1] ? k.L[1] : 0), nk = (k.gate[0][m] ? k.L[0] : 0) + (k.gate[1][m] ? k.L[1] : 0);
        need = need > nq ? need : nq;
        need = need > nk ? need : nk;
    }
    const int NS = need <= 192 ? 3 : need <= 320 ? 5 : 8;
    return g_lds_bytes(k, a->dh ? a->dh : 64, NS, bwd);
}

// called by vk_gated_attn_fwd / _bwd (attention.hip) for what its MFMA kernels do not hold: head sizes other than 64 / 128, rows beyond their tiles
int attn_generic(const vk_attn_args* a, const vk_attn_bwd_args* bw, vk_stream_t stream) {
    AttnG k;
    for (int m = 0; m < 2; ++m) {
        k.q[m] = (const uint16_t*)a->q[m]; k.k[m] = (const uint16_t*)a->k[m]; k.v[m] = (const uint16_t*)a->v[m];
        k.ld[m] = a->ld[m]; k.L[m] = a->L[m]; k.mask[m] = a->mask[m];
        k.ctx[m] = (uint16_t*)a->ctx[m]; k.ldo[m] = a->ldo[m]; k.lse[m] = a->lse[m];
        for (int j = 0; j < 2; ++j) { k.gate[m][j] = a->gate[m][j]; k.drop[m][j] = a->drop[m][j]; }
        k.dctx[m] = bw ? (const uint16_t*)bw->dctx[m] : nullptr;
        k.dq[m] = bw ? (uint16_t*)bw->dq[m] : nullptr; k.dk[m] = bw ? (uint16_t*)bw->dk[m] : nullptr;
        k.dv[m] = bw ? (uint16_t*)bw->dv[m] : nullptr; k.ldg[m] = bw ? bw->ldg[m] : 0;
        for (int j = 0; j < 2; ++j) k.probs[m][j] = (!bw && a->gate[m][j]) ? a->probs[m][j] : nullptr;
    }
    k.B = a->B; k.nh = a->nh; k.scale = a->scale;
    for (int m = 0; m < 2; ++m) {
        const bool qa = a->gate[m][0] || a->gate[m][1], ka = a->gate[0][m] || a->gate[1][m];
        if (!qa && !ka) { k.L[m] = 0; continue; }
        if (a->L[m] <= 0) return set_error("vk_gated_attn: modality %d is gated on but has length %d", m, a->L[m]);
        if ((a->ld[m] & 7) || (a->ldo[m] & 7) || (bw && (bw->ldg[m] & 7))) return set_error("vk_gated_attn: row strides must be multiples of 8");
        if (ka && (!a->k[m] || !a->v[m] || !a->mask[m])) return set_error("vk_gated_attn: K/V/mask of modality %d missing", m);
        if (qa && (!a->q[m] || !a->ctx[m] || !a->lse[m])) return set_error("vk_gated_attn: Q/ctx/lse of modality %d missing", m);
        if (bw && qa && (!bw->dctx[m] || !bw->dq[m])) return set_error("vk_gated_attn_bwd: dctx/dq of modality %d missing", m);
        if (bw && ka && (!bw->dk[m] || !bw->dv[m])) return set_error("vk_gated_attn_bwd: dk/dv of modality %d missing", m);
    }
    if (a->B <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    switch (a->dh ? a->dh : 64) {
        case 32: return g_launch<32>(k, bw != nullptr, s);
        case 64: return g_launch<64>(k, bw != nullptr, s);
        case 96: return g_launch<96>(k, bw != nullptr, s);
        case 128: return g_launch<128>(k, bw != nullptr, s);
        default: return set_error("vk_gated_attn: head size %d (32, 64, 96 or 128)", a->dh);
    }
}

}  // namespace vk
