// Gated bimodal attention for gfx950, forward and backward: ONE launch per attention sub-layer covers
// every present block of volta's BertGatedSelfAttention (tt, tv, vt, vv; volta/encoders.py:228-358):
//   scores = Q K^T / sqrt(dh) + mask ; joint softmax of a query row over [text keys | vision keys] that
//   its modality attends (encoders.py:285-314) ; per-block dropout ; context = P V summed over blocks.
// One workgroup per (batch element, head).  K/V (and Q/dO in backward) of both modalities are staged
// once into LDS in a "dual-use" image that serves ds_read_b128 row fragments and ds_read_b64_tr_b16
// transposed fragments (cdna guide T10).  Every 16-query tile (forward, dQ) / 16-key tile (dK, dV) is
// one wave's task.  MFMAs are issued in the swapped orientation (D = K Q^T) so that a lane owns one
// query and the accumulator tile is directly the next MFMA's B operand (no LDS round trip for P).
// Scores never touch HBM; the backward recomputes P from the saved log-sum-exp.
// Head size 64 (every ctrl_* config) or 128 (config/vilbert_base.json), a template parameter; sequence lengths up to 64 text / 128 vision
// tokens (128-wide heads: up to 96 padded rows, beyond that and for other head sizes attention_generic.hip).
#include "common.h"
#include "../../include/volta_hip.h"
#include "util.h"

namespace vk {



// dual-use LDS image of a [rows][DHT] bf16 tile: rows of 2 DHT bytes (128 for the 64-wide heads of every ctrl_* config, 256 for the 128-wide
// heads of config/vilbert_base.json), 32-B blocks XOR-swizzled by (row>>1)&3
template <int DHT>
__device__ __forceinline__ uint32_t img_off(int row, int col) {
    return (uint32_t)(row * (2 * DHT) + ((((col >> 4) ^ ((row >> 1) & 3)) << 5) | ((col & 15) << 1)));
}
// 8 consecutive dims [c8, c8+8) of one row: MFMA fragment for rows = (lane&15), k = dims
template <int DHT>
__device__ __forceinline__ bf16x8 img_row_frag(uint32_t img, int row0, int ks, int lane) {
    const int row = row0 + (lane & 15);
    return *(const bf16x8 VK_LDS*)(uintptr_t)(img + img_off<DHT>(row, ks * 32 + (lane >> 4) * 8));
}
// transposed fragment: output rows = the 16 dims [d0, d0+16), k = 32 tile rows in "pair order":
// slot j<4 of lane group g is row rbase + 4g + j, slot j>=4 is row rbase + 16 + 4g + (j-4).
template <int DHT>
__device__ __forceinline__ bf16x8 img_tr_frag(uint32_t img, int rbase, int d0, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int r0 = rbase + 4 * g + q, r1 = r0 + 16;
    bf16x4 lo = lds_read_tr16(img + img_off<DHT>(r0, d0 + 4 * p));
    bf16x4 hi = lds_read_tr16(img + img_off<DHT>(r1, d0 + 4 * p));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// Cooperative staging of rows [0, L) x 64 dims of one head into an image of RPAD rows (zero padded), split into a
// LOAD half and a STORE half: a kernel first issues the global loads of ALL its images (K, V, Q, dO, O of both
// modalities) and only then writes LDS, so the workgroup pays one memory latency instead of one per image
// (staging image after image cost 8-12 dependent round trips, most of a backward workgroup's lifetime).
// Needs >= 256 threads (pieces per thread are sized for 256).
template <int RPAD, int DHT> struct StagePieces { static constexpr int CH = DHT / 8, N = (RPAD * CH + 255) / 256; };      // CH 16-byte pieces per row
template <int RPAD, int DHT>
__device__ __forceinline__ void stage_load(u32x4 (&r)[(StagePieces<RPAD, DHT>::N)], const uint16_t* base, int ld, int L, int tid, int nthr) {
    constexpr int CH = DHT / 8;
#pragma unroll
    for (int i = 0; i < StagePieces<RPAD, DHT>::N; ++i) {
        const int idx = i * nthr + tid, row = idx / CH, ch = idx % CH;
        r[i] = u32x4{0u, 0u, 0u, 0u};
        if (idx < RPAD * CH && row < L) r[i] = *(const u32x4*)(base + (size_t)row * ld + ch * 8);
    }
}
template <int RPAD, int DHT>
__device__ __forceinline__ void stage_store(uint32_t img, const u32x4 (&r)[(StagePieces<RPAD, DHT>::N)], int tid, int nthr) {
    constexpr int CH = DHT / 8;
#pragma unroll
    for (int i = 0; i < StagePieces<RPAD, DHT>::N; ++i) {
        const int idx = i * nthr + tid, row = idx / CH, ch = idx % CH;
        if (idx < RPAD * CH) *(u32x4 VK_LDS*)(uintptr_t)(img + img_off<DHT>(row, ch * 8)) = r[i];
    }
}
// delta[row] = sum_d dO[row][d] * O[row][d] from the staged registers (DHT / 8 lanes per row), plus the lse copy
template <int RPAD, int DHT>
__device__ __forceinline__ void delta_rows(const u32x4 (&rg)[(StagePieces<RPAD, DHT>::N)], const u32x4 (&ro)[(StagePieces<RPAD, DHT>::N)], float VK_LDS* del_s,
                                           float VK_LDS* lse_s, const float* lse, int L, int tid, int nthr) {
    constexpr int CH = DHT / 8;
#pragma unroll
    for (int i = 0; i < StagePieces<RPAD, DHT>::N; ++i) {
        const int idx = i * nthr + tid, row = idx / CH, ch = idx % CH;
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            part += bf2f(rg[i][k] & 0xFFFF) * bf2f(ro[i][k] & 0xFFFF) + bf2f(rg[i][k] >> 16) * bf2f(ro[i][k] >> 16);
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if (CH > 8) part += __shfl_xor(part, 8, 64);
        if (ch == 0 && idx < RPAD * CH) {
            del_s[row] = part;
            lse_s[row] = row < L ? lse[row] : 0.f;
        }
    }
}

__device__ __forceinline__ float group_max(float v) {   // over the 4 lanes sharing lane&15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ bf16x8 pack_pair(const f32x4& a, const f32x4& b) {      // 4 x v_cvt_pk_bf16_f32
    const u32x4 w = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, w);
}

// word `sel` (0..3) of the u32x4 held by lane R of this lane's quad (4 consecutive lanes): DPP quad broadcasts + selects
template <int R>
__device__ __forceinline__ uint32_t quad_word(const u32x4& w, int sel) {
    constexpr int ctrl = R * 0x55;       // quad_perm [R, R, R, R]
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[0], ctrl, 0xF, 0xF, false);
    const uint32_t b1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[1], ctrl, 0xF, 0xF, false);
    const uint32_t b2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[2], ctrl, 0xF, 0xF, false);
    const uint32_t b3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[3], ctrl, 0xF, 0xF, false);
    return sel == 0 ? b0 : sel == 1 ? b1 : sel == 2 ? b2 : b3;
}

struct AttnK {                       // kernel-side copy of vk_attn_args (+ backward pointers)
    const uint16_t* q[2]; const uint16_t* k[2]; const uint16_t* v[2];
    int32_t ld[2], L[2];
    const float* mask[2];
    uint16_t* ctx[2]; int32_t ldo[2];
    float* lse[2];
    int32_t B, nh;
    int32_t gate[2][2];
    vk_dropout drop[2][2];
    float scale;
    // backward only
    const uint16_t* dctx[2];
    uint16_t* dq[2]; uint16_t* dk[2]; uint16_t* dv[2];
    int32_t ldg[2];
};

template <int P0, int P1> struct Pads { static constexpr int P[2] = {P0, P1}; };

// ------------------------------------------------------------------------------------------------ forward
template <int TP, int RP, int OCC, int DHT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void attn_fwd_kernel(const AttnK a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(VK_LDS char*)smem;
    constexpr int NKT[2] = {TP / 16, RP / 16};
    constexpr int RB = 2 * DHT, KS = DHT / 32, DT = DHT / 16;      // image row bytes, 32-deep contraction steps, 16-wide output tiles per head
    const uint32_t kimg[2] = {lds0, lds0 + TP * RB};
    const uint32_t vimg[2] = {lds0 + (TP + RP) * RB, lds0 + (TP + RP) * RB + TP * RB};
    const uint32_t qimg[2] = {lds0 + 2 * (TP + RP) * RB, lds0 + 2 * (TP + RP) * RB + TP * RB};
    // additive key masks of this batch element as floats behind the images: [TP | RP]
    float VK_LDS* const mask_s = (float VK_LDS*)(uintptr_t)(lds0 + 3 * (TP + RP) * RB);
    constexpr int MB[2] = {0, TP};
    const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
    const int b = blockIdx.x / a.nh, h = blockIdx.x - b * a.nh;
    const int g = lane >> 4, lq = lane & 15;

    // Everything a task reads is staged up front -- K, V, Q rows and the key masks -- so that the task loop makes no global load: with Q rows
    // and masks fetched inside the loop every task began with two more dependent memory round trips.
    {
        const int nthr = blockDim.x;
        const bool k0 = a.gate[0][0] || a.gate[1][0], k1 = a.gate[0][1] || a.gate[1][1];
        const bool q0 = a.gate[0][0] || a.gate[0][1], q1 = a.gate[1][0] || a.gate[1][1];
        u32x4 rk0[StagePieces<TP, DHT>::N], rv0[StagePieces<TP, DHT>::N], rk1[StagePieces<RP, DHT>::N], rv1[StagePieces<RP, DHT>::N];
        u32x4 rq0[StagePieces<TP, DHT>::N], rq1[StagePieces<RP, DHT>::N];
        float mk = 0.f;
        if (k0) {
            stage_load<TP, DHT>(rk0, a.k[0] + ((size_t)b * a.L[0]) * a.ld[0] + h * DHT, a.ld[0], a.L[0], tid, nthr);
            stage_load<TP, DHT>(rv0, a.v[0] + ((size_t)b * a.L[0]) * a.ld[0] + h * DHT, a.ld[0], a.L[0], tid, nthr);
        }
        if (k1) {
            stage_load<RP, DHT>(rk1, a.k[1] + ((size_t)b * a.L[1]) * a.ld[1] + h * DHT, a.ld[1], a.L[1], tid, nthr);
            stage_load<RP, DHT>(rv1, a.v[1] + ((size_t)b * a.L[1]) * a.ld[1] + h * DHT, a.ld[1], a.L[1], tid, nthr);
        }
        if (q0) stage_load<TP, DHT>(rq0, a.q[0] + ((size_t)b * a.L[0]) * a.ld[0] + h * DHT, a.ld[0], a.L[0], tid, nthr);
        if (q1) stage_load<RP, DHT>(rq1, a.q[1] + ((size_t)b * a.L[1]) * a.ld[1] + h * DHT, a.ld[1], a.L[1], tid, nthr);
        if (tid < TP) { if (k0 && tid < a.L[0]) mk = a.mask[0][(size_t)b * a.L[0] + tid]; }
        else if (tid < TP + RP) { if (k1 && tid - TP < a.L[1]) mk = a.mask[1][(size_t)b * a.L[1] + (tid - TP)]; }
        if (k0) { stage_store<TP, DHT>(kimg[0], rk0, tid, nthr); stage_store<TP, DHT>(vimg[0], rv0, tid, nthr); }
        if (k1) { stage_store<RP, DHT>(kimg[1], rk1, tid, nthr); stage_store<RP, DHT>(vimg[1], rv1, tid, nthr); }
        if (q0) stage_store<TP, DHT>(qimg[0], rq0, tid, nthr);
        if (q1) stage_store<RP, DHT>(qimg[1], rq1, tid, nthr);
        if (tid < TP + RP) mask_s[tid] = mk;
        if (tid == 0) *(int VK_LDS*)(uintptr_t)(lds0 + 3 * (TP + RP) * RB + (TP + RP) * 4) = 0;
    }
    __syncthreads();

    const int nqt0 = (a.gate[0][0] || a.gate[0][1]) ? (a.L[0] + 15) / 16 : 0;
    const int nqt1 = (a.gate[1][0] || a.gate[1][1]) ? (a.L[1] + 15) / 16 : 0;
    // Tasks (16-query tiles) are claimed from a counter in LDS, vision tiles (more keys per tile) first: 5 tiles on 4 waves in fixed
    // round-robin order left one wave with text + vision tile while the others idled.
    int VK_LDS* const next_task = (int VK_LDS*)(uintptr_t)(lds0 + 3 * (TP + RP) * RB + (TP + RP) * 4);
    (void)nwaves;
    for (;;) {
        int task = 0;
        if (lane == 0) task = __hip_atomic_fetch_add(next_task, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= nqt0 + nqt1) break;
        const int mq = task < nqt1 ? 1 : 0;
        const int qt = mq ? task : task - nqt1;
        const int Lq = a.L[mq];
        const int qi = qt * 16 + lq;
        const bool qvalid = qi < Lq;
        const int qc = qvalid ? qi : Lq - 1;
        bf16x8 qf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = img_row_frag<DHT>(qimg[mq], qt * 16, ks, lane);      // rows past Lq are zero: their results are dropped

        f32x4 s0[NKT[0]], s1[NKT[1]];
        float mx = -INFINITY;
        // scores for both key modalities (uniform branches; arrays statically indexed)
#define VK_SCORES(MK, S)                                                                                   \
        _Pragma("unroll") for (int kt = 0; kt < NKT[MK]; ++kt) {                                           \
            S[kt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};                                     \
            if (a.gate[mq][MK] && kt * 16 < a.L[MK]) {                                                     \
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};                                                          \
                _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                          \
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag<DHT>(kimg[MK], kt * 16, ks, lane), qf[ks], acc, 0, 0, 0); \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                            \
                    const int key = kt * 16 + 4 * g + r;                                                   \
                    if (key < a.L[MK]) {                                                                   \
                        S[kt][r] = acc[r] * a.scale + mask_s[MB[MK] + key];                                \
                        mx = fmaxf(mx, S[kt][r]);                                                          \
                    }                                                                                      \
                }                                                                                          \
            }                                                                                              \
        }
        VK_SCORES(0, s0)
        VK_SCORES(1, s1)
#undef VK_SCORES
        mx = group_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT[0]; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s0[kt][r] = __expf(s0[kt][r] - mx); sum += s0[kt][r]; }
#pragma unroll
        for (int kt = 0; kt < NKT[1]; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s1[kt][r] = __expf(s1[kt][r] - mx); sum += s1[kt][r]; }
        sum = group_sum(sum);
        const float inv = 1.0f / sum;
        if (qvalid && g == 0) a.lse[mq][((size_t)b * a.nh + h) * Lq + qi] = mx + __logf(sum);
        const uint32_t drow = (uint32_t)(((size_t)b * a.nh + h) * Lq + qc);

        f32x4 o[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#define VK_PV(MK, S)                                                                                       \
        if (a.gate[mq][MK]) {                                                                              \
            const vk_dropout dc = a.drop[mq][MK];                                                          \
            const bool don = dc.threshold != 0;                                                            \
            const uint64_t seed = don ? *dc.seed : 0;                                                      \
            _Pragma("unroll") for (int kt = 0; kt < NKT[MK]; ++kt) {                                       \
                u32x4 w = {~0u, ~0u, ~0u, ~0u};                                                            \
                if (don && kt * 16 < a.L[MK])                                                              \
                    w = philox4((uint32_t)(kt * 4 + g), drow, dc.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32)); \
                _Pragma("unroll") for (int r = 0; r < 4; ++r)                                              \
                    S[kt][r] = (w[r] >= dc.threshold) ? S[kt][r] * inv * dc.scale : 0.f;                   \
            }                                                                                              \
            _Pragma("unroll") for (int pp = 0; pp < NKT[MK] / 2; ++pp) {                                   \
                if (pp * 32 < a.L[MK]) {                                                                   \
                    const bf16x8 pb = pack_pair(S[2 * pp], S[2 * pp + 1]);                                 \
                    _Pragma("unroll") for (int dt = 0; dt < DT; ++dt)                                      \
                        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr_frag<DHT>(vimg[MK], pp * 32, dt * 16, lane), pb, o[dt], 0, 0, 0); \
                }                                                                                          \
            }                                                                                              \
        }
        VK_PV(0, s0)
        VK_PV(1, s1)
#undef VK_PV
        if (qvalid) {
            uint16_t* orow = a.ctx[mq] + ((size_t)b * Lq + qi) * a.ldo[mq] + h * DHT + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                *(u32x2*)(orow + dt * 16) = u32x2{pack2bf(o[dt][0], o[dt][1]), pack2bf(o[dt][2], o[dt][3])};
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// OCC = waves per SIMD the register allocation is tuned for: 2 -> <= 256 VGPRs (one 8-wave workgroup per CU), 3 -> <= 168
// VGPRs so that two 5-wave workgroups share a CU and one's staging phase overlaps the other's MFMA phase.
template <int TP, int RP, int OCC, int DHT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void attn_bwd_kernel(const AttnK a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(VK_LDS char*)smem;
    constexpr int PADS[2] = {TP, RP};
    constexpr int NKT[2] = {TP / 16, RP / 16};
    constexpr int ROWS = TP + RP;
    // images: Q, K, V, dO for both modalities; then lse[ROWS], delta[ROWS] floats
    constexpr int RB = 2 * DHT, KS = DHT / 32, DT = DHT / 16;      // image row bytes, 32-deep contraction steps, 16-wide output tiles per head
    const uint32_t qimg[2] = {lds0, lds0 + TP * RB};
    const uint32_t kimg[2] = {qimg[0] + ROWS * RB, qimg[1] + ROWS * RB};
    const uint32_t vimg[2] = {kimg[0] + ROWS * RB, kimg[1] + ROWS * RB};
    const uint32_t gimg[2] = {vimg[0] + ROWS * RB, vimg[1] + ROWS * RB};
    float VK_LDS* lse_s = (float VK_LDS*)(uintptr_t)(lds0 + 4 * ROWS * RB);
    float VK_LDS* del_s = lse_s + ROWS;
    float VK_LDS* const mask_s = del_s + ROWS;                 // additive key masks of this batch element, [TP | RP]
    const int rbase[2] = {0, TP};
    const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
    const int b = blockIdx.x / a.nh, h = blockIdx.x - b * a.nh;
    const int g = lane >> 4, lq = lane & 15;

    bool qact[2], kact[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        qact[m] = a.gate[m][0] || a.gate[m][1];
        kact[m] = a.gate[0][m] || a.gate[1][m];
    }
    {
        const int nthr = blockDim.x;
        const size_t r0 = (size_t)b * a.L[0], r1 = (size_t)b * a.L[1];
        u32x4 rk0[StagePieces<TP, DHT>::N], rv0[StagePieces<TP, DHT>::N], rq0[StagePieces<TP, DHT>::N], rg0[StagePieces<TP, DHT>::N], ro0[StagePieces<TP, DHT>::N];
        u32x4 rk1[StagePieces<RP, DHT>::N], rv1[StagePieces<RP, DHT>::N], rq1[StagePieces<RP, DHT>::N], rg1[StagePieces<RP, DHT>::N], ro1[StagePieces<RP, DHT>::N];
        if (kact[0]) {
            stage_load<TP, DHT>(rk0, a.k[0] + r0 * a.ld[0] + h * DHT, a.ld[0], a.L[0], tid, nthr);
            stage_load<TP, DHT>(rv0, a.v[0] + r0 * a.ld[0] + h * DHT, a.ld[0], a.L[0], tid, nthr);
        }
        if (qact[0]) {
            stage_load<TP, DHT>(rq0, a.q[0] + r0 * a.ld[0] + h * DHT, a.ld[0], a.L[0], tid, nthr);
            stage_load<TP, DHT>(rg0, a.dctx[0] + r0 * a.ldo[0] + h * DHT, a.ldo[0], a.L[0], tid, nthr);
            stage_load<TP, DHT>(ro0, a.ctx[0] + r0 * a.ldo[0] + h * DHT, a.ldo[0], a.L[0], tid, nthr);
        }
        if (kact[1]) {
            stage_load<RP, DHT>(rk1, a.k[1] + r1 * a.ld[1] + h * DHT, a.ld[1], a.L[1], tid, nthr);
            stage_load<RP, DHT>(rv1, a.v[1] + r1 * a.ld[1] + h * DHT, a.ld[1], a.L[1], tid, nthr);
        }
        if (qact[1]) {
            stage_load<RP, DHT>(rq1, a.q[1] + r1 * a.ld[1] + h * DHT, a.ld[1], a.L[1], tid, nthr);
            stage_load<RP, DHT>(rg1, a.dctx[1] + r1 * a.ldo[1] + h * DHT, a.ldo[1], a.L[1], tid, nthr);
            stage_load<RP, DHT>(ro1, a.ctx[1] + r1 * a.ldo[1] + h * DHT, a.ldo[1], a.L[1], tid, nthr);
        }
        float mkv = 0.f;
        if (tid < ROWS) {
            const int m = tid < TP ? 0 : 1, key = tid - rbase[m];
            if (kact[m] && key < a.L[m]) mkv = a.mask[m][(size_t)b * a.L[m] + key];
        }
        if (kact[0]) { stage_store<TP, DHT>(kimg[0], rk0, tid, nthr); stage_store<TP, DHT>(vimg[0], rv0, tid, nthr); }
        if (qact[0]) {
            stage_store<TP, DHT>(qimg[0], rq0, tid, nthr); stage_store<TP, DHT>(gimg[0], rg0, tid, nthr);
            delta_rows<TP, DHT>(rg0, ro0, del_s + rbase[0], lse_s + rbase[0], a.lse[0] + ((size_t)b * a.nh + h) * a.L[0], a.L[0], tid, nthr);
        }
        if (kact[1]) { stage_store<RP, DHT>(kimg[1], rk1, tid, nthr); stage_store<RP, DHT>(vimg[1], rv1, tid, nthr); }
        if (qact[1]) {
            stage_store<RP, DHT>(qimg[1], rq1, tid, nthr); stage_store<RP, DHT>(gimg[1], rg1, tid, nthr);
            delta_rows<RP, DHT>(rg1, ro1, del_s + rbase[1], lse_s + rbase[1], a.lse[1] + ((size_t)b * a.nh + h) * a.L[1], a.L[1], tid, nthr);
        }
        if (tid < ROWS) mask_s[tid] = mkv;
        if (tid == 0) *(int VK_LDS*)(uintptr_t)(lds0 + 4 * ROWS * RB + 3 * ROWS * 4) = 0;
    }
    __syncthreads();

    const int nkt0 = kact[0] ? (a.L[0] + 15) / 16 : 0, nkt1 = kact[1] ? (a.L[1] + 15) / 16 : 0;
    const int nqt0 = qact[0] ? (a.L[0] + 15) / 16 : 0, nqt1 = qact[1] ? (a.L[1] + 15) / 16 : 0;
    const int nktasks = nkt0 + nkt1, ntasks = nktasks + nqt0 + nqt1;
    // Tasks are claimed from a counter in LDS in the order key tiles (two accumulators: the heavier role) before query tiles, vision
    // before text: 10 tasks of unequal weight on 4 waves in fixed round-robin order left one wave with 4.5 units of work against 3.5 on
    // average.
    int VK_LDS* const next_task = (int VK_LDS*)(uintptr_t)(lds0 + 4 * ROWS * RB + 3 * ROWS * 4);
    (void)nwaves;
    for (;;) {
        int task = 0;
        if (lane == 0) task = __hip_atomic_fetch_add(next_task, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= ntasks) break;
        if (task < nktasks) {
            // ---------------- key-tile role: dK, dV of 16 keys of modality mk ----------------
            const int mk = task < nkt1 ? 1 : 0;
            const int kt = mk ? task : task - nkt1;
            const int Lk = a.L[mk];
            const int key = kt * 16 + lq;
            const bool kvalid = key < Lk;
            const float kmask = kvalid ? mask_s[rbase[mk] + key] : 0.f;
            bf16x8 kf[KS], vf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { kf[ks] = img_row_frag<DHT>(kimg[mk], kt * 16, ks, lane); vf[ks] = img_row_frag<DHT>(vimg[mk], kt * 16, ks, lane); }
            f32x4 dk[DT], dv[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#define VK_KROLE(MQ)                                                                                       \
            if (a.gate[MQ][mk]) {                                                                          \
                const int Lq = a.L[MQ];                                                                    \
                const vk_dropout dc = a.drop[MQ][mk];                                                      \
                const bool don = dc.threshold != 0;                                                        \
                const uint64_t seed = don ? *dc.seed : 0;                                                  \
                _Pragma("unroll") for (int pp = 0; pp < NKT[MQ] / 2; ++pp) {                               \
                    if (pp * 32 < Lq) {                                                                    \
                        f32x4 pd[2], ds[2];                                                                \
                        _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                 \
                            const int qt = pp * 2 + hh;                                                    \
                            pd[hh] = f32x4{0.f, 0.f, 0.f, 0.f}; ds[hh] = f32x4{0.f, 0.f, 0.f, 0.f};        \
                            if (qt * 16 >= Lq) continue;         /* tile of padding rows only */          \
                            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};                     \
                            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                            \
                                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag<DHT>(qimg[MQ], qt * 16, ks, lane), kf[ks], s, 0, 0, 0); \
                                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag<DHT>(gimg[MQ], qt * 16, ks, lane), vf[ks], dp, 0, 0, 0); \
                            }                                                                              \
                            /* dropout words: a lane holds one key and needs, for its 4 query rows, word (key & 3) of    */ \
                            /* philox(key >> 2, row).  The 4 lanes of a quad share key >> 2: lane t evaluates row 4g + t */ \
                            /* once and the quad exchanges words (one Philox per lane and tile instead of four).          */ \
                            uint32_t wr[4] = {~0u, ~0u, ~0u, ~0u};                                         \
                            if (don) {                                                                     \
                                const uint32_t drow_t = (uint32_t)(((size_t)b * a.nh + h) * Lq + (qt * 16 + 4 * g + (lq & 3))); \
                                const u32x4 wq = philox4((uint32_t)(key >> 2), drow_t, dc.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32)); \
                                wr[0] = quad_word<0>(wq, lq & 3); wr[1] = quad_word<1>(wq, lq & 3);        \
                                wr[2] = quad_word<2>(wq, lq & 3); wr[3] = quad_word<3>(wq, lq & 3);        \
                            }                                                                              \
                            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                \
                                const int qi = qt * 16 + 4 * g + r;                                        \
                                float p = 0.f, keep = 1.f;                                                 \
                                if (qi < Lq && kvalid) {                                                   \
                                    p = __expf(s[r] * a.scale + kmask - lse_s[rbase[MQ] + qi]);            \
                                    if (don) keep = (wr[r] >= dc.threshold) ? dc.scale : 0.f;              \
                                }                                                                          \
                                pd[hh][r] = p * keep;                                                      \
                                ds[hh][r] = p * (dp[r] * keep - del_s[rbase[MQ] + (qi < PADS[MQ] ? qi : 0)]); \
                            }                                                                              \
                        }                                                                                  \
                        const bf16x8 pb = pack_pair(pd[0], pd[1]), sb = pack_pair(ds[0], ds[1]);           \
                        _Pragma("unroll") for (int dt = 0; dt < DT; ++dt) {                                \
                            dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr_frag<DHT>(gimg[MQ], pp * 32, dt * 16, lane), pb, dv[dt], 0, 0, 0); \
                            dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr_frag<DHT>(qimg[MQ], pp * 32, dt * 16, lane), sb, dk[dt], 0, 0, 0); \
                        }                                                                                  \
                    }                                                                                      \
                }                                                                                          \
            }
            VK_KROLE(0)
            VK_KROLE(1)
#undef VK_KROLE
            if (kvalid) {
                const size_t off = ((size_t)b * Lk + key) * a.ldg[mk] + h * DHT + 4 * g;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    *(u32x2*)(a.dk[mk] + off + dt * 16) = u32x2{pack2bf(dk[dt][0] * a.scale, dk[dt][1] * a.scale), pack2bf(dk[dt][2] * a.scale, dk[dt][3] * a.scale)};
                    *(u32x2*)(a.dv[mk] + off + dt * 16) = u32x2{pack2bf(dv[dt][0], dv[dt][1]), pack2bf(dv[dt][2], dv[dt][3])};
                }
            }
        } else {
            // ---------------- query-tile role: dQ of 16 queries of modality mq ----------------
            const int t2 = task - nktasks;
            const int mq = t2 < nqt1 ? 1 : 0;
            const int qt = mq ? t2 : t2 - nqt1;
            const int Lq = a.L[mq];
            const int qi = qt * 16 + lq;
            const bool qvalid = qi < Lq;
            bf16x8 qf[KS], gf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { qf[ks] = img_row_frag<DHT>(qimg[mq], qt * 16, ks, lane); gf[ks] = img_row_frag<DHT>(gimg[mq], qt * 16, ks, lane); }
            const float lse = lse_s[rbase[mq] + qt * 16 + lq], delta = del_s[rbase[mq] + qt * 16 + lq];
            const uint32_t drow = (uint32_t)(((size_t)b * a.nh + h) * Lq + (qvalid ? qi : 0));
            f32x4 dq[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#define VK_QROLE(MK)                                                                                       \
            if (a.gate[mq][MK]) {                                                                          \
                const int Lk = a.L[MK];                                                                    \
                const vk_dropout dc = a.drop[mq][MK];                                                      \
                const bool don = dc.threshold != 0;                                                        \
                const uint64_t seed = don ? *dc.seed : 0;                                                  \
                _Pragma("unroll") for (int pp = 0; pp < NKT[MK] / 2; ++pp) {                               \
                    if (pp * 32 < Lk) {                                                                    \
                        f32x4 ds[2];                                                                       \
                        _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                 \
                            const int kt = pp * 2 + hh;                                                    \
                            ds[hh] = f32x4{0.f, 0.f, 0.f, 0.f};                                            \
                            if (kt * 16 >= Lk) continue;         /* tile of padding keys only */          \
                            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};                     \
                            _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                            \
                                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag<DHT>(kimg[MK], kt * 16, ks, lane), qf[ks], s, 0, 0, 0); \
                                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_row_frag<DHT>(vimg[MK], kt * 16, ks, lane), gf[ks], dp, 0, 0, 0); \
                            }                                                                              \
                            u32x4 w = {~0u, ~0u, ~0u, ~0u};                                                \
                            if (don) w = philox4((uint32_t)(kt * 4 + g), drow, dc.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32)); \
                            _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                \
                                const int key = kt * 16 + 4 * g + r;                                       \
                                float v = 0.f;                                                             \
                                if (key < Lk && qvalid) {                                                  \
                                    const float p = __expf(s[r] * a.scale + mask_s[rbase[MK] + key] - lse);      \
                                    const float keep = (w[r] >= dc.threshold) ? dc.scale : 0.f;            \
                                    v = p * (dp[r] * keep - delta);                                        \
                                }                                                                          \
                                ds[hh][r] = v;                                                             \
                            }                                                                              \
                        }                                                                                  \
                        const bf16x8 sb = pack_pair(ds[0], ds[1]);                                         \
                        _Pragma("unroll") for (int dt = 0; dt < DT; ++dt)                                  \
                            dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(img_tr_frag<DHT>(kimg[MK], pp * 32, dt * 16, lane), sb, dq[dt], 0, 0, 0); \
                    }                                                                                      \
                }                                                                                          \
            }
            VK_QROLE(0)
            VK_QROLE(1)
#undef VK_QROLE
            if (qvalid) {
                const size_t off = ((size_t)b * Lq + qi) * a.ldg[mq] + h * DHT + 4 * g;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    *(u32x2*)(a.dq[mq] + off + dt * 16) = u32x2{pack2bf(dq[dt][0] * a.scale, dq[dt][1] * a.scale), pack2bf(dq[dt][2] * a.scale, dq[dt][3] * a.scale)};
            }
        }
    }
}

// launch-shape constants (measured); mutable only in VK_STUDY builds (tools/bench_small.py)
#ifdef VK_STUDY
#define VK_ATTN_TUNABLE static int
#else
#define VK_ATTN_TUNABLE static constexpr int
#endif

static int fill(AttnK& k, const vk_attn_args* a, const vk_attn_bwd_args* bw) {
    for (int m = 0; m < 2; ++m) {
        k.q[m] = (const uint16_t*)a->q[m]; k.k[m] = (const uint16_t*)a->k[m]; k.v[m] = (const uint16_t*)a->v[m];
        k.ld[m] = a->ld[m]; k.L[m] = a->L[m]; k.mask[m] = a->mask[m];
        k.ctx[m] = (uint16_t*)a->ctx[m]; k.ldo[m] = a->ldo[m]; k.lse[m] = a->lse[m];
        for (int n = 0; n < 2; ++n) { k.gate[m][n] = a->gate[m][n]; k.drop[m][n] = a->drop[m][n]; }
        k.dctx[m] = bw ? (const uint16_t*)bw->dctx[m] : nullptr;
        k.dq[m] = bw ? (uint16_t*)bw->dq[m] : nullptr; k.dk[m] = bw ? (uint16_t*)bw->dk[m] : nullptr;
        k.dv[m] = bw ? (uint16_t*)bw->dv[m] : nullptr; k.ldg[m] = bw ? bw->ldg[m] : 0;
    }
    k.B = a->B; k.nh = a->nh; k.scale = a->scale;
    for (int m = 0; m < 2; ++m) {
        const bool qa = a->gate[m][0] || a->gate[m][1], ka = a->gate[0][m] || a->gate[1][m];
        if (!qa && !ka) continue;
        if (a->L[m] <= 0) return set_error("vk_gated_attn: modality %d is gated on but has length %d", m, a->L[m]);
        if ((a->ld[m] & 7) || (a->ldo[m] & 7)) return set_error("vk_gated_attn: row strides must be multiples of 8");
        if (ka && (!a->k[m] || !a->v[m] || !a->mask[m])) return set_error("vk_gated_attn: K/V/mask of modality %d missing", m);
        if (qa && (!a->q[m] || !a->ctx[m] || !a->lse[m])) return set_error("vk_gated_attn: Q/ctx/lse of modality %d missing", m);
        if (bw && qa && (!bw->dctx[m] || !bw->dq[m])) return set_error("vk_gated_attn_bwd: dctx/dq of modality %d missing", m);
        if (bw && ka && (!bw->dk[m] || !bw->dv[m])) return set_error("vk_gated_attn_bwd: dk/dv of modality %d missing", m);
    }
    if (a->L[0] > 64 || a->L[1] > 128) return set_error("vk_gated_attn: lengths (%d, %d) exceed the (64, 128) tile budget", a->L[0], a->L[1]);
    return 0;
}

VK_ATTN_TUNABLE g_attn_fwd_waves = 4;    // waves per workgroup (query tiles are looped): 4 workgroups of 4 waves fill the CU's 16 wave slots, one wave per tile (5) leaves it at 3 workgroups (measured -18 %)
VK_ATTN_TUNABLE g_attn_fwd_occ = 4;      // tuning hook: waves per SIMD the register allocation targets (5 spills 16-74 dwords: slower)
template <int TP, int RP, int DHT>
static int launch_fwd(const AttnK& k, int nq_tiles, hipStream_t s) {
    const int lds = 3 * (TP + RP) * 2 * DHT + (TP + RP) * 4 + 16;          // K, V, Q images, key masks, the task counter
    int waves = nq_tiles < 4 ? 4 : (nq_tiles > 8 ? 8 : nq_tiles);      // staging is sized for >= 256 threads
    if (g_attn_fwd_waves >= 4 && g_attn_fwd_waves < waves && 4 * lds <= 160 * 1024) waves = g_attn_fwd_waves;     // only where four workgroups fit the CU's LDS
    if (DHT == 64 && g_attn_fwd_occ == 5) {
        hipLaunchKernelGGL((attn_fwd_kernel<TP, RP, 5, DHT>), dim3(k.B * k.nh), dim3(64 * waves), lds, s, k);
    } else {
        auto kern = attn_fwd_kernel<TP, RP, (DHT == 64 ? 4 : 3), DHT>;      // 128-wide heads: twice the output accumulators, 3 waves per SIMD
        static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); (void)attr;      // once per process, thread-safe
        hipLaunchKernelGGL(kern, dim3(k.B * k.nh), dim3(64 * waves), lds, s, k);
    }
    return check_launch("vk_gated_attn_fwd");
}
VK_ATTN_TUNABLE g_attn_bwd_waves = 4;    // waves per workgroup (tasks are looped): 4 lets two workgroups share a CU (measured -15 %)
VK_ATTN_TUNABLE g_attn_bwd_occ = 3;      // waves per SIMD the register allocation targets: 3 (<= 168 VGPRs, a few spilled dwords) lets three 4-wave workgroups share a CU (-3..-9 % against 2)

template <int TP, int RP, int DHT>
static int launch_bwd(const AttnK& k, int ntasks, hipStream_t s) {
    const int lds = 4 * (TP + RP) * 2 * DHT + 3 * (TP + RP) * 4 + 16;          // four images, lse / delta / key masks, the task counter
    if (DHT == 64 && g_attn_bwd_occ == 3 && 3 * lds <= 160 * 1024) {       // three workgroups only fit with the small images
        auto kern = attn_bwd_kernel<TP, RP, 3, DHT>;
        static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); (void)attr;      // once per process, thread-safe
        int waves = g_attn_bwd_waves < 4 ? 4 : (ntasks > g_attn_bwd_waves ? g_attn_bwd_waves : (ntasks < 4 ? 4 : ntasks));
        hipLaunchKernelGGL(kern, dim3(k.B * k.nh), dim3(64 * waves), lds, s, k);
    } else {
        auto kern = attn_bwd_kernel<TP, RP, 2, DHT>;
        static const hipError_t attr = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); (void)attr;      // once per process, thread-safe
        int waves = g_attn_bwd_waves < 4 ? 4 : (ntasks > g_attn_bwd_waves ? g_attn_bwd_waves : (ntasks < 4 ? 4 : ntasks));      // staging is sized for >= 256 threads
        // images so large that only ONE workgroup fits a CU (VL-BERT at 100 regions: 32 + 128 padded rows = 82 KiB): four waves would leave half
        // the CU's wave slots empty (508 us per launch, profiles/r04_vlbert_r100_kernel_stats.md) -- give the workgroup all eight
        if (2 * lds > 160 * 1024) waves = ntasks >= 8 ? 8 : (ntasks < 4 ? 4 : ntasks);
        hipLaunchKernelGGL(kern, dim3(k.B * k.nh), dim3(64 * waves), lds, s, k);
    }
    return check_launch("vk_gated_attn_bwd");
}

// LDS of the backward's four images per row (+ statistics): what decides whether 128-wide heads fit the MFMA kernel
static bool bwd_fits(int TP, int RP, int dht) { return 4 * (TP + RP) * 2 * dht + 3 * (TP + RP) * 4 + 16 <= 160 * 1024; }

}  // namespace vk

namespace vk { int attn_generic(const vk_attn_args* a, const vk_attn_bwd_args* bw, vk_stream_t stream); }      // attention_generic.hip

// Head sizes: 64 (every ctrl_* config) and 128 (config/vilbert_base.json) run on the MFMA kernels above; 128 with the longest sequences
// (more than 96 padded rows: the backward's images exceed the LDS), the other sizes (32, 96) and every head size with rows beyond the
// MFMA tiles (more than 64 text tokens or 128 regions: the task configs of config_tasks/all_tasks.yml) on the generic kernels.
#ifdef VK_STUDY
static int g_attn_force_generic = 0;      // A/B hook: 128-wide heads on the generic kernels
extern "C" void vk_attn_set_force_generic(int v) { g_attn_force_generic = v; }
#else
static constexpr int g_attn_force_generic = 0;
#endif
static bool attn_mfma_ok(const vk_attn_args* a) {
    if (a->probs[0][0] || a->probs[0][1] || a->probs[1][0] || a->probs[1][1]) return false;      // attention maps requested: the generic kernels write them
    for (int m = 0; m < 2; ++m) {             // rows beyond the MFMA kernels' tiles (64 text tokens, 128 regions): the generic kernels, at any head size
        const bool on = a->gate[m][0] || a->gate[m][1] || a->gate[0][m] || a->gate[1][m];
        if (on && a->L[m] > (m == 0 ? 64 : 128)) return false;
    }
    if (a->dh == 0 || a->dh == 64) return true;
    if (a->dh != 128 || g_attn_force_generic) return false;
    const int TP = a->L[0] > 32 ? 64 : 32, RP = a->L[1] > 64 ? 128 : 64;
    return vk::bwd_fits(TP, RP, 128);           // forward and backward of one sub-layer take the same path
}

namespace vk { size_t attn_generic_lds(const vk_attn_args* a, bool bwd); }

extern "C" size_t vk_gated_attn_lds_bytes(const vk_attn_args* a, int backward) {
    if (!a || attn_mfma_ok(a)) return 0;
    return vk::attn_generic_lds(a, backward != 0);
}

extern "C" int vk_gated_attn_fwd(const vk_attn_args* a, vk_stream_t stream) {
    using namespace vk;
    if (!attn_mfma_ok(a)) return attn_generic(a, nullptr, stream);
    AttnK k;
    if (int rc = fill(k, a, nullptr)) return rc;
    if (a->B <= 0) return 0;
    int nq = 0;
    for (int m = 0; m < 2; ++m) if (a->gate[m][0] || a->gate[m][1]) nq += (a->L[m] + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    const bool bigT = a->L[0] > 32, bigR = a->L[1] > 64;
    if (a->dh == 128) {
        if (!bigT && !bigR) return launch_fwd<32, 64, 128>(k, nq, s);
        return launch_fwd<64, 64, 128>(k, nq, s);          // (bigR never fits: attn_mfma_ok)
    }
    if (!bigT && !bigR) return launch_fwd<32, 64, 64>(k, nq, s);
    if (bigT && !bigR) return launch_fwd<64, 64, 64>(k, nq, s);
    if (!bigT && bigR) return launch_fwd<32, 128, 64>(k, nq, s);
    return launch_fwd<64, 128, 64>(k, nq, s);
}

extern "C" int vk_gated_attn_bwd(const vk_attn_args* a, const vk_attn_bwd_args* bw, vk_stream_t stream) {
    using namespace vk;
    if (!attn_mfma_ok(a)) return attn_generic(a, bw, stream);
    AttnK k;
    if (int rc = fill(k, a, bw)) return rc;
    if (a->B <= 0) return 0;
    int nt = 0;
    for (int m = 0; m < 2; ++m) {
        if (a->gate[m][0] || a->gate[m][1]) nt += (a->L[m] + 15) / 16;
        if (a->gate[0][m] || a->gate[1][m]) nt += (a->L[m] + 15) / 16;
    }
    hipStream_t s = (hipStream_t)stream;
    const bool bigT = a->L[0] > 32, bigR = a->L[1] > 64;
    if (a->dh == 128) {
        if (!bigT && !bigR) return launch_bwd<32, 64, 128>(k, nt, s);
        return launch_bwd<64, 64, 128>(k, nt, s);
    }
    if (!bigT && !bigR) return launch_bwd<32, 64, 64>(k, nt, s);
    if (bigT && !bigR) return launch_bwd<64, 64, 64>(k, nt, s);
    if (!bigT && bigR) return launch_bwd<32, 128, 64>(k, nt, s);
    return launch_bwd<64, 128, 64>(k, nt, s);
}
#ifdef VK_STUDY
extern "C" void vk_attn_set_bwd_occupancy(int v) { vk::g_attn_bwd_occ = v; }
extern "C" void vk_attn_set_bwd_waves(int v) { vk::g_attn_bwd_waves = v; }
extern "C" void vk_attn_set_fwd_waves(int v) { vk::g_attn_fwd_waves = v; }
extern "C" void vk_attn_set_fwd_occupancy(int v) { vk::g_attn_fwd_occ = v; }
#endif
