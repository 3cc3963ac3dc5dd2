// Grouped bf16 GEMM for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate): dispatcher + the 128 x 128 tile.
// Three operand layouts, fused epilogues (gemm_common.h):
//   NT  both operands K-contiguous            -> fragments by ds_read_b128 from an XOR-swizzled image
//   NN  B stored [K][N]                       -> B fragments by ds_read_b64_tr_b16 (hardware transpose)
//   TN  A stored [K][M], B stored [K][N]      -> both by ds_read_b64_tr_b16
// Two geometries:
//   128 x 128, 4 waves (64 x 64 each), 32 KiB LDS per stage, several workgroups per CU (this file).  Measured on
//       MI355X (profiles/r01_gemm_*.txt) it needs 32 KiB of LDS fill per 512 MFMA cycles, i.e. the whole LDS write
//       rate of a CU, and saturates at ~45 % of the MFMA peak; it serves the problems too small to fill the chip
//       with 256^2 tiles.
//   256 x 256, 8 waves, 128 KiB LDS, 8-phase LDS-DMA pipeline (gemm256.hip) for everything else.
// Staging: bounds-checked LDS-DMA (buffer_load ... lds, 16 B / lane, out-of-range rows read as 0, which is
// what zero-pads ragged M / N / K) into a double buffer, the swizzle lives on the per-lane GLOBAL source
// address and on the read address (cdna guide rule 21).  The 128^2 NN / TN kernels stage through registers
// instead (loads of two K-tiles in flight, ds_write_b128 after the MFMAs), which measured faster there.
// MFMA operands are swapped (D = B-frag x A-frag) so that every lane owns 4 consecutive output columns of
// one output row: 8-byte bf16 / 16-byte fp32 stores.
//
// Replaces every nn.Linear forward/backward of volta/encoders.py and volta/embeddings.py
// (site list in include/volta_hip.h).
#include "gemm_common.h"
#include <cstdlib>

namespace vk {

template <int WM, int WN, int TM = 1> struct Geo {      // TM: 64-row blocks per wave (wave tile = 64*TM x 64)
    static constexpr int THREADS = 64 * WM * WN, BM = 64 * TM * WM, BN = 64 * WN;
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
};

// Stage one operand tile of EXT rows (T=false: image [EXT][64 k], 128-B rows) or EXT columns (T=true: image
// [64 k][EXT], 2*EXT-byte rows) with all THREADS threads; 16-byte chunks, linear LDS image.
template <bool T, int EXT, int THREADS>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rs, uint32_t lds_tile, int ld, int row0, int col0, int tid) {
    constexpr int NP = EXT * 8 / THREADS;          // pieces per thread
    constexpr int CPR = T ? EXT / 8 : 8;           // 16-byte chunks per image row
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int lin = i * THREADS + tid;
        const int r = lin / CPR, cp = lin % CPR;
        const int cl = T ? (cp ^ tswz(r)) : (cp ^ (r & 7));
        const uint32_t voff = ((uint32_t)(row0 + r) * (uint32_t)ld + (uint32_t)(col0 + cl * 8)) * 2u;
        const uint32_t dst = lds_tile + (i * THREADS + wave * 64) * 16;   // wave-uniform; hardware adds lane*16
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (VK_LDS void*)(uintptr_t)dst, 16, voff, 0, 0, 0);
    }
}
template <bool T, int EXT, int THREADS>
__device__ __forceinline__ void stage_load(__amdgpu_buffer_rsrc_t rs, u32x4 (&reg)[EXT * 8 / THREADS], int ld, int row0, int col0, int tid) {
    constexpr int NP = EXT * 8 / THREADS;
    constexpr int CPR = T ? EXT / 8 : 8;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int lin = i * THREADS + tid;
        const int r = lin / CPR, cp = lin % CPR;
        const int cl = T ? (cp ^ tswz(r)) : (cp ^ (r & 7));
        const uint32_t voff = ((uint32_t)(row0 + r) * (uint32_t)ld + (uint32_t)(col0 + cl * 8)) * 2u;
        reg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
    }
}
template <int NP, int THREADS>
__device__ __forceinline__ void stage_store(uint32_t lds_tile, const u32x4 (&reg)[NP], int tid) {
#pragma unroll
    for (int i = 0; i < NP; ++i) *(u32x4 VK_LDS*)(uintptr_t)(lds_tile + (i * THREADS + tid) * 16) = reg[i];
}

template <bool AT, bool BT, int EPI, int WM, int WN, bool REGSTAGE, int TM = 1>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(WM * WN == 4 ? (REGSTAGE ? 2 : 3) : 1)))
void gemm_kernel(const KGroup g) {
    using G = Geo<WM, WN, TM>;
    constexpr int TI = 4 * TM;            // 16-row MFMA tiles per wave along M
    constexpr int THREADS = G::THREADS, BM = G::BM, BN = G::BN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(VK_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int bid = (g.stagger & GROUP_PLAIN_ORDER) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < VK_GEMM_MAX_GROUP; ++i)
        if (i < g.nprob && bid >= g.p[i].tile_start) pi = i;
    const KProb& P = g.p[pi];
    const int t = bid - P.tile_start;
    const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    int M = P.M, K = P.K;
    if (P.dyn) {
        const int d = *P.dyn;
        if (AT) K = d < K ? d : K; else M = d < M ? d : M;
    }
    if (m0 >= M) return;

    // buffer extents: last valid row + valid row length (everything beyond reads as zero)
    const int a_rows = AT ? K : M, a_cols = AT ? P.lda : even_up(K, P.lda);
    const int b_rows = BT ? K : P.N, b_cols = BT ? P.ldb : even_up(K, P.ldb);
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(P.A, a_rows > 0 ? (uint32_t)(((uint32_t)(a_rows - 1) * P.lda + a_cols) * 2u) : 0u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(P.B, b_rows > 0 ? (uint32_t)(((uint32_t)(b_rows - 1) * P.ldb + b_cols) * 2u) : 0u);

    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias_grad = AT && BT && (P.bias_grad != nullptr) && (tn == 0) && (wn == 0);
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (short)0x3F80;

    const int nk = (K + BK - 1) / BK;
    auto compute_ks = [&](int cur, int ks) {
        const uint32_t ta = lds0 + cur * G::STAGE, tb = ta + G::A_BYTES;
        {
            bf16x8 a[TI], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                b[i] = BT ? frag_cols<BN * 2>(tb, wn * 64 + i * 16, ks, lane) : frag_rows(tb, wn * 64 + i * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < TI; ++i)
                a[i] = AT ? frag_cols<BM * 2>(ta, wm * 64 * TM + i * 16, ks, lane) : frag_rows(ta, wm * 64 * TM + i * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            if (do_bias_grad) {
#pragma unroll
                for (int i = 0; i < TI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[i], accb[i], 0, 0, 0);
            }
        }
    };
    auto compute = [&](int cur) { compute_ks(cur, 0); compute_ks(cur, 1); };

    if constexpr (REGSTAGE) {
        // register-staged, prefetch distance 2: tile kt is multiplied out of LDS while tile kt+1 waits in one register
        // set (written to the other LDS buffer after the MFMAs) and the loads of tile kt+2 are issued into the second.
        constexpr int NPA = BM * 8 / THREADS, NPB = BN * 8 / THREADS;
        u32x4 ra0[NPA], rb0[NPB], ra1[NPA], rb1[NPB];
#define VK_GLOAD(RA, RB, KT)                                                                               \
        do {                                                                                               \
            if (AT) stage_load<true, BM, THREADS>(rsA, RA, P.lda, (KT) * BK, m0, tid);                      \
            else stage_load<false, BM, THREADS>(rsA, RA, P.lda, m0, (KT) * BK, tid);                       \
            if (BT) stage_load<true, BN, THREADS>(rsB, RB, P.ldb, (KT) * BK, n0, tid);                      \
            else stage_load<false, BN, THREADS>(rsB, RB, P.ldb, n0, (KT) * BK, tid);                       \
        } while (0)
#define VK_LSTORE(RA, RB, BUF)                                                                             \
        do {                                                                                               \
            stage_store<NPA, THREADS>(lds0 + (BUF) * G::STAGE, RA, tid);                                   \
            stage_store<NPB, THREADS>(lds0 + (BUF) * G::STAGE + G::A_BYTES, RB, tid);                      \
        } while (0)
        if (nk > 0) VK_GLOAD(ra0, rb0, 0);
        if (nk > 1) VK_GLOAD(ra1, rb1, 1);
        if (nk > 0) VK_LSTORE(ra0, rb0, 0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            if (kt + 2 < nk) VK_GLOAD(ra0, rb0, kt + 2);
            compute(0);
            if (kt + 1 < nk) VK_LSTORE(ra1, rb1, 1);
            __syncthreads();
            if (kt + 1 < nk) {
                if (kt + 3 < nk) VK_GLOAD(ra1, rb1, kt + 3);
                compute(1);
                if (kt + 2 < nk) VK_LSTORE(ra0, rb0, 0);
                __syncthreads();
            }
        }
#undef VK_GLOAD
#undef VK_LSTORE
    } else {
        // LDS-DMA double buffer: the DMA of tile kt+1 is issued right after the barrier that retires tile kt-1's reads
        auto stage = [&](int buf, int kt) {
            const uint32_t ta = lds0 + buf * G::STAGE, tb = ta + G::A_BYTES;
            if (AT) stage_tile<true, BM, THREADS>(rsA, ta, P.lda, kt * BK, m0, tid);
            else    stage_tile<false, BM, THREADS>(rsA, ta, P.lda, m0, kt * BK, tid);
            if (BT) stage_tile<true, BN, THREADS>(rsB, tb, P.ldb, kt * BK, n0, tid);
            else    stage_tile<false, BN, THREADS>(rsB, tb, P.ldb, n0, kt * BK, tid);
        };
        if (nk > 0) stage(0, 0);
        // Waves 4..7 and 12..15 of a 16-wave workgroup share their SIMDs with waves 0..3 / 8..11: they issue the DMA of
        // the next tile after their first 16 MFMAs instead of before them, so that on every SIMD two waves feed the
        // matrix pipe while the other two pay the LDS-DMA issue cost (60-185 cycles per piece).
        const bool late = (g.stagger & 0xFF) && (WM * WN > 4) && ((__builtin_amdgcn_readfirstlane(wave) >> 2) & 1);   // SIMD partners: w, w+4, ...
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // tile kt landed for every wave; everyone is done reading tile kt-1
            if (!late && kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
            compute_ks(kt & 1, 0);
            if (late && kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
            compute_ks(kt & 1, 1);
        }
    }

    gemm_epilogue<AT, EPI, TI, 4>(P, acc, accb, do_bias_grad, m0 + wm * 64 * TM, n0 + wn * 64, M, lane);
    retire_mark(g);
}

template <bool AT, bool BT, int WM, int WN, bool REGSTAGE, int TM = 1>
static int launch_cfg(int epi, const KGroup& g, int total, hipStream_t s) {
    using G = Geo<WM, WN, TM>;
    constexpr int LDS = 2 * G::STAGE;
#define VK_CASE(E)                                                                                        \
    case E: {                                                                                             \
        auto k = gemm_kernel<AT, BT, E, WM, WN, REGSTAGE, TM>;                                                \
        static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); (void)attr; /* once per process, thread-safe */ \
        hipLaunchKernelGGL(k, dim3(total), dim3(G::THREADS), LDS, s, g);                                  \
        break;                                                                                            \
    }
    switch (epi) {
        VK_CASE(VK_EPI_BF16) VK_CASE(VK_EPI_GELU) VK_CASE(VK_EPI_MULR) VK_CASE(VK_EPI_ADDR) VK_CASE(VK_EPI_F32) VK_CASE(VK_EPI_RELU) VK_CASE(VK_EPI_F32_ACC)
        default: return set_error("vk_gemm_grouped: unknown epilogue %d", epi);
    }
#undef VK_CASE
    return check_launch("vk_gemm_grouped");
}

// Geometry selection constants (measured).  In VK_STUDY builds (libvolta_hip_study.so, tools/ only) they can be
// overridden through exported setters / the environment; the shipped library has no mutable dispatch state.
#ifdef VK_STUDY
#define VK_TUNABLE static int
#else
#define VK_TUNABLE static constexpr int
#endif
VK_TUNABLE g_stagger = 1;
VK_TUNABLE g_min_tiles256_tn = 50;   // TN (weight gradients, side stream): fewer, longer workgroups leave more CUs to the critical path (19.71 -> 19.54 ms)
VK_TUNABLE g_min_tiles256 = 160;     // fewest 256 x 256 tiles for which that geometry is chosen (256 CUs)
VK_TUNABLE g_debug = 0;              // ablation switches of the 4-phase study kernel (gemm256.hip)
VK_TUNABLE g_regstage_override = -1; // -1 = heuristic, 0 / 1 = force LDS-DMA / register staging (128^2 only)
VK_TUNABLE g_persistent = 1;         // 0: never take the persistent kernel
#ifdef VK_STUDY
static int g_tile_override = 0;      // geometry forced for plain vk_gemm_grouped calls (0 = heuristic)
#endif

static int total_tiles(const vk_gemm_problem* probs, int nprob, int epilogue, int bm, int bn) {
    int total = 0;
    for (int i = 0; i < nprob; ++i) {
        const vk_gemm_problem& q = probs[i];
        const bool f32 = epilogue == VK_EPI_F32 || epilogue == VK_EPI_F32_ACC;
        const int ncols = (f32 && q.n_store > q.N) ? q.n_store : q.N;
        total += ((q.M + bm - 1) / bm) * ((ncols + bn - 1) / bn);
    }
    return total;
}

}  // namespace vk

static int gemm_dispatch(int layout, int epilogue, const vk_gemm_problem* probs, int nprob, int geometry, vk_stream_t stream) {
    using namespace vk;
    if (nprob < 1 || nprob > VK_GEMM_MAX_GROUP) return set_error("vk_gemm_grouped: nprob %d out of range", nprob);
    const bool f32out = epilogue == VK_EPI_F32 || epilogue == VK_EPI_F32_ACC;
    bool any_dyn = false, any_split = false;
    for (int i = 0; i < nprob; ++i) {
        const vk_gemm_problem& q = probs[i];
        if (q.M < 0 || q.N <= 0 || q.K < 0) return set_error("vk_gemm_grouped: bad shape %d %d %d", q.M, q.N, q.K);
        if ((q.lda & 7) || (q.ldb & 7)) return set_error("vk_gemm_grouped: lda/ldb must be multiples of 8 (got %d %d)", q.lda, q.ldb);
        if (((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15) || ((uintptr_t)q.C & 15)) return set_error("vk_gemm_grouped: operands must be 16-byte aligned");
        if (!f32out && (q.ldc & 3)) return set_error("vk_gemm_grouped: ldc must be a multiple of 4");
        if (layout != VK_TN && (q.K % 64) != 0 && q.lda < ((q.K + 63) / 64) * 64)
            return set_error("vk_gemm_grouped: K=%d needs lda padded to a multiple of 64", q.K);
        if (layout != VK_TN && q.bias_grad) return set_error("vk_gemm_grouped: bias_grad is a TN (wgrad) feature");
        if ((epilogue == VK_EPI_MULR || epilogue == VK_EPI_ADDR) && !q.R) return set_error("vk_gemm_grouped: R missing");
        if (epilogue == VK_EPI_GELU && !q.C2) return set_error("vk_gemm_grouped: C2 missing");
        any_dyn |= q.dyn != nullptr;
        any_split |= q.nparts > 1;
        if (q.nparts > 1) {
            if (!q.ws || !q.cnt || q.part < 0 || q.part >= q.nparts) return set_error("vk_gemm_grouped: split accumulation needs ws, cnt and 0 <= part < nparts");
            if (q.dyn) return set_error("vk_gemm_grouped: split accumulation does not take a device-side row count");
            if (((uintptr_t)q.ws & 15)) return set_error("vk_gemm_grouped: ws must be 16-byte aligned");
            if (q.nparts > 30) return set_error("vk_gemm_grouped: a split accumulation has at most 30 parts (got %d)", q.nparts);
            int seen = 0, count = 0;
            for (int j = 0; j < nprob; ++j)
                if (probs[j].ws == q.ws && probs[j].nparts > 1) {
                    if (probs[j].M != q.M || probs[j].N != q.N || probs[j].C != q.C || probs[j].cnt != q.cnt || probs[j].nparts != q.nparts)
                        return set_error("vk_gemm_grouped: the parts of one split accumulation must agree in M, N, C, cnt and nparts");
                    if (probs[j].part < 0 || probs[j].part >= q.nparts) return set_error("vk_gemm_grouped: split accumulation part %d out of range [0, %d)", probs[j].part, q.nparts);
                    seen |= 1 << probs[j].part;
                    ++count;               // a duplicated part beside a complete set would hand the tile nparts + 1 tickets
                }
            if (count != q.nparts || seen != (1 << q.nparts) - 1) return set_error("vk_gemm_grouped: a split accumulation needs each of its %d parts exactly once in the launch", q.nparts);
        }
    }
#ifdef VK_STUDY
    static const bool env_once = [] {       // tuning overrides from the environment (study builds only)
        if (const char* e = getenv("VK_GEMM_MIN_TILES256")) g_min_tiles256 = atoi(e);
        if (const char* e = getenv("VK_GEMM_MIN_TILES256_TN")) g_min_tiles256_tn = atoi(e);
        if (const char* e = getenv("VK_GEMM_TILE")) g_tile_override = atoi(e);
        if (const char* e = getenv("VK_GEMM_PERSISTENT")) g_persistent = atoi(e);
        return true;
    }();
    (void)env_once;
    if (geometry == 0) geometry = g_tile_override;
#endif
    // Geometry codes: 128 x 128 [128], 256 x 256 K-split [258], 256 x 192 K-split [259], 256 x 128 K-split [260: never faster than
    // 128 x 128 on the model shapes], 4-phase study kernel [256, study builds]; + VK_GEMM_PERSISTENT / VK_GEMM_ONE_TILE_PER_WG force the
    // tile walk.  The heuristic takes 256-row tiles whenever they still
    // yield >= g_min_tiles256 workgroups, and of the two widths the one with less work on the busiest CU:
    // rounds(tiles / 256 CUs) x tile width.  N = 768 -> 4 column tiles of 192 instead of 3 of 256 (228 instead of
    // 171 workgroups for the ViLBERT row counts: one round of smaller tiles), N = 2304 -> 12 instead of 9.
    if (probs[0].retire_flag && (!probs[0].retire_stamp || ((uintptr_t)probs[0].retire_flag & 7) || ((uintptr_t)probs[0].retire_stamp & 7)))
        return set_error("vk_gemm_grouped: retire_flag needs retire_stamp, both 8-byte aligned");
    const bool soft = (geometry & VK_GEMM_SOFT_START) != 0;
    bool any_handoff = false;
    for (int i = 0; i < nprob; ++i) any_handoff |= probs[i].sig != nullptr || probs[i].dep != nullptr;
    const int walk = geometry & (VK_GEMM_PERSISTENT | VK_GEMM_ONE_TILE_PER_WG);
    int edge = geometry & 0xFFF;
    if (edge == 0) {
        const int t256 = total_tiles(probs, nprob, epilogue, 256, 256), t192 = total_tiles(probs, nprob, epilogue, 256, 192);
        if (t256 < (layout == VK_TN ? g_min_tiles256_tn : g_min_tiles256)) edge = 128;
        else {
            const double c256 = (double)((t256 + 255) / 256) * 256.0, c192 = (double)((t192 + 255) / 256) * 192.0;
            edge = c192 < 0.97 * c256 ? 259 : 258;
        }
    }
    if (edge == 128 && (geometry & 0xFFF) == 0 && layout == VK_NT && !any_dyn) {
        // launches too small for 256-row tiles, K-contiguous operands: the 4-wave ring kernel with 128 x 128 tiles (five K-steps in flight,
        // one workgroup per CU) instead of the double-buffered one -- text-only FFN-down 56.5 -> 48.1 us, attention output 22.0 -> 19.0
        // (profiles/r03_gemm_shapes.txt); the transposed-operand layouts measured no better on it
        bool deep = true;
        for (int i = 0; i < nprob; ++i) deep &= probs[i].M >= 1024 && probs[i].K >= 512;
        if (deep) edge = 262;
    }
    if (edge == 256 || edge == 258 || edge == 259 || edge == 260 || edge == 261 || edge == 262) {
        for (int i = 0; i < nprob; ++i) {
            const vk_gemm_problem& q = probs[i];
            const uint64_t ea = (uint64_t)(layout == VK_TN ? q.K : q.M) * q.lda * 2, eb = (uint64_t)(layout == VK_NT ? q.N : q.K) * q.ldb * 2;
            if (ea >= 0x7FFFFFF0ull || eb >= 0x7FFFFFF0ull) { edge = 128; break; }     // the LDS-DMA kernels address operands below 2 GiB
        }
    }
    if (edge != 256 && edge != 258 && edge != 259 && edge != 260 && edge != 261 && edge != 262) edge = 128;
    if ((any_handoff || soft) && (layout == VK_TN || any_dyn || any_split || (edge != 258 && edge != 259)))
        return set_error("vk_gemm_grouped: row-block hand-off (sig / dep / VK_GEMM_SOFT_START) runs on the NT / NN layouts with the 256-row geometries 258 / 259, without dyn or split accumulation (layout %d, geometry %d)", layout, geometry & 0xFFF);
    if (any_handoff) {
        const int bn_ = edge == 259 ? 192 : 256;
        for (int i = 0; i < nprob; ++i) {
            const vk_gemm_problem& q = probs[i];
            if (!q.sig && !q.dep) continue;
            // whole tiles only: the handed-off rows leave the epilogue as full-line write-through stores (no ragged path), and a row block is 256 rows
            if (q.sig && ((q.M & 255) || (q.N % bn_) || f32out || ((q.ldc * 2) & 127) || ((uintptr_t)q.C & 127) || (edge == 259)))
                return set_error("vk_gemm_grouped: a signalling problem needs M %% 256 == 0, N %% 256 == 0 under geometry 258, a bf16 C with 128-byte aligned rows (M %d N %d ldc %d)", q.M, q.N, q.ldc);
            if (q.dep && (q.dep_need <= 0 || (q.M & 255))) return set_error("vk_gemm_grouped: a guarded problem needs dep_need > 0 and M %% 256 == 0");
        }
    }
    if (any_split && edge != 258 && edge != 259) return set_error("vk_gemm_grouped: split accumulation runs on the 256 x 256 / 256 x 192 geometries (258 / 259), got %d", geometry & 0xFFF);
    const int bm = (edge == 128 || edge == 262) ? 128 : 256, bn = edge == 259 ? 192 : (edge == 260 || edge == 261 || edge == 262) ? 128 : bm;
    KGroup g{};
    g.nprob = nprob;
    g.stagger = (g_stagger & 0xFF) | ((g_debug & 0xFF) << 8);
    // Problems with a device-side row count (heads on labelled rows) keep only their first few row tiles alive; the
    // XCD-chunked order would hand all of them to XCD 0 (LM decoder: 240 live tiles on 32 CUs, 188 us instead of ~40).
    if (any_dyn) g.stagger |= GROUP_PLAIN_ORDER;
    int total = 0;
    for (int i = 0; i < nprob; ++i) {
        const vk_gemm_problem& q = probs[i];
        KProb& k = g.p[i];
        k.A = (const char*)q.A; k.B = (const char*)q.B; k.C = (char*)q.C; k.C2 = (char*)q.C2; k.bias = q.bias;
        k.R = (const char*)q.R; k.bias_grad = q.bias_grad; k.dyn = q.dyn; k.C8 = nullptr; k.c8_mul = 0.f; k.ldc8 = 0;
        k.M = q.M; k.N = q.N; k.K = q.K; k.lda = q.lda; k.ldb = q.ldb; k.ldc = q.ldc; k.ldr = q.ldr; k.n_store = q.n_store;
        k.ws = (char*)q.ws; k.cnt = q.cnt; k.part = q.part; k.nparts = q.nparts;
        k.sig = q.sig; k.dep = q.dep; k.err = q.err; k.dep_need = q.dep_need;
        if (i == 0) { g.retire_flag = (unsigned long long*)q.retire_flag; g.retire_stamp = (const unsigned long long*)q.retire_stamp; }
        const int ncols = (f32out && q.n_store > q.N) ? q.n_store : q.N;
        k.tiles_n = (ncols + bn - 1) / bn;
        k.tile_start = total;
        total += ((q.M + bm - 1) / bm) * k.tiles_n;
    }
    if (total == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (edge == 261 || edge == 262) return launch_gemm4w(layout, epilogue, g, total, s, edge == 262 ? 128 : 256);      // 4-wave ring kernels
    if (edge == 256 || edge == 258 || edge == 259 || edge == 260) {
        // more than one round of tiles: one workgroup per CU walks the list (gemm256p_kernel); device-side row counts keep the
        // one-tile-per-workgroup launch (dead tiles exit at once there)
        bool persistent = (edge == 258 || edge == 259) && layout != VK_TN && !any_dyn && !any_split;
        if (walk == VK_GEMM_PERSISTENT) persistent = persistent && true;
        else if (walk == VK_GEMM_ONE_TILE_PER_WG) persistent = false;
        else persistent = persistent && g_persistent && total > NUM_CU;
        return launch_gemm256(layout, epilogue, g, total, s, edge == 258 ? 4 : edge == 259 ? 3 : edge == 260 ? 2 : 0, persistent, soft);
    }
    {
        const bool reg = g_regstage_override >= 0 ? g_regstage_override != 0 : layout != VK_NT;
        if (layout == VK_NT) return reg ? launch_cfg<false, false, 2, 2, true>(epilogue, g, total, s) : launch_cfg<false, false, 2, 2, false>(epilogue, g, total, s);
        if (layout == VK_NN) return reg ? launch_cfg<false, true, 2, 2, true>(epilogue, g, total, s) : launch_cfg<false, true, 2, 2, false>(epilogue, g, total, s);
        if (layout == VK_TN) return reg ? launch_cfg<true, true, 2, 2, true>(epilogue, g, total, s) : launch_cfg<true, true, 2, 2, false>(epilogue, g, total, s);
    }
    return set_error("vk_gemm_grouped: unknown layout %d", layout);
}

extern "C" size_t vk_gemm_split_workspace_bytes(int layout, int M, int N, int nparts, int geometry, int* tiles) {
    const int edge = geometry & 0xFFF;
    if ((edge != 258 && edge != 259) || M <= 0 || N <= 0 || nparts < 2) {
        if (tiles) *tiles = 0;
        return 0;
    }
    const int bn = edge == 259 ? 192 : 256, tj = bn / 64;
    const int nt = ((M + 255) / 256) * ((N + bn - 1) / bn);
    if (tiles) *tiles = nt;
    const size_t per = (size_t)8 * (size_t)(8 * tj + (layout == VK_TN ? 8 : 0)) * 1024u;       // split_slab_bytes<8, TJ, BG>(8)
    return (size_t)nt * (size_t)nparts * per;
}

extern "C" int vk_gemm_chain(int layout, int epi_p, const vk_gemm_problem* prod, int np, int epi_c, const vk_gemm_problem* cons, int nc, vk_stream_t stream) {
    using namespace vk;
    if (np < 1 || nc < 1 || np + nc > VK_GEMM_MAX_GROUP) return set_error("vk_gemm_chain: %d + %d problems (1 .. %d in all)", np, nc, VK_GEMM_MAX_GROUP);
    if (layout != VK_NT && layout != VK_NN) return set_error("vk_gemm_chain: layout %d (NT / NN)", layout);
    KGroup g{};
    g.nprob = np + nc;
    g.stagger = 0;
    int total = 0, nprod = 0;
    for (int i = 0; i < np + nc; ++i) {
        const bool is_p = i < np;
        const vk_gemm_problem& q = is_p ? prod[i] : cons[i - np];
        const int bn = is_p ? 256 : 192;
        if (q.M <= 0 || q.N <= 0 || q.K <= 0 || (q.M & 255) || (q.N % bn) || (q.K & 63)) return set_error("vk_gemm_chain: problem %d: whole tiles only (M %% 256, N %% %d, K %% 64; got %d %d %d)", i, bn, q.M, q.N, q.K);
        if ((q.lda & 7) || (q.ldb & 7) || (q.ldc & 3) || ((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15) || ((uintptr_t)q.C & 15)) return set_error("vk_gemm_chain: operand alignment (problem %d)", i);
        if (q.dyn || q.nparts > 1 || q.bias_grad) return set_error("vk_gemm_chain: no dyn / split accumulation / bias gradient (problem %d)", i);
        const uint64_t ea = (uint64_t)q.M * q.lda * 2, eb = (uint64_t)(layout == VK_NT ? q.N : q.K) * q.ldb * 2;
        if (ea >= 0x7FFFFFF0ull || eb >= 0x7FFFFFF0ull) return set_error("vk_gemm_chain: operands must stay below 2 GiB");
        if (is_p) {
            if (!q.sig || q.dep || ((q.ldc * 2) & 127) || ((uintptr_t)q.C & 127)) return set_error("vk_gemm_chain: producer %d needs sig, no dep, and C rows 128-byte aligned", i);
            if (epi_p == VK_EPI_GELU && !q.C2) return set_error("vk_gemm_chain: C2 missing");
            if (epi_p == VK_EPI_MULR && !q.R) return set_error("vk_gemm_chain: R missing");
        } else {
            // a consumer reads the C of ONE producer through its A operand and waits for that producer's counters: every column tile of the row block
            int src = -1;
            for (int j = 0; j < np; ++j)
                if (prod[j].sig == q.dep) src = j;
            if (src < 0 || q.sig) return set_error("vk_gemm_chain: consumer %d: dep must be a producer's sig (and no sig of its own)", i - np);
            const vk_gemm_problem& pr = prod[src];
            if (q.A != pr.C || q.M != pr.M || q.K != pr.N || q.lda != pr.ldc || q.dep_need != pr.N / 256)
                return set_error("vk_gemm_chain: consumer %d does not read producer %d's C as its A operand (A, M, K, lda, dep_need = N / 256)", i - np, src);
            if (epi_c == VK_EPI_ADDR && !q.R) return set_error("vk_gemm_chain: R missing");
        }
        KProb& k = g.p[i];
        k.A = (const char*)q.A; k.B = (const char*)q.B; k.C = (char*)q.C; k.C2 = (char*)q.C2; k.bias = q.bias;
        k.R = (const char*)q.R; k.bias_grad = nullptr; k.dyn = nullptr; k.C8 = nullptr; k.c8_mul = 0.f; k.ldc8 = 0;
        k.M = q.M; k.N = q.N; k.K = q.K; k.lda = q.lda; k.ldb = q.ldb; k.ldc = q.ldc; k.ldr = q.ldr; k.n_store = 0;
        k.ws = nullptr; k.cnt = nullptr; k.part = 0; k.nparts = 0;
        k.sig = q.sig; k.dep = q.dep; k.err = q.err; k.dep_need = q.dep_need;
        k.tiles_n = q.N / bn;
        k.tile_start = total;
        total += (q.M / 256) * k.tiles_n;
        if (is_p) nprod = total;
    }
    g.retire_flag = (unsigned long long*)prod[0].retire_flag; g.retire_stamp = (const unsigned long long*)prod[0].retire_stamp;
    if (g.retire_flag && (!g.retire_stamp || ((uintptr_t)g.retire_flag & 7) || ((uintptr_t)g.retire_stamp & 7))) return set_error("vk_gemm_chain: retire_flag needs retire_stamp, both 8-byte aligned");
    return launch_gemm256_chain(layout, epi_p, epi_c, g, nprod, total - nprod, (hipStream_t)stream);
}

extern "C" int vk_gemm_grouped(int layout, int epilogue, const vk_gemm_problem* probs, int nprob, vk_stream_t stream) {
    return gemm_dispatch(layout, epilogue, probs, nprob, 0, stream);
}

extern "C" int vk_gemm_grouped_ex(int layout, int epilogue, const vk_gemm_problem* probs, int nprob, int geometry, vk_stream_t stream) {
    return gemm_dispatch(layout, epilogue, probs, nprob, geometry, stream);
}

#ifdef VK_STUDY
/* measurement hooks for tools/bench_gemm.py (study builds only) */
extern "C" int vk_gemm_repeat(int layout, int epilogue, const vk_gemm_problem* probs, int nprob, int iters, vk_stream_t stream) {
    for (int i = 0; i < iters; ++i) {       // back-to-back launches from native code: no interpreter time between kernels
        const int rc = vk_gemm_grouped(layout, epilogue, probs, nprob, stream);
        if (rc != 0) return rc;
    }
    return 0;
}
extern "C" void vk_gemm_set_tile(int edge) { vk::g_tile_override = edge; }
extern "C" void vk_gemm_set_regstage(int v) { vk::g_regstage_override = v; }
extern "C" void vk_gemm_set_stagger(int v) { vk::g_stagger = v; }
extern "C" void vk_gemm_set_debug(int v) { vk::g_debug = v; }
extern "C" void vk_gemm_set_min_tiles256(int v) { vk::g_min_tiles256 = v; }
extern "C" void vk_gemm_set_persistent(int v) { vk::g_persistent = v; }
#endif
