// 256 x 256 x 64 bf16 GEMM tile for gfx950: 8 waves (2 along M x 4 along N, 128 x 64 outputs each), 128 KiB of
// LDS, LDS-DMA staging with counted vmcnt, and a K loop of 4 phases per K-tile in which the two halves of the
// workgroup run half a phase apart -- on every SIMD one wave multiplies while its partner reads LDS and issues
// the next DMA (the "8-phase" structure of the CDNA4 guide, section 5).
//
// LDS holds 2 K-tiles x 4 half-tiles of 16 KiB.  A half-tile is SUB-TILE-major: it holds, for every wave, the
// part of the operand that one phase consumes,
//     A0 / A1 : rows  wr*128 + s*64 + [0,64)   of the 256-row A tile (s = 0 / 1),   image [128][64 k]
//     B0 / B1 : cols  wc*64  + s*32 + [0,32)   of the 256-col B tile,               image [128][64 k]
// (transposed operands: image [64 k][128], read with ds_read_b64_tr_b16), so the four half-tiles of a K-tile
// are needed -- and die -- one phase after another:
//     phase 1: read B0, A0   acc[0:4][0:2] += A0 B0        phase 3: read A1       acc[4:8][2:4] += A1 B1
//     phase 2: read B1       acc[0:4][2:4] += A0 B1        phase 4: (registers)   acc[4:8][0:2] += A1 B0
// Half-tiles are staged in consumption order, index idx = 4*kt + {A0:0, B0:1, B1:2, A1:3}; global phase
// g = 4*kt + p stages idx g + 5 into the slot of idx g - 3, whose last read was in phase <= g - 2 (WAR: a slot
// is re-staged >= 2 phases after its last ds_read).  Every wave waits vmcnt(8) after its stage -- the 4 newest
// half-tiles (2 DMA instructions each) stay in flight, everything up to idx g + 1 has landed -- BEFORE the
// phase's first barrier, and idx g + 1 is first read in phase g + 1 (RAW: read one phase after the wait).
// Stages past the end of K are issued with an out-of-range offset (the buffer bounds check turns them into
// zero fills without memory traffic) so that the vmcnt arithmetic is the same in every iteration.
#include "gemm_common.h"
#include <hip/hip_ext.h>
#include <type_traits>
#include <atomic>

namespace vk {

constexpr uint32_t HT = 16384;          // bytes of one half-tile slot
constexpr uint32_t OOB = 0x80000000u;   // beyond any operand (host checks extents < 2 GiB)

// Byte offsets (s = 0, kt = 0) of the two 16-byte pieces a thread stages of a half-tile; the XOR swizzle of the
// LDS image is applied here, on the global source (the LDS-DMA destination is lane-linear).
template <bool T, bool IS_A>
__device__ __forceinline__ void stage_offsets(uint32_t (&off)[2], int ld, int ext0, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lin = i * 512 + tid;
        if (!T) {
            const int hr = lin >> 3, cp = lin & 7;                       // image row, 16-byte chunk position
            const int row = IS_A ? ((hr >> 6) * 128 + (hr & 63)) : ((hr >> 5) * 64 + (hr & 31));
            off[i] = ((uint32_t)(ext0 + row) * (uint32_t)ld + (uint32_t)((cp ^ (hr & 7)) * 8)) * 2u;
        } else {
            const int kr = lin >> 4, cp = lin & 15;
            const int c = cp ^ tswz(kr);
            const int col = IS_A ? ((c >> 3) * 128 + (c & 7) * 8) : ((c >> 2) * 64 + (c & 3) * 8);
            off[i] = ((uint32_t)kr * (uint32_t)ld + (uint32_t)(ext0 + col)) * 2u;
        }
    }
}

template <int AUX = 0>      // AUX 2 = non-temporal: the operand's last use (a weight gradient's saved activation), kept from displacing live tensors in the Infinity Cache
__device__ __forceinline__ void stage_half(__amdgpu_buffer_rsrc_t rs, uint32_t slot, const uint32_t (&off)[2], uint32_t add, int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint32_t dst = slot + (i * 512 + wave * 64) * 16;       // wave-uniform; hardware adds lane * 16
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (VK_LDS void*)(uintptr_t)dst, 16, off[i] + add, 0, 0, AUX);
    }
}

#define VK_SYNC()                                 \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)
#define VK_WAIT_DMA() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define VK_WAIT_LDS()                                          \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_sched_barrier(0);                     \
    } while (0)

#ifdef VK_STUDY      // the 4-phase kernel and its ablation switches: measurement code, not shipped in libvolta_hip.so
template <int V> using ic = std::integral_constant<int, V>;

template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(const KGroup g) {
    constexpr bool BG = AT && BT;          // bias gradient (column sums of A) rides on the wgrad layout only
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(VK_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int bid = (g.stagger & GROUP_PLAIN_ORDER) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < VK_GEMM_MAX_GROUP; ++i)
        if (i < g.nprob && bid >= g.p[i].tile_start) pi = i;
    const KProb& P = g.p[pi];
    const int t = bid - P.tile_start;
    const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;

    int M = P.M, K = P.K;
    if (P.dyn) {
        const int d = *P.dyn;
        if (AT) K = d < K ? d : K; else M = d < M ? d : M;
    }
    if (m0 >= M) return;

    const int a_rows = AT ? K : M, a_cols = AT ? P.lda : even_up(K, P.lda);
    const int b_rows = BT ? K : P.N, b_cols = BT ? P.ldb : even_up(K, P.ldb);
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(P.A, a_rows > 0 ? (uint32_t)(((uint32_t)(a_rows - 1) * P.lda + a_cols) * 2u) : 0u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(P.B, b_rows > 0 ? (uint32_t)(((uint32_t)(b_rows - 1) * P.ldb + b_cols) * 2u) : 0u);

    uint32_t offA[2], offB[2];
    stage_offsets<AT, true>(offA, P.lda, m0, tid);
    stage_offsets<BT, false>(offB, P.ldb, n0, tid);
    const uint32_t sA = AT ? 128u : 128u * (uint32_t)P.lda, kA = AT ? 128u * (uint32_t)P.lda : 128u;   // bytes per sub-tile / K-tile
    const uint32_t sB = BT ? 64u : 64u * (uint32_t)P.ldb, kB = BT ? 128u * (uint32_t)P.ldb : 128u;
    const int nk = (K + BK - 1) / BK;
    const int dbg = (g.stagger >> 8) & 0xFF;               // ablation switches (tools/bench_gemm.py): 1 no DMA, 2 no MFMA, 4 no LDS reads, 8 no epilogue, 16 exit at once, 32 no K loop
    bool in_loop = false;
    if (dbg & 16) return;

    auto stage = [&](auto X_, int kt) {           // X: 0 = A0, 1 = B0, 2 = B1, 3 = A1
        if ((dbg & 1) && in_loop) return;
        constexpr int X = decltype(X_)::value;
        constexpr bool isA = (X == 0 || X == 3);
        constexpr uint32_t s = (X >= 2) ? 1u : 0u;
        const uint32_t slot = lds0 + (uint32_t)((kt & 1) * 4 + X) * HT;
        const bool live = kt < nk;
        if (isA) stage_half(rsA, slot, offA, live ? s * sA + (uint32_t)kt * kA : OOB, wave);
        else     stage_half(rsB, slot, offB, live ? s * sB + (uint32_t)kt * kB : OOB, wave);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias_grad = BG && (P.bias_grad != nullptr) && (tn == 0) && (wc == 0);
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (short)0x3F80;

    bf16x8 a[4][2], b0[2][2], b1[2][2];
    auto load_a = [&](uint32_t slot) {
        if (dbg & 4) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                a[i][ks] = AT ? frag_cols<256>(slot, wr * 64 + i * 16, ks, lane) : frag_rows(slot, wr * 64 + i * 16, ks, lane);
    };
    auto load_b = [&](bf16x8 (&b)[2][2], uint32_t slot) {
        if (dbg & 4) return;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                b[j][ks] = BT ? frag_cols<256>(slot, wc * 32 + j * 16, ks, lane) : frag_rows(slot, wc * 32 + j * 16, ks, lane);
    };
    auto quad = [&](auto I0_, auto J0_, const bf16x8 (&b)[2][2], bool with_bias) {
        constexpr int I0 = decltype(I0_)::value, J0 = decltype(J0_)::value;
        if (dbg & 2) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[I0 + i][J0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][ks], a[i][ks], acc[I0 + i][J0 + j], 0, 0, 0);
        if (BG && with_bias && do_bias_grad) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    accb[I0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[i][ks], accb[I0 + i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };

    // prologue: idx 0..5 in flight, idx 0 and 1 (A0, B0 of K-tile 0) landed
    stage(ic<0>{}, 0); stage(ic<1>{}, 0); stage(ic<2>{}, 0); stage(ic<3>{}, 0); stage(ic<0>{}, 1); stage(ic<1>{}, 1);
    VK_WAIT_DMA();
    VK_SYNC();
    if (wr == 1) VK_SYNC();        // the upper half of the workgroup runs half a phase behind
    in_loop = true;

    for (int kt = 0; kt < ((dbg & 32) ? 0 : nk); ++kt) {
        const uint32_t buf = lds0 + (uint32_t)(kt & 1) * 4u * HT;
        // ---- phase 1
        load_b(b0, buf + 1 * HT);
        __builtin_amdgcn_sched_barrier(0);
        load_a(buf + 0 * HT);
        stage(ic<2>{}, kt + 1);
        VK_WAIT_DMA();
        VK_SYNC();
        VK_WAIT_LDS();
        quad(ic<0>{}, ic<0>{}, b0, true);
        VK_SYNC();
        // ---- phase 2
        load_b(b1, buf + 2 * HT);
        stage(ic<3>{}, kt + 1);
        VK_WAIT_DMA();
        VK_SYNC();
        VK_WAIT_LDS();
        quad(ic<0>{}, ic<2>{}, b1, false);
        VK_SYNC();
        // ---- phase 3
        load_a(buf + 3 * HT);
        stage(ic<0>{}, kt + 2);
        VK_WAIT_DMA();
        VK_SYNC();
        VK_WAIT_LDS();
        quad(ic<4>{}, ic<2>{}, b1, true);
        VK_SYNC();
        // ---- phase 4
        stage(ic<1>{}, kt + 2);
        VK_WAIT_DMA();
        VK_SYNC();
        quad(ic<4>{}, ic<0>{}, b0, false);
        VK_SYNC();
    }
    if (wr == 0) VK_SYNC();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // retire the zero-fill stages issued past the end of K
    if (dbg & 8) return;
    VK_SYNC();                                            // ... of every wave: LDS is now free for the epilogue's transposition

    gemm_epilogue<AT, EPI, 8, 4>(P, acc, accb, do_bias_grad, m0 + wr * 128, n0 + wc * 64, M, lane, lds0 + (uint32_t)wave * 16384u);
}


#endif  // VK_STUDY

// ---------------------------------------------------------------------------------------------------------
// Variant B: one phase per 32-deep K-step.  Every phase reads the whole wave tile's fragments (8 A + 4 B,
// 12 ds_read_b128), issues the DMA of one K-step (A and B strips of 256 x 32, 16 KiB each, 4 instructions per
// wave) and runs 32 MFMAs -- half the barriers of the 4-phase loop and the same LDS work in every phase.
// LDS is a ring of 5 K-steps (160 KiB).  Phase P stages K-step P + 3 into the slot K-step P - 2 was read from
// (WAR: >= 2 phases after its last ds_read) and waits vmcnt(8): K-steps P + 2 and P + 3 stay in flight, K-step
// P + 1 has landed and is first read one phase later (RAW).
// Strip image, K-contiguous operand: [256 rows][32 k], 64-byte rows, 16-byte chunk c of row r stored at chunk
// c ^ f(r), f(r) = (-(r >> 2)) & 3 -- with that every 16-lane group of a ds_read_b128 covers the 16 slots of
// a 256-byte bank row once.  Transposed operand: [32 k][256], 512-byte rows, read with ds_read_b64_tr_b16.
__device__ __forceinline__ int kswz(int r) { return (-(r >> 2)) & 3; }

constexpr uint32_t MASKED = 0x7FFFFFF0u;     // per-lane offset of a piece outside the tile: stays out of range for every K-step

// `ext` rows (K-contiguous operand) / columns (transposed operand) of the 256-wide strip image belong to the tile;
// the pieces beyond are issued out of range (zero fill, no traffic) so that every wave issues the same DMA count.
template <bool T>
__device__ __forceinline__ void strip_offsets(uint32_t (&off)[2], int ld, int ext0, int ext, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lin = i * 512 + tid;
        if (!T) {
            const int r = lin >> 2, cp = lin & 3;
            off[i] = r < ext ? ((uint32_t)(ext0 + r) * (uint32_t)ld + (uint32_t)((cp ^ kswz(r)) * 8)) * 2u : MASKED;
        } else {
            const int kr = lin >> 5, cp = lin & 31;
            const int c = cp ^ tswz(kr);
            off[i] = c * 8 < ext ? ((uint32_t)kr * (uint32_t)ld + (uint32_t)(ext0 + c * 8)) * 2u : MASKED;
        }
    }
}

__device__ __forceinline__ bf16x8 frag_strip(uint32_t strip, int r0, int lane) {
    const int r = r0 + (lane & 15);
    return *(const bf16x8 VK_LDS*)(uintptr_t)(strip + r * 64 + (((lane >> 4) ^ kswz(r)) << 4));
}

#ifdef VK_STUDY
__device__ unsigned long long* g_kstamps = nullptr;      // tools/stamp_soft.py: start / end s_memrealtime of every workgroup of the one-tile kernel
__device__ int g_desync_ticks = 0;                       // tools/stamp_gemm.py: every other workgroup of an XCD starts this many 10 ns ticks late (are lockstep epilogues the cost?)
#endif

// TJ = 16-column tiles per wave along N: 4 -> 256 x 256 tile, 3 -> 256 x 192 (N = 768 / 2304 split into 4 / 12 column
// tiles instead of 3 / 9: 171 -> 228 and 513 -> 684 workgroups for the ViLBERT shapes, i.e. fuller rounds of smaller tiles).
template <bool AT, bool BT, int EPI, int TJ>
__global__ __launch_bounds__(512) void gemm256k_kernel(const KGroup g) {
    constexpr bool BG = AT && BT;
    constexpr int RING = 5;
    constexpr int BN = 64 * TJ, WN = 16 * TJ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(VK_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int bid = (g.stagger & GROUP_PLAIN_ORDER) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
#ifdef VK_STUDY
    if (g_kstamps && tid == 0) g_kstamps[blockIdx.x * 4] = __builtin_amdgcn_s_memrealtime();
#endif
    int pi = 0;
#pragma unroll
    for (int i = 1; i < VK_GEMM_MAX_GROUP; ++i)
        if (i < g.nprob && bid >= g.p[i].tile_start) pi = i;
    const KProb& P = g.p[pi];
    const int t = bid - P.tile_start;
    const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
    const int m0 = tm * 256, n0 = tn * BN;

    int M = P.M, K = P.K;
    if (P.dyn) {
        const int d = *P.dyn;
        if (AT) K = d < K ? d : K; else M = d < M ? d : M;
    }
    if (m0 >= M) return;

    const int a_rows = AT ? K : M, a_cols = AT ? P.lda : even_up(K, P.lda);
    const int b_rows = BT ? K : P.N, b_cols = BT ? P.ldb : even_up(K, P.ldb);
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(P.A, a_rows > 0 ? (uint32_t)(((uint32_t)(a_rows - 1) * P.lda + a_cols) * 2u) : 0u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(P.B, b_rows > 0 ? (uint32_t)(((uint32_t)(b_rows - 1) * P.ldb + b_cols) * 2u) : 0u);

    uint32_t offA[2], offB[2];
    strip_offsets<AT>(offA, P.lda, m0, 256, tid);
    strip_offsets<BT>(offB, P.ldb, n0, BN, tid);
    const uint32_t kA = AT ? 64u * (uint32_t)P.lda : 64u, kB = BT ? 64u * (uint32_t)P.ldb : 64u;     // bytes per 32-deep K-step
    const int np = (K + 31) / 32;
    const bool guarded = !AT && P.dep != nullptr;          // A rows handed over by an earlier launch that may still be running (soft boundary)
    auto stage = [&](int p, int slot) {
        const bool live = p < np;
        const uint32_t sa = lds0 + (uint32_t)slot * 2u * HT;
        if (guarded) stage_half<AUX_SC1>(rsA, sa, offA, live ? (uint32_t)p * kA : OOB, wave);
        else stage_half(rsA, sa, offA, live ? (uint32_t)p * kA : OOB, wave);
        stage_half<(AT && BT) ? 2 : 0>(rsB, sa + HT, offB, live ? (uint32_t)p * kB : OOB, wave);
    };
    if (guarded) soft_wait(P.dep, tm, P.dep_need, P.err);
#ifdef VK_STUDY
    if (g_kstamps && tid == 0) g_kstamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
#endif

    f32x4 acc[8][TJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias_grad = BG && (P.bias_grad != nullptr) && (tn == 0) && (wc == 0);
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (short)0x3F80;

    stage(0, 0); stage(1, 1); stage(2, 2);
    VK_WAIT_DMA();
    VK_SYNC();
    if (wr == 1) VK_SYNC();        // the upper half of the workgroup runs half a phase behind
    int rd = 0, wrs = 3;           // ring slots of the K-step read / staged in this phase
    for (int p = 0; p < np; ++p) {
        const uint32_t sa = lds0 + (uint32_t)rd * 2u * HT, sb = sa + HT;
        bf16x8 a[8], b[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j) b[j] = BT ? frag_cols<512>(sb, wc * WN + j * 16, 0, lane) : frag_strip(sb, wc * WN + j * 16, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = AT ? frag_cols<512>(sa, wr * 128 + i * 16, 0, lane) : frag_strip(sa, wr * 128 + i * 16, lane);
        stage(p + 3, wrs);
        VK_WAIT_DMA();
        VK_SYNC();
        VK_WAIT_LDS();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        if (BG && do_bias_grad) {
#pragma unroll
            for (int i = 0; i < 8; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[i], accb[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        VK_SYNC();
        rd = rd == RING - 1 ? 0 : rd + 1;
        wrs = wrs == RING - 1 ? 0 : wrs + 1;
    }
    if (wr == 0) VK_SYNC();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VK_SYNC();

    if (P.nparts > 1) {            // K-slice of a split accumulation: only the workgroup that arrives last at the tile goes on
        if (!split_combine<8, TJ, BG, 8>(P, t, acc, accb, wave, lane, lds0)) return;
        __syncthreads();           // the ticket word lies in wave 0's staging region
    }
    gemm_epilogue<AT, EPI, 8, TJ>(P, acc, accb, do_bias_grad, m0 + wr * 128, n0 + wc * WN, M, lane, lds0 + (uint32_t)wave * 16384u);
    if (!AT && P.sig) soft_signal(P.sig, tm);
    retire_mark(g);
#ifdef VK_STUDY
    if (g_kstamps && tid == 0) g_kstamps[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Persistent form of the K-split kernel for launches of more than one round of tiles (NT / NN; 684 tiles for the dual-stream
// Q|K|V, FFN-up and FFN-down-dgrad GEMMs of ViLBERT): one workgroup per CU walks tiles v = blockIdx.x, + gridDim.x, ...
// Between two tiles the first three K-steps of the NEXT tile are requested (ring slots 0-2) BEFORE the epilogue of the current one
// runs out of slots 3-4 (8 KiB per wave), so the epilogue's conversion / LDS transposition / store issue hides the operand latency
// of the next tile, the output stores drain into L2 under the next tile's K loop instead of holding the CU until the workgroup
// retires, and launch ramp + end-of-kernel write-back are paid once per launch instead of once per round.
// vmcnt counts loads and stores together in issue order: the next tile's prologue loads are OLDER than the epilogue's stores, so the
// loop's first `vmcnt(8)` (all but the 8 youngest operations done) covers them -- conservatively, it also waits for most stores.
// Tile iterators of the persistent walk: next() yields the next tile index of THIS workgroup (already in XCD-chunked order) or -1.
struct StrideTiles {            // v = blockIdx.x, + gridDim.x, ...: the whole tile list of a launch
    int v, total;
    __device__ __forceinline__ int next() { const int t = v < total ? xcd_remap(v, total) : -1; v += (int)gridDim.x; return t; }
};
// A chain launch (vk_gemm_chain): tiles [0, nprod) are the producers' (walked first, by every workgroup), [nprod, nprod + ncons) the
// consumers'.  Consumer tiles are dealt in REVERSE inside an XCD's chunk: the workgroups with one producer tile fewer (the high
// blockIdx.x) finish first and take the chunk's first row blocks, which the XCD completes first.
struct ProducerTiles {
    int v, nprod;
    __device__ __forceinline__ int next() { const int t = v < nprod ? xcd_remap(v, nprod) : -1; v += (int)gridDim.x; return t; }
};
struct ConsumerTiles {
    int k, nprod, ncons;        // k: this workgroup's rank inside its XCD, + workgroups per XCD per round
    __device__ __forceinline__ int next() {
        const int x = blockIdx.x & 7, q = ncons >> 3, r = ncons & 7;
        const int len = q + (x < r ? 1 : 0), start = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        const int local = len - 1 - k;
        k += (int)(gridDim.x >> 3);
        return local >= 0 ? nprod + start + local : -1;
    }
};

template <bool AT, bool BT, int EPI, int TJ, typename Tiles>
__device__ __forceinline__ void persistent_walk(const KGroup& g, Tiles tiles, unsigned long long* const stamps) {
    // `stamps` (study builds; NULL in the shipped library's launches): per workgroup and tile four s_memrealtime readings (100 MHz) --
    // loop top, K loop done, ring drained + next prologue issued, epilogue done -- into a buffer no other code reads.
    static_assert(!(AT && BT), "the persistent kernel serves the NT / NN layouts (no bias-gradient accumulators)");
    constexpr int RING = 5;
    constexpr int BN = 64 * TJ, WN = 16 * TJ;
    constexpr uint32_t EPI_BASE = 6u * HT, EPI_REGION = 8192u;       // ring slots 3 and 4
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(VK_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    int pi, m0, n0, np;
    bool guarded;
    __amdgpu_buffer_rsrc_t rsA, rsB;
    uint32_t offA[2], offB[2], kA, kB;
    auto setup = [&](int bid) {
        pi = 0;
#pragma nounroll
        for (int i = 1; i < g.nprob; ++i)              // scalar loop: an unrolled search keeps 32 tile_start words live across the tile loop
            if (bid >= g.p[i].tile_start) pi = i;
        const KProb& P = g.p[pi];
        const int t = bid - P.tile_start;
        const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
        m0 = tm * 256; n0 = tn * BN;
        const int M = P.M, K = P.K;
        const int a_rows = AT ? K : M, a_cols = AT ? P.lda : even_up(K, P.lda);
        const int b_rows = BT ? K : P.N, b_cols = BT ? P.ldb : even_up(K, P.ldb);
        rsA = make_rsrc(P.A, a_rows > 0 ? (uint32_t)(((uint32_t)(a_rows - 1) * P.lda + a_cols) * 2u) : 0u);
        rsB = make_rsrc(P.B, b_rows > 0 ? (uint32_t)(((uint32_t)(b_rows - 1) * P.ldb + b_cols) * 2u) : 0u);
        strip_offsets<AT>(offA, P.lda, m0, 256, tid);
        strip_offsets<BT>(offB, P.ldb, n0, BN, tid);
        kA = AT ? 64u * (uint32_t)P.lda : 64u; kB = BT ? 64u * (uint32_t)P.ldb : 64u;
        np = (K + 31) / 32;
        guarded = P.dep != nullptr;
    };
    auto stage = [&](int p, int slot) {
        const bool live = p < np;
        const uint32_t sa = lds0 + (uint32_t)slot * 2u * HT;
        if (guarded) stage_half<AUX_SC1>(rsA, sa, offA, live ? (uint32_t)p * kA : OOB, wave);
        else stage_half(rsA, sa, offA, live ? (uint32_t)p * kA : OOB, wave);
        stage_half(rsB, sa + HT, offB, live ? (uint32_t)p * kB : OOB, wave);
    };

    int tile = tiles.next();
    if (tile < 0) return;
#ifdef VK_STUDY
    if (g_desync_ticks > 0 && ((blockIdx.x >> 3) & 1)) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)g_desync_ticks) __builtin_amdgcn_s_sleep(8);
    }
#endif
    setup(tile);
    if (guarded) soft_wait(g.p[pi].dep, m0 >> 8, g.p[pi].dep_need, g.p[pi].err);
    stage(0, 0); stage(1, 1); stage(2, 2);
    for (int round = 0;; ++round) {
        f32x4 acc[8][TJ];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned long long* const st = stamps ? stamps + ((size_t)blockIdx.x * 8 + (size_t)round) * 4 : nullptr;
        const bool stamping = st != nullptr && tid == 0 && round < 8;
        if (stamping) st[0] = __builtin_amdgcn_s_memrealtime();
        VK_WAIT_DMA();
        VK_SYNC();
        if (wr == 1) VK_SYNC();        // the upper half of the workgroup runs half a phase behind
        int rd = 0, wrs = 3;
        for (int p = 0; p < np; ++p) {
            const uint32_t sa = lds0 + (uint32_t)rd * 2u * HT, sb = sa + HT;
            bf16x8 a[8], b[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = BT ? frag_cols<512>(sb, wc * WN + j * 16, 0, lane) : frag_strip(sb, wc * WN + j * 16, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = AT ? frag_cols<512>(sa, wr * 128 + i * 16, 0, lane) : frag_strip(sa, wr * 128 + i * 16, lane);
            stage(p + 3, wrs);
            VK_WAIT_DMA();
            VK_SYNC();
            VK_WAIT_LDS();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            VK_SYNC();
            rd = rd == RING - 1 ? 0 : rd + 1;
            wrs = wrs == RING - 1 ? 0 : wrs + 1;
        }
        if (wr == 0) VK_SYNC();
        if (stamping) st[1] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // retire the zero-fill stages issued past the end of K
        VK_SYNC();                                            // ... of every wave: the whole ring is free

        const int cpi = pi, cm0 = m0, cn0 = n0;
        const int nt = tiles.next();
        const bool more = nt >= 0;
        bool staged = false;
        if (more) {                                           // next tile's operands first, then this tile's output
            setup(nt);
            if (!guarded) { stage(0, 0); stage(1, 1); stage(2, 2); staged = true; }      // a guarded tile asks for its rows behind its poll, below
        }
        if (stamping) st[2] = __builtin_amdgcn_s_memrealtime();
        f32x4 accb[8];
        gemm_epilogue<AT, EPI, 8, TJ, (int)EPI_REGION>(g.p[cpi], acc, accb, false, cm0 + wr * 128, cn0 + wc * WN, g.p[cpi].M, lane,
                                                       lds0 + EPI_BASE + (uint32_t)wave * EPI_REGION);
        if (g.p[cpi].sig) soft_signal(g.p[cpi].sig, cm0 >> 8);
        if (stamping) st[3] = __builtin_amdgcn_s_memrealtime();
        if (!more) break;
        if (!staged) {
            soft_wait(g.p[pi].dep, m0 >> 8, g.p[pi].dep_need, g.p[pi].err);
            stage(0, 0); stage(1, 1); stage(2, 2);
        }
    }
}

template <bool AT, bool BT, int EPI, int TJ>
__global__ __launch_bounds__(512) void gemm256p_kernel(const KGroup g, const int total, unsigned long long* const stamps) {
    persistent_walk<AT, BT, EPI, TJ>(g, StrideTiles{(int)blockIdx.x, total}, stamps);
    retire_mark(g);
}

// ---------------------------------------------------------------------------------------------------------
// Chain launch (vk_gemm_chain): a producer group (256 x 256 tiles, epilogue EPI_P) and the consumer group whose A operands are the
// producers' outputs (256 x 192 tiles, epilogue EPI_C) in ONE persistent launch.  Every workgroup walks its producer tiles, signalling
// row blocks (soft_signal), then its consumer tiles, each behind the poll of its row block (soft_wait): no launch boundary between the
// two GEMMs, and the workgroups that run out of producer tiles early start on consumer tiles while the others finish.  Deadlock-free
// under any residency: a waiting workgroup has no producer tile left, and a producer tile never waits.
template <bool BT, int EPI_P, int EPI_C>
__global__ __launch_bounds__(512) void gemm256c_kernel(const KGroup g, const int nprod, const int ncons, unsigned long long* const stamps) {
    persistent_walk<false, BT, EPI_P, 4>(g, ProducerTiles{(int)blockIdx.x, nprod}, stamps);
    persistent_walk<false, BT, EPI_C, 3>(g, ConsumerTiles{(int)(blockIdx.x >> 3), nprod, ncons}, stamps ? stamps + 16 : nullptr);
    retire_mark(g);
}

#ifdef VK_STUDY
static unsigned long long* g_stamps = nullptr;          // tools/stamp_gemm.py: where the persistent kernel writes its phase stamps
#else
static constexpr unsigned long long* g_stamps = nullptr;
#endif

// CUs the persistent launches leave unclaimed (vk_gemm_reserve_cus): room for a collective's channel kernels beside the backward pass
static std::atomic<int> g_reserved_cus{0};

// Workgroups of a persistent launch: as few as walk the tile list in the same number of rounds, in whole multiples of 8 (a workgroup stays on its
// XCD's chunk).  684 tiles on 256 CUs are 2.67 rounds = three tile times with 84 CUs idle in the last one; 232 workgroups x 3 tiles take the same three
// tile times, finish together, and leave 24 CUs to whatever else is running for the WHOLE launch (the weight-gradient stream, LayerNorm rows, the
// optimizer under the next forward).  Measured: forward + backward 15.74 -> 15.56 ms (profiles/r04_experiments.md 10).
static int persistent_grid(int total) {
    const int ncu = NUM_CU - g_reserved_cus.load(std::memory_order_relaxed);
    if (total <= ncu) return total;
    const int rounds = (total + ncu - 1) / ncu;
    const int even = ((total + rounds - 1) / rounds + 7) & ~7;
    return even < ncu ? even : ncu;
}

template <bool AT, bool BT, int KSPLIT>      // 4 / 3 / 2: K-split kernel with 256 / 192 / 128 columns; 0: 4-phase 256 x 256 (study builds)
static int launch_layout(int epi, const KGroup& g, int total, hipStream_t s, bool persistent, bool soft) {
    // soft: VK_GEMM_SOFT_START -- the dispatch packet goes out without the barrier bit (hipExtAnyOrderLaunch): workgroups start as CUs come
    // free while earlier launches of the stream still run; what they read is guarded by the problems' `dep` counters
    const int flags = soft ? hipExtAnyOrderLaunch : 0;
    constexpr int LDS = (KSPLIT ? 10 : 8) * HT;
#ifdef VK_STUDY
#define VK_KERNEL_OF(E) (KSPLIT == 2 ? gemm256k_kernel<AT, BT, E, 2> : KSPLIT == 3 ? gemm256k_kernel<AT, BT, E, 3> : KSPLIT == 4 ? gemm256k_kernel<AT, BT, E, 4> : gemm256_kernel<AT, BT, E>)
#else
#define VK_KERNEL_OF(E) (KSPLIT == 2 ? gemm256k_kernel<AT, BT, E, 2> : KSPLIT == 3 ? gemm256k_kernel<AT, BT, E, 3> : gemm256k_kernel<AT, BT, E, 4>)
#endif
#define VK_CASE(E)                                                                                        \
    case E: {                                                                                             \
        if constexpr (!(AT && BT) && (KSPLIT == 3 || KSPLIT == 4)) {                                      \
            if (persistent) {                                                                             \
                auto kp = gemm256p_kernel<AT, BT, E, KSPLIT>;                                             \
                static const hipError_t attr_p = hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); (void)attr_p; \
                hipExtLaunchKernelGGL(kp, dim3(persistent_grid(total)), dim3(512), LDS, s, nullptr, nullptr, flags, g, total, g_stamps); \
                break;                                                                                    \
            }                                                                                             \
        }                                                                                                 \
        auto k = VK_KERNEL_OF(E);                                                                         \
        static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); (void)attr; /* once per process, thread-safe */ \
        hipExtLaunchKernelGGL(k, dim3(total), dim3(512), LDS, s, nullptr, nullptr, flags, g);             \
        break;                                                                                            \
    }
    switch (epi) {
        VK_CASE(VK_EPI_BF16) VK_CASE(VK_EPI_GELU) VK_CASE(VK_EPI_MULR) VK_CASE(VK_EPI_ADDR) VK_CASE(VK_EPI_F32) VK_CASE(VK_EPI_RELU) VK_CASE(VK_EPI_F32_ACC)
        default: return set_error("vk_gemm_grouped: unknown epilogue %d", epi);
    }
#undef VK_CASE
#undef VK_KERNEL_OF
    return check_launch("vk_gemm_grouped");
}

int launch_gemm256_chain(int layout, int epi_p, int epi_c, const KGroup& g, int nprod, int ncons, hipStream_t s) {
    constexpr int LDS = 10 * HT;
    const int ncu = (NUM_CU - g_reserved_cus.load(std::memory_order_relaxed)) & ~7;       // whole multiples of 8: the consumer deal assumes equal shares per XCD
#define VK_CHAIN_CASE(BT_, P_, C_)                                                                                  \
    if (layout == ((BT_) ? VK_NN : VK_NT) && epi_p == (P_) && epi_c == (C_)) {                                        \
        auto k = gemm256c_kernel<BT_, P_, C_>;                                                                        \
        static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); (void)attr; \
        hipLaunchKernelGGL(k, dim3(ncu), dim3(512), LDS, s, g, nprod, ncons, g_stamps);                               \
        return check_launch("vk_gemm_chain");                                                                         \
    }
    VK_CHAIN_CASE(false, VK_EPI_GELU, VK_EPI_BF16)          // FFN-up + GELU -> FFN-down          (encoders.py:486-501 -> 541-566)
    VK_CHAIN_CASE(true, VK_EPI_MULR, VK_EPI_ADDR)           // FFN-down dgrad x gelu' -> FFN-up dgrad + residual gradient
#undef VK_CHAIN_CASE
    return set_error("vk_gemm_chain: layout %d with epilogues %d -> %d is not built (NT: GELU -> BF16, NN: MULR -> ADDR)", layout, epi_p, epi_c);
}

template <int KSPLIT>
static int launch_variant(int layout, int epilogue, const KGroup& g, int total, hipStream_t s, bool persistent, bool soft) {
    if (layout == VK_NT) return launch_layout<false, false, KSPLIT>(epilogue, g, total, s, persistent, soft);
    if (layout == VK_NN) return launch_layout<false, true, KSPLIT>(epilogue, g, total, s, persistent, soft);
    if (layout == VK_TN) return launch_layout<true, true, KSPLIT>(epilogue, g, total, s, false, false);
    return set_error("vk_gemm_grouped: unknown layout %d", layout);
}

int launch_gemm256(int layout, int epilogue, const KGroup& g, int total, hipStream_t s, int variant, bool persistent, bool soft) {
    if (variant == 2) return launch_variant<2>(layout, epilogue, g, total, s, false, soft);
    if (variant == 3) return launch_variant<3>(layout, epilogue, g, total, s, persistent, soft);
    if (variant == 4) return launch_variant<4>(layout, epilogue, g, total, s, persistent, soft);
#ifdef VK_STUDY
    return launch_variant<0>(layout, epilogue, g, total, s, false, false);
#else
    return set_error("vk_gemm_grouped: geometry variant %d is not part of this build", variant);
#endif
}

}  // namespace vk

extern "C" int vk_gemm_reserve_cus(int n) {
    const int prev = vk::g_reserved_cus.load(std::memory_order_relaxed);
    if (n >= 0) vk::g_reserved_cus.store(n > 128 ? 128 : (n + 7) & ~7, std::memory_order_relaxed);     // whole multiples of 8: the walk keeps a workgroup on one XCD's chunk
    return prev;
}

#ifdef VK_STUDY
extern "C" void vk_gemm_set_stamp_buffer(void* p) { vk::g_stamps = (unsigned long long*)p; }
extern "C" int vk_gemm_set_kstamp_buffer(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(vk::g_kstamps), &p, sizeof(p)); }
extern "C" int vk_gemm_set_desync(int ticks) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(vk::g_desync_ticks), &ticks, sizeof(ticks)); }
#endif
