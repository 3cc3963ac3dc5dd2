// Half-CU bf16 GEMM tile for gfx950: 256 x 128 x 32 per K-step, FOUR waves (2 along M x 2 along N, 128 x 64 outputs each -- the same
// per-wave work as the 8-wave 256 x 256 kernel of gemm256.hip), <= 256 VGPRs and 72 KiB of LDS, so that TWO workgroups share a CU:
// one wave of each per SIMD.  What that buys (profiles/r03_timeline_before.md):
//   * the epilogue of one workgroup (bias / GELU / residual, LDS transposition, stores: 2.7-8 us per tile with the matrix pipe idle
//     in the 8-wave kernel, 15-35 % of a K = 768 tile) runs under the K loop of its neighbour;
//   * a weight-gradient workgroup leaves half the CU's registers and LDS to the LayerNorm / attention workgroups of the main chain,
//     which otherwise queue behind 110 us workgroups that own whole CUs.
// The price is operand traffic: 24 KiB of LDS fill per 32 MFMAs per wave instead of 32 KiB per 64 (1.5 x per FLOP).
//
// LDS: ring of 3 K-steps, slot = A strip [256][32 k] (16 KiB) | B strip [128][32 k] (8 KiB); images, swizzles and fragment reads are
// those of gemm256.hip's K-split kernel.  One phase per K-step, ONE barrier per phase:
//     read fragments of K-step p | issue the DMA of K-step p + 2 into the slot of K-step p - 1 | vmcnt(6): K-step p + 1 has landed
//     (6 DMA instructions per wave and K-step; p + 2 stays in flight) | lgkmcnt(0) | 32 MFMAs | barrier
// WAR: the slot of K-step p - 1 was last read in phase p - 1, those reads retired (lgkmcnt(0)) before that phase's MFMAs and every wave
// has passed that phase's barrier.  RAW: every wave's vmcnt(6) of phase p precedes the barrier of phase p, K-step p + 1 is first read in
// phase p + 1.  The matrix pipe of a SIMD is kept busy across a wave's read / wait bubble by the co-resident workgroup's wave.
//
// The same loop with 64-row wave tiles (TI = 4: a 128 x 128 workgroup tile, 16 KiB per K-step) and a ring of 6 K-steps serves the launches
// that cannot fill the chip with 256-row tiles and are deep in K (the text-only sub-layers' [5120 x 768 x 3072]: 80 tiles of 256 x 192,
// 240 of 128 x 128): one workgroup per CU, five K-steps in flight in front of the one being read.
#include "gemm_common.h"

namespace vk {

constexpr uint32_t W4_OOB = 0x80000000u;

__device__ __forceinline__ int kswz4(int r) { return (-(r >> 2)) & 3; }

// byte offsets (K-step 0) of the 16-byte pieces a thread stages of a strip of EXT rows (K-contiguous: image [EXT][32 k], 64-byte rows,
// chunk c of row r at chunk c ^ kswz4(r)) or EXT columns (transposed: image [32 k][EXT], 2 EXT-byte rows, chunk XOR tswz(k row))
template <bool T, int EXT, int NMAX>
__device__ __forceinline__ void strip_offsets4(uint32_t (&off)[NMAX], int ld, int ext0, int tid) {
    static_assert(EXT / 64 <= NMAX, "offset array too small");
#pragma unroll
    for (int i = 0; i < EXT / 64; ++i) {
        const int lin = i * 256 + tid;
        if (!T) {
            const int r = lin >> 2, cp = lin & 3;
            off[i] = ((uint32_t)(ext0 + r) * (uint32_t)ld + (uint32_t)((cp ^ kswz4(r)) * 8)) * 2u;
        } else {
            constexpr int CPR = EXT / 8;
            const int kr = lin / CPR, cp = lin % CPR;
            const int c = cp ^ tswz(kr);
            off[i] = ((uint32_t)kr * (uint32_t)ld + (uint32_t)(ext0 + c * 8)) * 2u;
        }
    }
}

__device__ __forceinline__ bf16x8 frag_strip4(uint32_t strip, int r0, int lane) {
    const int r = r0 + (lane & 15);
    return *(const bf16x8 VK_LDS*)(uintptr_t)(strip + r * 64 + (((lane >> 4) ^ kswz4(r)) << 4));
}

#define W4_BARRIER()                              \
    do {                                          \
        __builtin_amdgcn_sched_barrier(0);        \
        __builtin_amdgcn_s_barrier();             \
        __builtin_amdgcn_sched_barrier(0);        \
    } while (0)

// TI: 16-row tiles per wave along M (8: 256-row workgroup tile, 4: 128); RING: K-steps of LDS.  (The derived constants live in a class
// template: taken from local constexpr variables as array bounds / template arguments they made hipcc drop the kernel without a diagnostic.)
template <int TI, int RING> struct W4Geo {
    static constexpr int BM = 32 * TI, NPA = BM / 64, NPW = NPA + 2, PF = RING - 1;       // DMA pieces per wave and K-step: A, A + B; K-steps staged ahead
    static constexpr uint32_t A_BYTES = (uint32_t)BM * 64u, SLOT = A_BYTES + 8192u;
};

template <bool AT, bool BT, int EPI, int TI, int RING>
__global__ __launch_bounds__(256, TI == 8 ? 2 : 1) void gemm4w_kernel(const KGroup g) {
    constexpr bool BG = AT && BT;
    using G = W4Geo<TI, RING>;
    constexpr int BM = G::BM, NPA = G::NPA, NPW = G::NPW, PF = G::PF;
    constexpr uint32_t A_BYTES = G::A_BYTES, SLOT = G::SLOT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(VK_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int bid = (g.stagger & GROUP_PLAIN_ORDER) ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma nounroll
    for (int i = 1; i < g.nprob; ++i)
        if (bid >= g.p[i].tile_start) pi = i;
    const KProb& P = g.p[pi];
    const int t = bid - P.tile_start;
    const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
    const int m0 = tm * BM, n0 = tn * 128;

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

    uint32_t offA[4], offB[2];            // (fixed bounds: an array whose bound depends on a template parameter, captured by the staging lambda, made hipcc drop the kernel without a diagnostic)
    strip_offsets4<AT, G::BM, 4>(offA, P.lda, m0, tid);
    strip_offsets4<BT, 128, 2>(offB, P.ldb, n0, tid);
    // (columns of a transposed strip beyond the operand's extent read whatever follows in memory: they only reach output rows / columns
    // that the epilogue never stores; reads past the end of the buffer return zero)
    const uint32_t kA = AT ? 64u * (uint32_t)P.lda : 64u, kB = BT ? 64u * (uint32_t)P.ldb : 64u;     // bytes per 32-deep K-step
    const int np = (K + 31) / 32;
    auto stage = [&](int p, int slot) {
        const bool live = p < np;
        const uint32_t sa = lds0 + (uint32_t)slot * SLOT;
        const uint32_t addA = live ? (uint32_t)p * kA : W4_OOB, addB = live ? (uint32_t)p * kB : W4_OOB;
#pragma unroll
        for (int i = 0; i < NPA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (VK_LDS void*)(uintptr_t)(sa + (uint32_t)(i * 256 + wave * 64) * 16u), 16, offA[i] + addA, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (VK_LDS void*)(uintptr_t)(sa + A_BYTES + (uint32_t)(i * 256 + wave * 64) * 16u), 16, offB[i] + addB, 0, 0, 0);
    };

    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias_grad = BG && (P.bias_grad != nullptr) && (tn == 0) && (wc == 0);
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (short)0x3F80;

    // vmcnt(NPW * (PF - 1)): of the PF K-steps staged ahead all but the oldest may still be in flight
    static_assert(NPW * (PF - 1) == 6 || NPW * (PF - 1) == 16, "add the wait count of this geometry");
#define W4_WAIT_STAGE()                                                               \
    do {                                                                              \
        if constexpr (NPW * (PF - 1) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  \
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                        \
    } while (0)
#pragma unroll
    for (int i = 0; i < PF; ++i) stage(i, i);
    W4_WAIT_STAGE();
    W4_BARRIER();
    int rd = 0, wrs = PF;
    for (int p = 0; p < np; ++p) {
        const uint32_t sa = lds0 + (uint32_t)rd * SLOT, sb = sa + A_BYTES;
        bf16x8 a[TI], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = BT ? frag_cols<256>(sb, wc * 64 + j * 16, 0, lane) : frag_strip4(sb, wc * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < TI; ++i) a[i] = AT ? frag_cols<2 * G::BM>(sa, wr * (BM / 2) + i * 16, 0, lane) : frag_strip4(sa, wr * (BM / 2) + i * 16, lane);
        stage(p + PF, wrs);
        W4_WAIT_STAGE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        if (BG && do_bias_grad) {
#pragma unroll
            for (int i = 0; i < TI; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[i], accb[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        W4_BARRIER();
        rd = rd == RING - 1 ? 0 : rd + 1;
        wrs = wrs == RING - 1 ? 0 : wrs + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // retire the zero-fill stages issued past the end of K
    W4_BARRIER();                                          // ... of every wave: the ring is free for the epilogue's transposition

#undef W4_WAIT_STAGE
    gemm_epilogue<AT, EPI, TI, 4>(P, acc, accb, do_bias_grad, m0 + wr * (BM / 2), n0 + wc * 64, M, lane, lds0 + (uint32_t)wave * 16384u);
    retire_mark(g);
}

template <bool AT, bool BT, int TI, int RING>
static int launch_layout4(int epi, const KGroup& g, int total, hipStream_t s) {
    constexpr int LDS = RING * (32 * TI * 64 + 8192);
    static_assert(LDS >= 4 * 16384, "the epilogue stages 16 KiB per wave in the ring");
#define VK_CASE(E)                                                                                        \
    case E: {                                                                                             \
        auto k = gemm4w_kernel<AT, BT, E, TI, RING>;                                                      \
        static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); (void)attr; \
        hipLaunchKernelGGL(k, dim3(total), dim3(256), LDS, s, g);                                         \
        break;                                                                                            \
    }
    switch (epi) {
        VK_CASE(VK_EPI_BF16) VK_CASE(VK_EPI_GELU) VK_CASE(VK_EPI_MULR) VK_CASE(VK_EPI_ADDR) VK_CASE(VK_EPI_F32) VK_CASE(VK_EPI_RELU) VK_CASE(VK_EPI_F32_ACC)
        default: return set_error("vk_gemm_grouped: unknown epilogue %d", epi);
    }
#undef VK_CASE
    return check_launch("vk_gemm_grouped");
}

int launch_gemm4w(int layout, int epilogue, const KGroup& g, int total, hipStream_t s, int bm) {
    if (bm == 128) {         // 128 x 128 tiles, ring of 6: one workgroup per CU with five K-steps in flight
        if (layout == VK_NT) return launch_layout4<false, false, 4, 6>(epilogue, g, total, s);
        if (layout == VK_NN) return launch_layout4<false, true, 4, 6>(epilogue, g, total, s);
        if (layout == VK_TN) return launch_layout4<true, true, 4, 6>(epilogue, g, total, s);
    } else {                 // 256 x 128 tiles, ring of 3: two workgroups per CU
        if (layout == VK_NT) return launch_layout4<false, false, 8, 3>(epilogue, g, total, s);
        if (layout == VK_NN) return launch_layout4<false, true, 8, 3>(epilogue, g, total, s);
        if (layout == VK_TN) return launch_layout4<true, true, 8, 3>(epilogue, g, total, s);
    }
    return set_error("vk_gemm_grouped: unknown layout %d", layout);
}

}  // namespace vk
