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
#include "gemm_common.h"

namespace vk {

constexpr uint32_t W4_SLOT = 24576, W4_A = 16384;
constexpr uint32_t W4_OOB = 0x80000000u;

__device__ __forceinline__ int kswz4(int r) { return (-(r >> 2)) & 3; }

// byte offsets (K-step 0) of the 16-byte pieces a thread stages of a strip of EXT rows (K-contiguous: image [EXT][32 k], 64-byte rows,
// chunk c of row r at chunk c ^ kswz4(r)) or EXT columns (transposed: image [32 k][EXT], 2 EXT-byte rows, chunk XOR tswz(k row))
template <bool T, int EXT>
__device__ __forceinline__ void strip_offsets4(uint32_t (&off)[EXT / 64], int ld, int ext0, int tid) {
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

template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm4w_kernel(const KGroup g) {
    constexpr bool BG = AT && BT;
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
    const int m0 = tm * 256, n0 = tn * 128;

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

    uint32_t offA[4], offB[2];
    strip_offsets4<AT, 256>(offA, P.lda, m0, tid);
    strip_offsets4<BT, 128>(offB, P.ldb, n0, tid);
    // (columns of a transposed strip beyond the operand's extent read whatever follows in memory: they only reach output rows / columns
    // that the epilogue never stores; reads past the end of the buffer return zero)
    const uint32_t kA = AT ? 64u * (uint32_t)P.lda : 64u, kB = BT ? 64u * (uint32_t)P.ldb : 64u;     // bytes per 32-deep K-step
    const int np = (K + 31) / 32;
    auto stage = [&](int p, int slot) {
        const bool live = p < np;
        const uint32_t sa = lds0 + (uint32_t)slot * W4_SLOT;
        const uint32_t addA = live ? (uint32_t)p * kA : W4_OOB, addB = live ? (uint32_t)p * kB : W4_OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (VK_LDS void*)(uintptr_t)(sa + (uint32_t)(i * 256 + wave * 64) * 16u), 16, offA[i] + addA, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (VK_LDS void*)(uintptr_t)(sa + W4_A + (uint32_t)(i * 256 + wave * 64) * 16u), 16, offB[i] + addB, 0, 0, 0);
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

    stage(0, 0); stage(1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    W4_BARRIER();
    int rd = 0, wrs = 2;
    for (int p = 0; p < np; ++p) {
        const uint32_t sa = lds0 + (uint32_t)rd * W4_SLOT, sb = sa + W4_A;
        bf16x8 a[8], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = BT ? frag_cols<256>(sb, wc * 64 + j * 16, 0, lane) : frag_strip4(sb, wc * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = AT ? frag_cols<512>(sa, wr * 128 + i * 16, 0, lane) : frag_strip4(sa, wr * 128 + i * 16, lane);
        stage(p + 2, wrs);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        if (BG && do_bias_grad) {
#pragma unroll
            for (int i = 0; i < 8; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[i], accb[i], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        W4_BARRIER();
        rd = rd == 2 ? 0 : rd + 1;
        wrs = wrs == 2 ? 0 : wrs + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // retire the zero-fill stages issued past the end of K
    W4_BARRIER();                                          // ... of every wave: the ring is free for the epilogue's transposition

    gemm_epilogue<AT, EPI, 8, 4>(P, acc, accb, do_bias_grad, m0 + wr * 128, n0 + wc * 64, M, lane, lds0 + (uint32_t)wave * 16384u);
}

template <bool AT, bool BT>
static int launch_layout4(int epi, const KGroup& g, int total, hipStream_t s) {
    constexpr int LDS = 3 * W4_SLOT;
#define VK_CASE(E)                                                                                        \
    case E: {                                                                                             \
        auto k = gemm4w_kernel<AT, BT, E>;                                                                \
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

int launch_gemm4w(int layout, int epilogue, const KGroup& g, int total, hipStream_t s) {
    if (layout == VK_NT) return launch_layout4<false, false>(epilogue, g, total, s);
    if (layout == VK_NN) return launch_layout4<false, true>(epilogue, g, total, s);
    if (layout == VK_TN) return launch_layout4<true, true>(epilogue, g, total, s);
    return set_error("vk_gemm_grouped: unknown layout %d", layout);
}

}  // namespace vk
