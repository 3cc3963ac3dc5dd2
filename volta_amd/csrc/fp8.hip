// fp8 (OCP e4m3) projection path for gfx950: the forward GEMMs y = x W^T of the encoder sub-layers on the block-scaled MFMA
// v_mfma_scale_f32_16x16x128_f8f6f4 (twice the bf16 MFMA rate per byte of operand, half the bytes), fp32 accumulation.
// BASELINE.json configs[4] (ctrl_vl-bert_base, 100 regions) names this path; the reference itself is fp32 throughout, the sites are
// the nn.Linear forwards of volta/encoders.py:242-255 (Q|K|V), :495-499 (FFN up) and :552-565 (FFN down).
//   * operands: x quantised PER ROW (x[m, :] ~ q[m, :] * sa[m], sa = amax / 448), W quantised per output channel (row of W[N, K]);
//     the MFMA's own block scales stay at 1.0 (E8M0 127) and the epilogue multiplies the fp32 accumulator by sa[m] * sb[n]: exact
//     de-quantisation, so the GEMM is tested against an fp32 matmul of the de-quantised operands;
//   * layout NT only (both operands K-contiguous): gradients stay on the bf16 kernels (straight-through estimator: the backward
//     differentiates the un-quantised operands, which the engine keeps in bf16 anyway);
//   * kernel: 256 x 256 (8 waves, 128 x 64 per wave) or 128 x 128 (4 waves) tile, 128-deep K-steps (128-byte LDS rows, the same XOR-
//     swizzled image as the bf16 kernels: bytes are bytes), LDS-DMA double buffer, one barrier per K-step, then the bf16 kernels'
//     epilogue (gemm_common.h) after the de-quantisation multiply.
#include "gemm_common.h"

namespace vk {

typedef __attribute__((ext_vector_type(8))) int i32x8;
constexpr float FP8_MAX = 448.0f;          // largest finite e4m3fn

struct KProb8 { KProb b; const float* sa; const float* sb; };
struct KGroup8 { int32_t nprob; int32_t plain_order; KProb8 p[VK_GEMM_FP8_MAX_GROUP]; };

// ---- quantisation ---------------------------------------------------------------------------------------------------------------
// one wave per row: amax, scale = amax / 448 (1 for an all-zero row), q = round-to-nearest-even(x / scale); 8 elements per lane and step
template <typename SRC>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(const SRC* __restrict__ src, int64_t ld, uint8_t* __restrict__ dst, int64_t ldq, float* __restrict__ scale,
                                                             int M, int K, const int32_t* dyn) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int Mr = dyn ? min(*dyn, M) : M;
    if (row >= Mr) return;
    const SRC* x = src + (size_t)row * ld;
    constexpr int MAXCH = 8;                 // K <= 64 * 8 * 8 = 4096
    float v[MAXCH][8];
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int c = (j * 64 + lane) * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[j][r] = 0.f;
        if (c < K) {
            if constexpr (sizeof(SRC) == 2) {
                const u32x4 w = *(const u32x4*)((const uint16_t*)x + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[j][2 * r] = bf2f(w[r] & 0xFFFF); v[j][2 * r + 1] = bf2f(w[r] >> 16); }
            } else {
                const f32x4 a = *(const f32x4*)((const float*)x + c), b = *(const f32x4*)((const float*)x + c + 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[j][r] = a[r]; v[j][4 + r] = b[r]; }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) amax = fmaxf(amax, fabsf(v[j][r]));
        }
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax / FP8_MAX : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) scale[row] = s;
    uint8_t* q = dst + (size_t)row * ldq;
#pragma unroll
    for (int j = 0; j < MAXCH; ++j) {
        const int c = (j * 64 + lane) * 8;
        if (c < K) *(u32x2*)(q + c) = u32x2{pack4_fp8(v[j][0] * inv, v[j][1] * inv, v[j][2] * inv, v[j][3] * inv),
                                              pack4_fp8(v[j][4] * inv, v[j][5] * inv, v[j][6] * inv, v[j][7] * inv)};
    }
}

// elementwise bf16 -> fp8 with one static scale (q = x * mul): the GELU output feeding the FFN-down projection
__global__ void cast_bf16_fp8_kernel(const uint16_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n, float mul) {
    const size_t n8 = n >> 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 w = *(const u32x4*)(src + i * 8);
        *(u32x2*)(dst + i * 8) = u32x2{pack4_fp8(bf2f(w[0] & 0xFFFF) * mul, bf2f(w[0] >> 16) * mul, bf2f(w[1] & 0xFFFF) * mul, bf2f(w[1] >> 16) * mul),
                                       pack4_fp8(bf2f(w[2] & 0xFFFF) * mul, bf2f(w[2] >> 16) * mul, bf2f(w[3] & 0xFFFF) * mul, bf2f(w[3] >> 16) * mul)};
    }
}

// ---- GEMM -------------------------------------------------------------------------------------------------------------------------
// 16 rows [r0, r0 + 16) x 128 k of a K-contiguous fp8 image (128-byte rows, 16-byte chunk c of row r stored at c ^ (r & 7)):
// lane group g = lane >> 4 supplies k = 32 g .. 32 g + 31 of row r0 + (lane & 15) -- two ds_read_b128
__device__ __forceinline__ i32x8 frag8(uint32_t tile, int r0, int lane) {
    const int r = r0 + (lane & 15), g = lane >> 4;
    const u32x4 lo = *(const u32x4 VK_LDS*)(uintptr_t)(tile + r * 128 + (((2 * g) ^ (r & 7)) << 4));
    const u32x4 hi = *(const u32x4 VK_LDS*)(uintptr_t)(tile + r * 128 + (((2 * g + 1) ^ (r & 7)) << 4));
    i32x8 o;
    o[0] = (int)lo[0]; o[1] = (int)lo[1]; o[2] = (int)lo[2]; o[3] = (int)lo[3];
    o[4] = (int)hi[0]; o[5] = (int)hi[1]; o[6] = (int)hi[2]; o[7] = (int)hi[3];
    return o;
}

template <int EXT, int THREADS>       // EXT rows x 128 bytes of one K-step, linear LDS image, swizzle on the global source
__device__ __forceinline__ void stage8(__amdgpu_buffer_rsrc_t rs, uint32_t lds_tile, int ld, int row0, int k0, int tid) {
    constexpr int NP = EXT * 8 / THREADS;
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int lin = i * THREADS + tid;
        const int r = lin >> 3, cp = lin & 7;
        const uint32_t voff = (uint32_t)(row0 + r) * (uint32_t)ld + (uint32_t)(k0 + ((cp ^ (r & 7)) << 4));
        const uint32_t dst = lds_tile + (i * THREADS + wave * 64) * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (VK_LDS void*)(uintptr_t)dst, 16, voff, 0, 0, 0);
    }
}

template <int EPI, int WM, int WN, int TM>
__global__ __launch_bounds__(64 * WM * WN) void gemm_fp8_kernel(const KGroup8 g) {
    constexpr int THREADS = 64 * WM * WN, BM = 64 * TM * WM, BN = 64 * WN, TI = 4 * TM;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(VK_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int bid = g.plain_order ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < VK_GEMM_FP8_MAX_GROUP; ++i)
        if (i < g.nprob && bid >= g.p[i].b.tile_start) pi = i;
    const KProb8& Q = g.p[pi];
    const KProb& P = Q.b;
    const int t = bid - P.tile_start;
    const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    int M = P.M;
    const int K = P.K;
    if (P.dyn) { const int d = *P.dyn; M = d < M ? d : M; }
    if (m0 >= M) return;
    // extents in BYTES (one byte per element): last valid row + valid row length, everything beyond reads as zero
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(P.A, M > 0 ? (uint32_t)((uint32_t)(M - 1) * P.lda + K) : 0u);
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(P.B, P.N > 0 ? (uint32_t)((uint32_t)(P.N - 1) * P.ldb + K) : 0u);

    f32x4 acc[TI][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (K + 127) / 128;
    auto stage = [&](int buf, int kt) {
        const uint32_t ta = lds0 + buf * STAGE, tb = ta + A_BYTES;
        stage8<BM, THREADS>(rsA, ta, P.lda, m0, kt * 128, tid);
        stage8<BN, THREADS>(rsB, tb, P.ldb, n0, kt * 128, tid);
    };
    if (nk > 0) stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // K-step kt landed for every wave; everyone is done reading K-step kt - 1
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const uint32_t ta = lds0 + (kt & 1) * STAGE, tb = ta + A_BYTES;
        i32x8 b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = frag8(tb, wn * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const i32x8 a = frag8(ta, wm * 64 * TM + i * 16, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j)      // swapped operands (D = B-frag x A-frag): a lane owns 4 consecutive output columns of one row
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b[j], a, acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
    }
    // de-quantise: acc[m][n] *= sa[m] * sb[n] (absent scale vectors read as 1)
    {
        const int gq = lane >> 4, lr = lane & 15;
        const __amdgpu_buffer_rsrc_t ra = make_rsrc(Q.sa, Q.sa ? (uint32_t)M * 4u : 0u), rb = make_rsrc(Q.sb, Q.sb ? (uint32_t)P.N * 4u : 0u);
        f32x4 sb4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sb4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (uint32_t)(n0 + wn * 64 + j * 16 + gq * 4) * 4u, 0, 0));
            if (!Q.sb) sb4[j] = f32x4{1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            float sa = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (uint32_t)(m0 + wm * 64 * TM + i * 16 + lr) * 4u, 0, 0));
            if (!Q.sa) sa = 1.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] *= sa * sb4[j][r];
        }
    }
    f32x4 accb[TI];
    __builtin_amdgcn_s_barrier();      // every wave is done with the last K-step: LDS is free for the epilogue's transposition (16 KiB per wave)
    gemm_epilogue<false, EPI, TI, 4>(P, acc, accb, false, m0 + wm * 64 * TM, n0 + wn * 64, M, lane, lds0 + (uint32_t)wave * 16384u);
}

template <int WM, int WN, int TM>
static int launch8(int epi, const KGroup8& g, int total, hipStream_t s) {
    constexpr int LDS = 2 * (64 * TM * WM + 64 * WN) * 128;
#define VK_CASE(E)                                                                                        \
    case E: {                                                                                             \
        auto k = gemm_fp8_kernel<E, WM, WN, TM>;                                                          \
        static const hipError_t attr = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); (void)attr; \
        hipLaunchKernelGGL(k, dim3(total), dim3(64 * WM * WN), LDS, s, g);                                \
        break;                                                                                            \
    }
    switch (epi) {
        VK_CASE(VK_EPI_BF16) VK_CASE(VK_EPI_GELU) VK_CASE(VK_EPI_F32) VK_CASE(VK_EPI_RELU)
        default: return set_error("vk_gemm_fp8_grouped: epilogue %d is not available on the fp8 path", epi);
    }
#undef VK_CASE
    return check_launch("vk_gemm_fp8_grouped");
}

}  // namespace vk

using namespace vk;

extern "C" int vk_gemm_fp8_grouped(int epilogue, const vk_gemm_fp8_problem* probs, int nprob, int geometry, vk_stream_t stream) {
    if (nprob < 1 || nprob > VK_GEMM_FP8_MAX_GROUP) return set_error("vk_gemm_fp8_grouped: nprob %d out of range", nprob);
    const bool f32out = epilogue == VK_EPI_F32;
    bool any_dyn = false;
    for (int i = 0; i < nprob; ++i) {
        const vk_gemm_problem& q = probs[i].p;
        if (q.M < 0 || q.N <= 0 || q.K < 0) return set_error("vk_gemm_fp8_grouped: bad shape %d %d %d", q.M, q.N, q.K);
        if ((q.lda & 15) || (q.ldb & 15)) return set_error("vk_gemm_fp8_grouped: lda/ldb must be multiples of 16 bytes (got %d %d)", q.lda, q.ldb);
        if (((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15) || ((uintptr_t)q.C & 15)) return set_error("vk_gemm_fp8_grouped: operands must be 16-byte aligned");
        if (!f32out && (q.ldc & 3)) return set_error("vk_gemm_fp8_grouped: ldc must be a multiple of 4");
        if ((q.K % 128) != 0 && q.lda < ((q.K + 127) / 128) * 128) return set_error("vk_gemm_fp8_grouped: K=%d needs lda padded to a multiple of 128", q.K);
        if (q.bias_grad || q.R) return set_error("vk_gemm_fp8_grouped: forward epilogues only");
        if (epilogue == VK_EPI_GELU && !q.C2) return set_error("vk_gemm_fp8_grouped: C2 missing");
        if (probs[i].c8 && (epilogue != VK_EPI_GELU || (probs[i].ldc8 & 3) || ((uintptr_t)probs[i].c8 & 3))) return set_error("vk_gemm_fp8_grouped: c8 is a GELU-epilogue output with a 4-byte aligned leading dimension");
        if ((uint64_t)q.M * q.lda >= 0x7FFFFFF0ull || (uint64_t)q.N * q.ldb >= 0x7FFFFFF0ull) return set_error("vk_gemm_fp8_grouped: operands must stay below 2 GiB");
        any_dyn |= q.dyn != nullptr;
    }
    int t256 = 0;
    for (int i = 0; i < nprob; ++i) t256 += ((probs[i].p.M + 255) / 256) * ((probs[i].p.N + 255) / 256);
    int edge = geometry ? geometry : (t256 >= 160 ? 256 : 128);
    if (edge != 128 && edge != 256) return set_error("vk_gemm_fp8_grouped: geometry %d (128 or 256)", geometry);
    KGroup8 g{};          // zero: no split accumulation, no row-block hand-off
    g.nprob = nprob;
    g.plain_order = any_dyn ? 1 : 0;
    int total = 0;
    for (int i = 0; i < nprob; ++i) {
        const vk_gemm_problem& q = probs[i].p;
        KProb& k = g.p[i].b;
        k.A = (const char*)q.A; k.B = (const char*)q.B; k.C = (char*)q.C; k.C2 = (char*)q.C2; k.bias = q.bias;
        k.R = nullptr; k.bias_grad = nullptr; k.dyn = q.dyn;
        k.M = q.M; k.N = q.N; k.K = q.K; k.lda = q.lda; k.ldb = q.ldb; k.ldc = q.ldc; k.ldr = 0; k.n_store = q.n_store;
        const int ncols = (f32out && q.n_store > q.N) ? q.n_store : q.N;
        k.tiles_n = (ncols + edge - 1) / edge;
        k.tile_start = total;
        total += ((q.M + edge - 1) / edge) * k.tiles_n;
        k.C8 = (char*)probs[i].c8; k.c8_mul = probs[i].c8_mul; k.ldc8 = probs[i].ldc8;
        g.p[i].sa = probs[i].scale_a; g.p[i].sb = probs[i].scale_b;
    }
    if (total == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    return edge == 256 ? launch8<2, 4, 2>(epilogue, g, total, s) : launch8<2, 2, 1>(epilogue, g, total, s);
}

extern "C" int vk_quant_rows_fp8(const void* src, int src_is_f32, int64_t ld, void* dst, int64_t ldq, float* scale, int M, int K, const int32_t* dyn, vk_stream_t stream) {
    if (M <= 0) return 0;
    if (K <= 0 || K > 4096 || (K & 7) || (ld & 7) || (ldq & 7)) return set_error("vk_quant_rows_fp8: K=%d (multiple of 8, <= 4096), ld / ldq multiples of 8", K);
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return set_error("vk_quant_rows_fp8: alignment");
    hipStream_t s = (hipStream_t)stream;
    if (src_is_f32) hipLaunchKernelGGL(quant_rows_fp8_kernel<float>, dim3((M + 3) / 4), dim3(256), 0, s, (const float*)src, ld, (uint8_t*)dst, ldq, scale, M, K, dyn);
    else hipLaunchKernelGGL(quant_rows_fp8_kernel<uint16_t>, dim3((M + 3) / 4), dim3(256), 0, s, (const uint16_t*)src, ld, (uint8_t*)dst, ldq, scale, M, K, dyn);
    return check_launch("vk_quant_rows_fp8");
}

extern "C" int vk_cast_bf16_fp8(const void* src, void* dst, int64_t n, float mul, vk_stream_t stream) {
    if (n <= 0) return 0;
    if ((n & 7) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 7)) return set_error("vk_cast_bf16_fp8: n %% 8 == 0 and aligned buffers required");
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(cast_bf16_fp8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src, (uint8_t*)dst, (size_t)n, mul);
    return check_launch("vk_cast_bf16_fp8");
}
