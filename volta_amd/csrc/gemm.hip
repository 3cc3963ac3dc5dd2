// Grouped bf16 GEMM for gfx950: 128x128x64 tiles, 4 waves (2x2) x 64x64 wave tiles of
// v_mfma_f32_16x16x32_bf16, operands staged HBM -> LDS by bounds-checked LDS-DMA
// (buffer_load ... lds, 16 B/lane), double-buffered.  Three operand layouts share one skeleton:
//   NT  both operands K-contiguous            -> fragments by ds_read_b128 (XOR-swizzled image)
//   NN  B stored [K][N]                       -> B fragments by ds_read_b64_tr_b16 (hardware transpose)
//   TN  A stored [K][M], B stored [K][N]      -> both by ds_read_b64_tr_b16
// The swizzle lives on the per-lane GLOBAL source address (LDS-DMA writes LDS linearly) and on the
// read address (cdna guide rule 21).  MFMA operands are swapped (D = B-frag x A-frag) so that every
// lane owns 4 consecutive output columns of one output row: 8-byte bf16 / 16-byte fp32 stores.
//
// Replaces every nn.Linear forward/backward of volta/encoders.py and volta/embeddings.py
// (see include/volta_hip.h for the site list).
#include "common.h"
#include "../../include/volta_hip.h"
#include "util.h"

namespace vk {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;
constexpr int TILE_BYTES = 128 * 64 * 2;          // one operand tile, either orientation
constexpr int STAGE_BYTES = 2 * TILE_BYTES;
constexpr int GEMM_LDS = 2 * STAGE_BYTES;         // 64 KiB -> 2 workgroups / CU

struct KProb {
    const char* A; const char* B; char* C; char* C2; const float* bias; const char* R; float* bias_grad;
    const int32_t* dyn;
    int32_t M, N, K, lda, ldb, ldc, ldr, n_store;
    int32_t tiles_n, tile_start;
};
struct KGroup {
    int32_t nprob;
    KProb p[VK_GEMM_MAX_GROUP];
};

// XOR applied to the 16-byte chunk index of a row of the transposed ([k][128 cols], 256-B rows) image
__device__ __forceinline__ int tswz(int row) { return ((row & 3) << 1) ^ (((row >> 3) & 1) << 3); }

// Stage one 16 KiB operand tile.  T=false: image [128 rows][64 k] (128-B rows), element (row0+r, col0+c).
// T=true: image [64 k][128 cols] (256-B rows), element (row0+r, col0+c) with row0 = k0.
template <bool T>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rs, uint32_t lds_tile, int ld, int row0, int col0,
                                           int tid) {
    const int wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r, cl;
        if (!T) {
            r = i * 32 + (tid >> 3);
            cl = (tid & 7) ^ (r & 7);
        } else {
            r = i * 16 + (tid >> 4);
            cl = (tid & 15) ^ tswz(r);
        }
        uint32_t voff = ((uint32_t)(row0 + r) * (uint32_t)ld + (uint32_t)(col0 + cl * 8)) * 2u;
        uint32_t dst = lds_tile + i * 4096 + wave * 1024;   // wave-uniform; hardware adds lane*16
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (VK_LDS void*)(uintptr_t)dst, 16, voff, 0, 0, 0);
    }
}

// fragment for the 16 rows [r0, r0+16) of a K-contiguous image, k-substep ks (32 wide)
__device__ __forceinline__ bf16x8 frag_rows(uint32_t tile, int r0, int ks, int lane) {
    const int r = r0 + (lane & 15);
    const int c = (ks * 4 + (lane >> 4)) ^ (r & 7);
    return *(const bf16x8 VK_LDS*)(uintptr_t)(tile + r * 128 + c * 16);
}
// fragment for the 16 columns [c0, c0+16) of a transposed image ([k][128]); k order = natural
__device__ __forceinline__ bf16x8 frag_cols(uint32_t tile, int c0, int ks, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row_a = ks * 32 + g * 8 + q;
    const int chunk = (c0 >> 3) + (p >> 1);
    const uint32_t a0 = tile + row_a * 256 + ((chunk ^ tswz(row_a)) << 4) + ((p & 1) << 3);
    const int row_b = row_a + 4;
    const uint32_t a1 = tile + row_b * 256 + ((chunk ^ tswz(row_b)) << 4) + ((p & 1) << 3);
    bf16x4 lo = lds_read_tr16(a0);
    bf16x4 hi = lds_read_tr16(a1);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(const KGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(VK_LDS char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int pi = 0;
#pragma unroll
    for (int i = 1; i < VK_GEMM_MAX_GROUP; ++i)
        if (i < g.nprob && (int)blockIdx.x >= g.p[i].tile_start) pi = i;
    const KProb& P = g.p[pi];
    const int t = blockIdx.x - P.tile_start;
    const int tm = t / P.tiles_n, tn = t - tm * P.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    int M = P.M, K = P.K;
    if (P.dyn) {
        const int d = *P.dyn;
        if (AT) K = d < K ? d : K; else M = d < M ? d : M;
    }
    if (m0 >= M) return;

    // buffer extents: last valid row + valid row length
    const int a_rows = AT ? K : M, a_cols = AT ? P.M : K;
    const int b_rows = BT ? K : P.N, b_cols = BT ? P.N : K;
    const __amdgpu_buffer_rsrc_t rsA =
        make_rsrc(P.A, a_rows > 0 ? (uint32_t)(((uint32_t)(a_rows - 1) * P.lda + (AT ? P.lda : a_cols)) * 2u) : 0u);
    const __amdgpu_buffer_rsrc_t rsB =
        make_rsrc(P.B, b_rows > 0 ? (uint32_t)(((uint32_t)(b_rows - 1) * P.ldb + (BT ? P.ldb : b_cols)) * 2u) : 0u);
    (void)a_cols; (void)b_cols;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias_grad = AT && BT && (P.bias_grad != nullptr) && (tn == 0) && (wn == 0);
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (short)0x3F80;

    const int nk = (K + BK - 1) / BK;
    auto stage = [&](int buf, int kt) {
        const uint32_t ta = lds0 + buf * STAGE_BYTES, tb = ta + TILE_BYTES;
        if (AT) stage_tile<true>(rsA, ta, P.lda, kt * BK, m0, tid);
        else    stage_tile<false>(rsA, ta, P.lda, m0, kt * BK, tid);
        if (BT) stage_tile<true>(rsB, tb, P.ldb, kt * BK, n0, tid);
        else    stage_tile<false>(rsB, tb, P.ldb, n0, kt * BK, tid);
    };

    if (nk > 0) stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const uint32_t ta = lds0 + cur * STAGE_BYTES, tb = ta + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = AT ? frag_cols(ta, wm * 64 + i * 16, ks, lane) : frag_rows(ta, wm * 64 + i * 16, ks, lane);
                b[i] = BT ? frag_cols(tb, wn * 64 + i * 16, ks, lane) : frag_rows(tb, wn * 64 + i * 16, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            if (do_bias_grad) {
#pragma unroll
                for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[i], accb[i], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane owns row m = ...+(lane&15), columns n..n+3 with n = ...+4*(lane>>4) ----
    const int gq = lane >> 4, lr = lane & 15;
    const int Mout = AT ? P.M : M;     // TN: M is the output row count and is never dynamic
    const int N = P.N;
    const int nlim = ((EPI == VK_EPI_F32 || EPI == VK_EPI_F32_ACC) && P.n_store > N) ? P.n_store : N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + lr;
        if (m >= Mout) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + gq * 4;
            if (n >= nlim) continue;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (EPI != VK_EPI_MULR && P.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < N) v[r] += P.bias[n + r];
            }
            const size_t off = (size_t)m * P.ldc + n;
            const bool full = (n + 3 < nlim);
            if (EPI == VK_EPI_F32 || EPI == VK_EPI_F32_ACC) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r >= N) v[r] = 0.f;
                float* c = (float*)P.C + off;
                if (EPI == VK_EPI_F32_ACC) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < nlim) v[r] += c[r];
                }
                if (full) *(f32x4*)c = f32x4{v[0], v[1], v[2], v[3]};
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < nlim) c[r] = v[r];
                continue;
            }
            float w[4] = {0.f, 0.f, 0.f, 0.f};
            if (EPI == VK_EPI_MULR || EPI == VK_EPI_ADDR) {
                const uint16_t* rp = (const uint16_t*)P.R + (size_t)m * P.ldr + n;
                if (full) {
                    u32x2 rr = *(const u32x2*)rp;
                    w[0] = bf2f(rr[0] & 0xFFFF); w[1] = bf2f(rr[0] >> 16); w[2] = bf2f(rr[1] & 0xFFFF); w[3] = bf2f(rr[1] >> 16);
                } else
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) w[r] = bf2f(rp[r]);
            }
            float o[4], o2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (EPI == VK_EPI_BF16) o[r] = v[r];
                else if (EPI == VK_EPI_GELU) { o[r] = gelu_f(v[r]); o2[r] = gelu_grad_f(v[r]); }
                else if (EPI == VK_EPI_MULR) o[r] = v[r] * w[r];
                else if (EPI == VK_EPI_ADDR) o[r] = v[r] + w[r];
                else o[r] = fmaxf(v[r], 0.f);
            }
            uint16_t* c = (uint16_t*)P.C + off;
            if (full) {
                *(u32x2*)c = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                if (EPI == VK_EPI_GELU) *(u32x2*)((uint16_t*)P.C2 + off) = u32x2{pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3])};
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r < N) {
                    c[r] = f2bf(o[r]);
                    if (EPI == VK_EPI_GELU) ((uint16_t*)P.C2 + off)[r] = f2bf(o2[r]);
                }
            }
        }
    }
    if (do_bias_grad && gq == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + lr;
            if (m < Mout) P.bias_grad[m] = (EPI == VK_EPI_F32_ACC ? P.bias_grad[m] : 0.f) + accb[i][0];
        }
    }
}

template <bool AT, bool BT>
static int launch_layout(int epi, const KGroup& g, int total, hipStream_t s) {
#define VK_CASE(E)                                                                                        \
    case E: {                                                                                             \
        auto k = gemm_kernel<AT, BT, E>;                                                                  \
        static bool once = false;                                                                         \
        if (!once) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS); once = true; } \
        hipLaunchKernelGGL(k, dim3(total), dim3(GEMM_THREADS), GEMM_LDS, s, g);                           \
        break;                                                                                            \
    }
    switch (epi) {
        VK_CASE(VK_EPI_BF16) VK_CASE(VK_EPI_GELU) VK_CASE(VK_EPI_MULR) VK_CASE(VK_EPI_ADDR) VK_CASE(VK_EPI_F32) VK_CASE(VK_EPI_RELU) VK_CASE(VK_EPI_F32_ACC)
        default: return set_error("vk_gemm_grouped: unknown epilogue %d", epi);
    }
#undef VK_CASE
    return check_launch("vk_gemm_grouped");
}

}  // namespace vk

extern "C" int vk_gemm_grouped(int layout, int epilogue, const vk_gemm_problem* probs, int nprob, vk_stream_t stream) {
    using namespace vk;
    if (nprob < 1 || nprob > VK_GEMM_MAX_GROUP) return set_error("vk_gemm_grouped: nprob %d out of range", nprob);
    KGroup g;
    g.nprob = nprob;
    int total = 0;
    for (int i = 0; i < nprob; ++i) {
        const vk_gemm_problem& q = probs[i];
        if (q.M < 0 || q.N <= 0 || q.K < 0) return set_error("vk_gemm_grouped: bad shape %d %d %d", q.M, q.N, q.K);
        if ((q.lda & 7) || (q.ldb & 7)) return set_error("vk_gemm_grouped: lda/ldb must be multiples of 8 (got %d %d)", q.lda, q.ldb);
        if (((uintptr_t)q.A & 15) || ((uintptr_t)q.B & 15) || ((uintptr_t)q.C & 15)) return set_error("vk_gemm_grouped: operands must be 16-byte aligned");
        if (epilogue != VK_EPI_F32 && epilogue != VK_EPI_F32_ACC && (q.ldc & 3)) return set_error("vk_gemm_grouped: ldc must be a multiple of 4");
        if (layout != VK_TN && (q.K % 64) != 0 && q.lda < ((q.K + 63) / 64) * 64)
            return set_error("vk_gemm_grouped: K=%d needs lda padded to a multiple of 64", q.K);
        if (layout != VK_TN && q.bias_grad) return set_error("vk_gemm_grouped: bias_grad is a TN (wgrad) feature");
        if ((epilogue == VK_EPI_MULR || epilogue == VK_EPI_ADDR) && !q.R) return set_error("vk_gemm_grouped: R missing");
        if (epilogue == VK_EPI_GELU && !q.C2) return set_error("vk_gemm_grouped: C2 missing");
        KProb& k = g.p[i];
        k.A = (const char*)q.A; k.B = (const char*)q.B; k.C = (char*)q.C; k.C2 = (char*)q.C2; k.bias = q.bias;
        k.R = (const char*)q.R; k.bias_grad = q.bias_grad; k.dyn = q.dyn;
        k.M = q.M; k.N = q.N; k.K = q.K; k.lda = q.lda; k.ldb = q.ldb; k.ldc = q.ldc; k.ldr = q.ldr; k.n_store = q.n_store;
        const int ncols = ((epilogue == VK_EPI_F32 || epilogue == VK_EPI_F32_ACC) && q.n_store > q.N) ? q.n_store : q.N;
        k.tiles_n = (ncols + BN - 1) / BN;
        k.tile_start = total;
        total += ((q.M + BM - 1) / BM) * k.tiles_n;
    }
    if (total == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    switch (layout) {
        case VK_NT: return launch_layout<false, false>(epilogue, g, total, s);
        case VK_NN: return launch_layout<false, true>(epilogue, g, total, s);
        case VK_TN: return launch_layout<true, true>(epilogue, g, total, s);
    }
    return set_error("vk_gemm_grouped: unknown layout %d", layout);
}
