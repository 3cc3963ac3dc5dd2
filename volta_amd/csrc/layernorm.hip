// Fused (dropout +) residual + LayerNorm, forward and backward, for gfx950.
// HBM-bound row kernels: one 64-lane wave per row, 8-byte bf16x4 accesses, fp32 statistics with the
// two-pass mean / biased variance of the reference (volta/encoders.py:57-61, eps inside the sqrt).
// Replaces apex's cuApplyLayerNorm / cuComputeGradInput / cuComputePartGradGammaBeta
// (apex/csrc/layer_norm_cuda_kernel.cu:279-322,523-637), which hard-code 32-lane warps, together with
// the dropout + residual add that volta runs as separate eager ops (volta/encoders.py:410-423).
#include "common.h"
#include "../../include/volta_hip.h"
#include "util.h"

namespace vk {

constexpr float LN_EPS = 1e-12f;
constexpr int LN_THREADS = 256;
constexpr int LN_BWD_ROWS = 16;     // rows per workgroup in the backward: 4 per wave, all requested up front (32 rows in two passes with the second pass
                                    // requested as the first one's registers free up measured SLOWER, 38.6 vs ~32 us at 14592 rows: a wave's rows run
                                    // one after the other, ~1 us each, and halving the wave count halves the rows in progress)

// Philox row of `row` under the two-segment mapping of vk_ln_args.seg (see include/volta_hip.h)
__device__ __forceinline__ uint32_t drop_row(const vk_drop_rows (&seg)[2], int split, int row, uint32_t& site) {
    const int si = row >= split ? 1 : 0;
    const int r = si ? row - split : row;
    site = seg[si].site;
    if (seg[si].div <= 0) return (uint32_t)r;
    return (uint32_t)((r / seg[si].div) * seg[si].mul + (r % seg[si].div) + seg[si].off);
}

__device__ __forceinline__ void load4(const uint16_t* p, float (&v)[4]) {
    u32x2 r = *(const u32x2*)p;
    v[0] = bf2f(r[0] & 0xFFFF); v[1] = bf2f(r[0] >> 16); v[2] = bf2f(r[1] & 0xFFFF); v[3] = bf2f(r[1] >> 16);
}
__device__ __forceinline__ void store4(uint16_t* p, const float (&v)[4]) {
    *(u32x2*)p = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
}

// One launch serves up to two independent jobs (the text and the vision stream of a sub-layer): workgroups
// [0, nb0) belong to job a0, the rest to job a1.  Each job is a launch-latency-sized problem (30-60 MB), so sharing
// the launch saves one ramp and fills the chip better than two half-empty grids.
// (The job's argument block is read from the kernel-argument segment at a workgroup-uniform offset: selecting between
// two by-value structs through a reference makes hipcc spill both to scratch.)
template <typename T> struct JobPair { T job[2]; int32_t nb0; int32_t stagger, stagger_mod; };      // stagger: study builds (0 otherwise)
template <typename T>
__device__ __forceinline__ T load_job(int which) {
    static_assert(sizeof(T) % 8 == 0 && alignof(T) == 8, "argument blocks are copied in 8-byte words");
    typedef __attribute__((ext_vector_type(16))) uint32_t w16;
    typedef __attribute__((ext_vector_type(2))) uint32_t w2;
    typedef __attribute__((address_space(4))) const char* kptr;
    const kptr base = (kptr)__builtin_amdgcn_kernarg_segment_ptr() + (which ? sizeof(T) : 0);
    constexpr int N16 = sizeof(T) / 64, N2 = (sizeof(T) % 64) / 8;
    struct { w16 a[N16 ? N16 : 1]; w2 b[N2 ? N2 : 1]; } buf;       // wide scalar loads (s_load_dwordx16): one wait, not one per word
#pragma unroll
    for (int i = 0; i < N16; ++i) buf.a[i] = *(__attribute__((address_space(4))) const w16*)(base + 64 * i);
#pragma unroll
    for (int i = 0; i < N2; ++i) buf.b[i] = *(__attribute__((address_space(4))) const w2*)(base + 64 * N16 + 8 * i);
    T out;
    __builtin_memcpy(&out, &buf.a[0], 64 * N16);
    __builtin_memcpy((char*)&out + 64 * N16, &buf.b[0], 8 * N2);
    return out;
}

template <int NCH>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const JobPair<vk_ln_args> jp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb0 = jp.nb0;
    const bool second = (int)blockIdx.x >= nb0;
    const vk_ln_args a = load_job<vk_ln_args>(second);
    const int row = (second ? (int)blockIdx.x - nb0 : (int)blockIdx.x) * 4 + wave;
    const int Mrows = a.dyn ? min(*a.dyn, a.M) : a.M;
    if (row >= Mrows) return;
    const int H = a.H;
    const bool drop_on = a.drop.threshold != 0;
    const uint64_t seed = drop_on ? *a.drop.seed : 0;
    const vk_dropout dc = a.drop;
    uint32_t dsite;
    const uint32_t drow = drop_row(a.seg, a.split_row, row, dsite);
    DropCfg dcfg{dc.seed, dsite, dc.threshold, dc.scale};
    const uint16_t* d = (const uint16_t*)a.d + (size_t)row * H;
    const uint16_t* x = a.x ? (const uint16_t*)a.x + (size_t)row * H : nullptr;

    float z[NCH][4];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = j * 256 + lane * 4;
        z[j][0] = z[j][1] = z[j][2] = z[j][3] = 0.f;
        if (c < H) {
            load4(d + c, z[j]);
            if (drop_on && !a.post) {
                u32x4 w = drop_words(dcfg, seed, drow, (uint32_t)(c >> 2));
#pragma unroll
                for (int r = 0; r < 4; ++r) z[j][r] = (w[r] >= dc.threshold) ? z[j][r] * dc.scale : 0.f;
            }
            if (x) {
                float xv[4];
                load4(x + c, xv);
#pragma unroll
                for (int r = 0; r < 4; ++r) z[j][r] += xv[r];
            }
            if (a.addvec) {
                const f32x4 av = *(const f32x4*)(a.addvec + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) z[j][r] += av[r];
            }
            sum += z[j][0] + z[j][1] + z[j][2] + z[j][3];
        }
    }
    const float mean = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = j * 256 + lane * 4;
        if (c < H) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float t = z[j][r] - mean; sq += t * t; }
        }
    }
    const float var = wave_sum(sq) / (float)H;
    const float rstd = 1.0f / sqrtf(var + LN_EPS);
    if (lane == 0) { a.mean[row] = mean; a.rstd[row] = rstd; }
    uint16_t* y = (uint16_t*)a.y + (size_t)row * H;
    uint16_t* zs = a.z ? (uint16_t*)a.z + (size_t)row * H : nullptr;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = j * 256 + lane * 4;
        if (c < H) {
            const f32x4 g = *(const f32x4*)(a.gamma + c);
            const f32x4 b = *(const f32x4*)(a.beta + c);
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = g[r] * ((z[j][r] - mean) * rstd) + b[r];
            if (drop_on && a.post) {
                u32x4 w = drop_words(dcfg, seed, drow, (uint32_t)(c >> 2));
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (w[r] >= dc.threshold) ? o[r] * dc.scale : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= a.out_scale;
            if (zs) store4(zs + c, z[j]);
            store4(y + c, o);
            if (a.y8) {                    // keep the outputs (z is dead now) for the row-wise e4m3 copy
#pragma unroll
                for (int r = 0; r < 4; ++r) { z[j][r] = o[r]; amax = fmaxf(amax, fabsf(o[r])); }
            }
        }
    }
    if (a.y8) {        // fp8 projection path: the consumer GEMM's A operand leaves this kernel already quantised (vk_quant_rows_fp8 semantics)
        amax = wave_max(amax);
        const float sc = amax > 0.f ? amax / 448.0f : 1.0f, inv = 1.0f / sc;
        if (lane == 0) a.y8_scale[row] = sc;
        uint8_t* q = (uint8_t*)a.y8 + (size_t)row * a.ld8;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < H) *(uint32_t*)(q + c) = pack4_fp8(z[j][0] * inv, z[j][1] * inv, z[j][2] * inv, z[j][3] * inv);
        }
    }
}

// `stamps` (study builds; NULL in the shipped library's launches): per workgroup four s_memrealtime readings (100 MHz) -- entry, first
// row's statistics done (= its operands arrived), last row stored, record written -- into a buffer no other code reads (tools/stamp_ln.py).
template <int NCH>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const JobPair<vk_ln_bwd_args> jp, unsigned long long* const stamps) {
    __shared__ float red[4][2][NCH * 256];
    const bool stamping = stamps != nullptr && threadIdx.x == 0;
#ifdef VK_STUDY
    if (jp.stagger > 0) {          // experiment: de-phase the workgroups of the single resident round (measured: no gain)
        const int k = jp.stagger_mod > 1 ? (int)(blockIdx.x % (unsigned)jp.stagger_mod) : 0;
        const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(k * jp.stagger);
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(4);
    }
#endif
    if (stamping) stamps[(size_t)blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
    const int nb0 = jp.nb0;
    const bool second = (int)blockIdx.x >= nb0;
    const vk_ln_bwd_args a = load_job<vk_ln_bwd_args>(second);
    const int blk = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int H = a.H;
    const bool drop_on = a.drop.threshold != 0;
    const uint64_t seed = drop_on ? *a.drop.seed : 0;
    float pg[NCH][4], pb[NCH][4];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) pg[j][r] = pb[j][r] = 0.f;

    const int Mrows = a.dyn ? min(*a.dyn, a.M) : a.M;
    // Each wave owns LN_BWD_ROWS / 4 rows; the bf16 inputs of ALL of them are requested up front (raw 8-byte
    // loads, 2 VGPRs per 4 elements) so that several rows' worth of HBM latency overlap instead of being paid
    // one row after the other.
    // A wave owns 4 rows.  Two of them are in flight at any time: a row's raw registers request the row after next as soon as they
    // have been unpacked (all four up front cost 154 VGPRs = 3 waves per SIMD, and 3648 waves on 3072 slots ran as two rounds with
    // the second one fifth full; two in flight fit 4 waves per SIMD: one round).
    constexpr int RPW = LN_BWD_ROWS / 4, PRE = 2;
    u32x2 rdy[PRE][NCH], rz[PRE][NCH];
    auto request = [&](int it, u32x2 (&dyv)[NCH], u32x2 (&zv)[NCH]) {
        const int row = blk * LN_BWD_ROWS + it * 4 + wave;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
            dyv[j] = u32x2{0u, 0u};
            zv[j] = u32x2{0u, 0u};
            if (it < RPW && row < Mrows && c < H) {
                dyv[j] = *(const u32x2*)((const uint16_t*)a.dy + (size_t)row * H + c);
                zv[j] = *(const u32x2*)((const uint16_t*)a.z + (size_t)row * H + c);
            }
        }
    };
#pragma unroll
    for (int it = 0; it < PRE; ++it) request(it, rdy[it], rz[it]);
#pragma unroll
    for (int it = 0; it < RPW; ++it) {
        const int row = blk * LN_BWD_ROWS + it * 4 + wave;
        if (row >= Mrows) break;
        uint32_t dsite;
        const uint32_t drow = drop_row(a.seg, a.split_row, row, dsite);
        DropCfg dcfg{a.drop.seed, dsite, a.drop.threshold, a.drop.scale};
        const float mean = a.mean[row], rstd = a.rstd[row];
        float xh[NCH][4], gh[NCH][4];
        uint32_t keep[NCH];              // bit r: element r of the chunk survives the dropout (4 bits instead of the 4 Philox words)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) xh[j][r] = gh[j][r] = 0.f;
            if (c < H) {
                const u32x2 d2 = rdy[it % PRE][j], z2 = rz[it % PRE][j];
                const float dv[4] = {bf2f(d2[0] & 0xFFFF), bf2f(d2[0] >> 16), bf2f(d2[1] & 0xFFFF), bf2f(d2[1] >> 16)};
                const float zv[4] = {bf2f(z2[0] & 0xFFFF), bf2f(z2[0] >> 16), bf2f(z2[1] & 0xFFFF), bf2f(z2[1] >> 16)};
                keep[j] = 0xFu;
                if (drop_on) {
                    const u32x4 w = drop_words(dcfg, seed, drow, (uint32_t)(c >> 2));
                    keep[j] = (w[0] >= a.drop.threshold ? 1u : 0u) | (w[1] >= a.drop.threshold ? 2u : 0u) | (w[2] >= a.drop.threshold ? 4u : 0u) | (w[3] >= a.drop.threshold ? 8u : 0u);
                }
                const f32x4 g = *(const f32x4*)(a.gamma + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float gy = dv[r] * a.out_scale;
                    if (drop_on && a.post) gy = ((keep[j] >> r) & 1u) ? gy * a.drop.scale : 0.f;
                    const float xv = (zv[r] - mean) * rstd;
                    pg[j][r] += gy * xv;
                    pb[j][r] += gy;
                    const float gx = gy * g[r];
                    xh[j][r] = xv;
                    gh[j][r] = gx;
                    s1 += gx;
                    s2 += gx * xv;
                }
            }
        }
        request(it + PRE, rdy[it % PRE], rz[it % PRE]);        // this row's raw registers are free: they fetch the row after next
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
        if (stamping && it == 0) stamps[(size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        uint16_t* dz = (uint16_t*)a.dz + (size_t)row * H;
        uint16_t* dd = a.dd ? (uint16_t*)a.dd + (size_t)row * H : nullptr;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < H) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = rstd * (gh[j][r] - s1 - xh[j][r] * s2);
                store4(dz + c, o);
                if (dd) {
                    if (drop_on && !a.post) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = ((keep[j] >> r) & 1u) ? o[r] * a.drop.scale : 0.f;
                    }
                    store4(dd + c, o);
                }
            }
        }
    }
    if (stamping) stamps[(size_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
    // reduce the four waves' column partials, one [2][H] record per workgroup
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[wave][0][j * 256 + lane * 4 + r] = pg[j][r];
            red[wave][1][j * 256 + lane * 4 + r] = pb[j][r];
        }
    __syncthreads();
    float* out = a.partial + (size_t)blk * 2 * H;
    for (int i = threadIdx.x; i < 2 * NCH * 256; i += LN_THREADS) {
        const int which = i / (NCH * 256), c = i - which * NCH * 256;
        if (c < H) out[which * H + c] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
    if (stamping) stamps[(size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
}

// column sums of the per-workgroup partial records: 64 columns x 16 row groups per workgroup
__global__ __launch_bounds__(1024) void ln_bwd_finalize_kernel(const float* partial, int nblk, int H, float* dgamma, float* dbeta, int accumulate) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (i < 2 * H)
        for (int b = ty; b < nblk; b += 16) s += partial[(size_t)b * 2 * H + i];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && i < 2 * H) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += red[k][tx];
        float* o = i < H ? dgamma + i : dbeta + (i - H);
        *o = accumulate ? *o + s : s;
    }
}

}  // namespace vk

static int ln_check(int H, int maxH, const char* who) {
    if (H % 4 || H > maxH || H <= 0) return vk::set_error("%s: H=%d must be a multiple of 4, <= %d", who, H, maxH);
    return 0;
}

extern "C" int vk_ln_fwd_pair(const vk_ln_args* a, const vk_ln_args* b, vk_stream_t stream) {
    using namespace vk;
    if (ln_check(a->H, 2048, "vk_ln_fwd")) return -1;
    if (b && (b->H != a->H || a->dyn || b->dyn)) return set_error("vk_ln_fwd_pair: both jobs need the same H and static row counts");
    const int nb0 = a->M > 0 ? (a->M + 3) / 4 : 0, nb1 = (b && b->M > 0) ? (b->M + 3) / 4 : 0;
    if (nb0 + nb1 == 0) return 0;
    const int nch = (a->H + 255) / 256;
    dim3 grid(nb0 + nb1), block(LN_THREADS);
    hipStream_t s = (hipStream_t)stream;
    JobPair<vk_ln_args> jp;
    jp.job[0] = *a;
    jp.job[1] = b ? *b : *a;
    jp.nb0 = nb0;
    jp.stagger = jp.stagger_mod = 0;
    switch (nch) {
        case 1: hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, block, 0, s, jp); break;
        case 2: hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, block, 0, s, jp); break;
        case 3: hipLaunchKernelGGL(ln_fwd_kernel<3>, grid, block, 0, s, jp); break;
        case 4: hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, block, 0, s, jp); break;
        default: hipLaunchKernelGGL(ln_fwd_kernel<8>, grid, block, 0, s, jp); break;
    }
    return check_launch("vk_ln_fwd");
}

extern "C" int vk_ln_fwd(const vk_ln_args* a, vk_stream_t stream) { return vk_ln_fwd_pair(a, nullptr, stream); }

extern "C" int vk_ln_bwd_partial_rows(int M) { return (M + vk::LN_BWD_ROWS - 1) / vk::LN_BWD_ROWS; }

#ifdef VK_STUDY
static unsigned long long* g_ln_stamps = nullptr;          // tools/stamp_ln.py
static int g_ln_stagger = 0, g_ln_stagger_mod = 0;          // experiment: workgroup i waits (i % mod) * stagger ticks (10 ns) before it starts
extern "C" void vk_ln_set_stamp_buffer(unsigned long long* p) { g_ln_stamps = p; }
extern "C" void vk_ln_set_stagger(int ticks, int mod) { g_ln_stagger = ticks; g_ln_stagger_mod = mod; }
#else
static constexpr unsigned long long* g_ln_stamps = nullptr;
static constexpr int g_ln_stagger = 0, g_ln_stagger_mod = 0;
#endif

static void ln_finalize_launch(const vk_ln_bwd_args* a, hipStream_t s) {
    hipLaunchKernelGGL(vk::ln_bwd_finalize_kernel, dim3((2 * a->H + 63) / 64), dim3(1024), 0, s, a->partial, vk_ln_bwd_partial_rows(a->M), a->H,
                       a->dgamma, a->dbeta, a->accumulate & 1);
}

extern "C" int vk_ln_bwd_pair(const vk_ln_bwd_args* a, const vk_ln_bwd_args* b, vk_stream_t stream) {
    using namespace vk;
    if (ln_check(a->H, 2048, "vk_ln_bwd")) return -1;
    if (b && (b->H != a->H || a->dyn || b->dyn)) return set_error("vk_ln_bwd_pair: both jobs need the same H and static row counts");
    const int nb0 = a->M > 0 ? vk_ln_bwd_partial_rows(a->M) : 0, nb1 = (b && b->M > 0) ? vk_ln_bwd_partial_rows(b->M) : 0;
    if (nb0 + nb1 == 0) return 0;
    const int nch = (a->H + 255) / 256;
    dim3 grid(nb0 + nb1), block(LN_THREADS);
    hipStream_t s = (hipStream_t)stream;
    JobPair<vk_ln_bwd_args> jp;
    jp.job[0] = *a;
    jp.job[1] = b ? *b : *a;
    jp.nb0 = nb0;
    jp.stagger = g_ln_stagger; jp.stagger_mod = g_ln_stagger_mod;
    switch (nch) {
        case 1: hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, block, 0, s, jp, g_ln_stamps); break;
        case 2: hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, block, 0, s, jp, g_ln_stamps); break;
        case 3: hipLaunchKernelGGL(ln_bwd_kernel<3>, grid, block, 0, s, jp, g_ln_stamps); break;
        case 4: hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, block, 0, s, jp, g_ln_stamps); break;
        case 5: case 6: hipLaunchKernelGGL(ln_bwd_kernel<6>, grid, block, 0, s, jp, g_ln_stamps); break;      // clf_hidden_size 1536 (task classifiers)
        default: hipLaunchKernelGGL(ln_bwd_kernel<8>, grid, block, 0, s, jp, g_ln_stamps); break;
    }
    if (nb0 && !(a->accumulate & 2)) ln_finalize_launch(a, s);
    if (nb1 && !(b->accumulate & 2)) ln_finalize_launch(b, s);
    return check_launch("vk_ln_bwd");
}

extern "C" int vk_ln_bwd(const vk_ln_bwd_args* a, vk_stream_t stream) { return vk_ln_bwd_pair(a, nullptr, stream); }

extern "C" int vk_ln_bwd_finalize(const vk_ln_bwd_args* a, vk_stream_t stream) {
    if (a->M <= 0) return 0;
    ln_finalize_launch(a, (hipStream_t)stream);
    return vk::check_launch("vk_ln_bwd_finalize");
}
