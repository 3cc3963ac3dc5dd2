// Optimizer-side kernels over the flat fp32 parameter / gradient arenas (HBM-bound, 16-byte accesses):
//   * global gradient L2 norm + clip coefficient (torch.nn.utils.clip_grad_norm_, train_concap.py:307-308),
//     left on the device: no host sync
//   * fused multi-tensor AdamW with pytorch-transformers 1.1.0 semantics (train_concap.py:227,310): decay
//     applied AFTER the Adam update with the un-corrected lr, eps added to sqrt(v) before bias correction.
//     One launch for all tensors (the reference steps 508 parameter groups from Python); also refreshes
//     the bf16 shadow copy of the weights used by the MFMA GEMMs.  Replaces nothing in apex that volta
//     calls (apex/csrc/multi_tensor_adam.cu is the closest relative, with different arithmetic).
#include "common.h"
#include "../../include/volta_hip.h"
#include "util.h"

namespace vk {

constexpr int NORM_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* g, size_t n4, float* partial, const uint8_t* chunk_class) {
    __shared__ float sh[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (chunk_class && chunk_class[i >> 8] == VK_CHUNK_SKIP) continue;      // 256 float4 per 1024-element chunk
        const f32x4 v = *(const f32x4*)(g + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// out[0] = ||g * pre_scale||, out[1] = clip coefficient (multiplies the gradients in the optimizer)
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* partial, int nblk, float pre_scale, float max_norm, float* out) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += (double)partial[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(sh[0] + sh[1] + sh[2] + sh[3]) * pre_scale;
        out[0] = norm;
        float coef = 1.f;
        if (max_norm > 0.f) { coef = max_norm / (norm + 1e-6f); if (coef > 1.f) coef = 1.f; }
        out[1] = coef;
    }
}

// shard-decomposable norm: one wave per 1024-element chunk, lanes sum 4 x float4 in a fixed order
__global__ __launch_bounds__(256) void sqnorm_chunks_kernel(const float* g, size_t chunk0, size_t nchunks, const uint8_t* chunk_class, float* sums) {
    const size_t c = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const size_t chunk = chunk0 + c;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    if (!(chunk_class && chunk_class[chunk] == VK_CHUNK_SKIP)) {
        const float* p = g + chunk * 1024;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 v = __builtin_nontemporal_load((const f32x4*)(p + (k * 64 + lane) * 4));
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
    }
    s = wave_sum(s);
    if (lane == 0) sums[chunk] = s;
}
// fixed-order sum of the chunk sums in two levels: SQ_GROUPS workgroups add contiguous runs in double, one workgroup adds their results
constexpr int SQ_GROUPS = 128;
__global__ __launch_bounds__(256) void sqnorm_groups_kernel(const float* sums, size_t n, double* group) {
    __shared__ double sh[4];
    const size_t per = (n + SQ_GROUPS - 1) / SQ_GROUPS, lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    double s = 0.0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) s += (double)sums[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) group[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(64) void sqnorm_from_groups_kernel(const double* group, float pre_scale, float max_norm, float* out) {
    double s = group[threadIdx.x] + group[threadIdx.x + 64];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(s) * pre_scale;
        out[0] = norm;
        float coef = 1.f;
        if (max_norm > 0.f) { coef = max_norm / (norm + 1e-6f); if (coef > 1.f) coef = 1.f; }
        out[1] = coef;
    }
}

// One element of the update, spelled with explicit fused multiply-adds and no further contraction: vk_adamw_step and vk_adamw_step_on give
// the same bits whatever the compiler makes of the code around it.
__device__ __forceinline__ void adamw_element(float& p, float& m, float& v, float g, float gs, float b1, float b2, float eps, float step, float lr_wd) {
#pragma clang fp contract(off)
    const float gr = g * gs;
    m = __builtin_fmaf(b1, m, (1.f - b1) * gr);
    v = __builtin_fmaf(b2, v, ((1.f - b2) * gr) * gr);
    p = __builtin_fmaf(-step, m / (sqrtf(v) + eps), p);
    if (lr_wd > 0.f) p = __builtin_fmaf(-lr_wd, p, p);
}

__global__ __launch_bounds__(256) void adamw_kernel(vk_adamw_args a) {
    const size_t chunk = blockIdx.x;
    const int cls = a.chunk_class ? a.chunk_class[chunk] : 0;
    if (cls == VK_CHUNK_SKIP) return;        // frozen / gradient-less parameters: weights, moments and the bf16 copy stay as they are
    const float lr = a.lr * a.cls_lr_mult[cls], wd = a.cls_wd[cls];
    const float gs = a.grad_scale * (a.clip ? a.clip[1] : 1.f);
    const size_t i = chunk * 1024 + threadIdx.x * 4;
    // one pass over 7 GB that nothing reads again before the next step's optimizer: non-temporal, so that a step pipelined under the next
    // forward does not sweep that forward's activations out of the Infinity Cache; the bf16 copies are what the forward reads
    const f32x4 g = __builtin_nontemporal_load((const f32x4*)(a.g + i));
    f32x4 p = __builtin_nontemporal_load((const f32x4*)(a.p + i)), m = __builtin_nontemporal_load((const f32x4*)(a.m + i)), v = __builtin_nontemporal_load((const f32x4*)(a.v + i));
    const float step = lr * a.step_mult;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float pr = p[r], mr = m[r], vr = v[r];
        adamw_element(pr, mr, vr, g[r], gs, a.beta1, a.beta2, a.eps, step, lr * wd);
        p[r] = pr; m[r] = mr; v[r] = vr;
    }
    __builtin_nontemporal_store(p, (f32x4*)(a.p + i)); __builtin_nontemporal_store(m, (f32x4*)(a.m + i)); __builtin_nontemporal_store(v, (f32x4*)(a.v + i));
    if (a.shadow) *(u32x2*)((uint16_t*)a.shadow + i) = u32x2{pack2bf(p[0], p[1]), pack2bf(p[2], p[3])};
}

// The same update by a few RESIDENT workgroups (vk_adamw_step_on): 512 threads, an LDS footprint that keeps the compute unit to itself, eight
// 2048-element blocks in flight per trip (256 KiB of loads per CU -- what one CU needs to stream at ~100 GB/s against HBM latency).  For the
// optimizer step that runs under the next forward pass: the persistent GEMM launches there claim 232 of the 256 CUs (gemm256.hip,
// persistent_grid) and a full-width AdamW launch only gets CUs when a GEMM launch gives them up -- it delays the forward by what it takes
// (measured: pipelined or not, clip + AdamW add 1.25 ms to the step).  24 workgroups on the CUs the GEMMs leave alone do not.
__global__ __launch_bounds__(512) void adamw_narrow_kernel(vk_adamw_args a) {
    constexpr int U = 8;                                      // 2048-element blocks in flight per workgroup: 512 threads x 8 x 4 arrays x 16 bytes = 256 KiB of loads
    const size_t nch = (size_t)a.n / 1024;
    const int sub = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)), t4 = (threadIdx.x & 255) * 4;
    const float gs = a.grad_scale * (a.clip ? a.clip[1] : 1.f);
    for (size_t blk = blockIdx.x; blk * 2 < nch; blk += (size_t)gridDim.x * U) {
        f32x4 g[U], p[U], m[U], v[U];
        int cls[U];
        size_t at[U];
        // every load of the trip goes out before anything is waited for: no branch in here (a chunk past the end re-reads the last one,
        // a skipped chunk is read and dropped)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t chunk = (blk + (size_t)u * gridDim.x) * 2 + sub;
            const size_t c = chunk < nch ? chunk : nch - 1;
            int k = 0;
            if (a.chunk_class) {                              // wave-uniform (a wave lies inside one chunk): fetched on the scalar path, as the aligned word that holds the byte
                const uintptr_t ca = (uintptr_t)(a.chunk_class + c);
                k = (int)((*(const uint32_t*)(ca & ~(uintptr_t)3) >> (8 * (ca & 3))) & 255u);
            }
            cls[u] = chunk < nch ? k : VK_CHUNK_SKIP;
            at[u] = c * 1024 + t4;
            g[u] = __builtin_nontemporal_load((const f32x4*)(a.g + at[u]));
            p[u] = __builtin_nontemporal_load((const f32x4*)(a.p + at[u]));
            m[u] = __builtin_nontemporal_load((const f32x4*)(a.m + at[u]));
            v[u] = __builtin_nontemporal_load((const f32x4*)(a.v + at[u]));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = __builtin_amdgcn_readfirstlane(cls[u]);
            if (c == VK_CHUNK_SKIP) continue;
            const float lr = a.lr * a.cls_lr_mult[c], wd = a.cls_wd[c], step = lr * a.step_mult;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr = p[u][r], mr = m[u][r], vr = v[u][r];
                adamw_element(pr, mr, vr, g[u][r], gs, a.beta1, a.beta2, a.eps, step, lr * wd);
                p[u][r] = pr; m[u][r] = mr; v[u][r] = vr;
            }
            const size_t i = at[u];
            __builtin_nontemporal_store(p[u], (f32x4*)(a.p + i)); __builtin_nontemporal_store(m[u], (f32x4*)(a.m + i)); __builtin_nontemporal_store(v[u], (f32x4*)(a.v + i));
            if (a.shadow) *(u32x2*)((uint16_t*)a.shadow + i) = u32x2{pack2bf(p[u][0], p[u][1]), pack2bf(p[u][2], p[u][3])};
        }
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* y, const float* x, float alpha, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 a = *(f32x4*)(y + i * 4);
        const f32x4 b = *(const f32x4*)(x + i * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] += alpha * b[r];
        *(f32x4*)(y + i * 4) = a;
    }
}

__global__ __launch_bounds__(256) void sum_slabs_kernel(float* dst, const float* src, size_t stride, int nslabs, size_t n) {
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 a = *(const f32x4*)(src + i * 4);
        for (int s = 1; s < nslabs; ++s) {
            const f32x4 b = *(const f32x4*)(src + (size_t)s * stride + i * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] += b[r];
        }
        *(f32x4*)(dst + i * 4) = a;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        float a = 0.f;
        for (int s = 0; s < nslabs; ++s) a += src[(size_t)s * stride + i];
        dst[i] = a;
    }
}

__global__ __launch_bounds__(256) void sum_slabs_bf16_kernel(uint16_t* dst, const float* src, size_t stride, int nslabs, size_t n, const int32_t* dyn_rows, int row_len) {
    if (dyn_rows) { const size_t lim = (size_t)(*dyn_rows) * row_len; if (lim < n) n = lim; }
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 a = *(const f32x4*)(src + i * 4);
        for (int s = 1; s < nslabs; ++s) {
            const f32x4 b = *(const f32x4*)(src + (size_t)s * stride + i * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] += b[r];
        }
        *(u32x2*)(dst + i * 4) = u32x2{pack2bf(a[0], a[1]), pack2bf(a[2], a[3])};
    }
}

struct TailJobs { vk_tail_job j[VK_TAIL_MAX_JOBS]; int32_t njobs; };

// One launch for every small reduction that ends a sub-layer's weight-gradient block (96 slab sums + 64 LayerNorm column
// reductions per ViLBERT step used to be 160 launches of 5-13 us on the side stream).  Workgroup -> job by block_start.
__global__ __launch_bounds__(256) void side_tail_kernel(const TailJobs t) {
    __shared__ float red[16][17];
    int ji = 0;
#pragma unroll
    for (int i = 1; i < VK_TAIL_MAX_JOBS; ++i)
        if (i < t.njobs && (int)blockIdx.x >= t.j[i].block_start) ji = i;
    const vk_tail_job& J = t.j[ji];
    const int blk = (int)blockIdx.x - J.block_start;
    if (J.kind == 0) {
        const size_t n4 = (size_t)J.n >> 2;
        const size_t i = (size_t)blk * 256 + threadIdx.x;
        if (i < n4) {
            // the slabs are read once, here: non-temporal loads keep them from displacing the backward chain's tensors in the Infinity Cache
            f32x4 a = __builtin_nontemporal_load((const f32x4*)(J.src + i * 4));
            for (int s = 1; s < J.count; ++s) {
                const f32x4 b = __builtin_nontemporal_load((const f32x4*)(J.src + (size_t)s * J.stride + i * 4));
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] += b[r];
            }
            __builtin_nontemporal_store(a, (f32x4*)(J.dst + i * 4));      // the gradient arena: next read by the norm / optimizer, a step's length away
        }
        if (blk == 0 && threadIdx.x < (J.n & 3)) {
            const size_t k = n4 * 4 + threadIdx.x;
            float a = 0.f;
            for (int s = 0; s < J.count; ++s) a += J.src[(size_t)s * J.stride + k];
            J.dst[k] = a;
        }
    } else {
        // 16 columns x 16 row groups per workgroup over the [count][2 H] partial records
        const int H = (int)J.n, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
        const int i = blk * 16 + tx;
        float s = 0.f;
        if (i < 2 * H) {
            for (int b = ty; b < J.count; b += 16) s += J.src[(size_t)b * 2 * H + i];
            if (J.src2)
                for (int b = ty; b < (int)J.stride; b += 16) s += J.src2[(size_t)b * 2 * H + i];
        }
        red[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && i < 2 * H) {
#pragma unroll
            for (int k = 1; k < 16; ++k) s += red[k][tx];
            float* o = i < H ? J.dst + i : J.dst2 + (i - H);
            *o = J.accumulate ? *o + s : s;
        }
    }
}

}  // namespace vk

using namespace vk;

extern "C" int vk_side_tail(const vk_tail_job* jobs, int njobs, vk_stream_t s) {
    if (njobs <= 0) return 0;
    if (njobs > VK_TAIL_MAX_JOBS) return set_error("vk_side_tail: %d jobs (max %d)", njobs, VK_TAIL_MAX_JOBS);
    TailJobs t;
    t.njobs = njobs;
    int total = 0;
    for (int i = 0; i < njobs; ++i) {
        t.j[i] = jobs[i];
        vk_tail_job& J = t.j[i];
        int nb;
        if (J.kind == 0) {
            if ((J.stride & 3) || ((uintptr_t)J.dst & 15) || ((uintptr_t)J.src & 15)) return set_error("vk_side_tail: slab job %d needs 16-byte alignment", i);
            nb = (int)((J.n / 4 + 255) / 256);
            if (nb < 1) nb = 1;
        } else if (J.kind == 1) {
            nb = (int)((2 * J.n + 15) / 16);
        } else return set_error("vk_side_tail: unknown job kind %d", J.kind);
        if (J.n <= 0 || J.count <= 0) nb = 0;
        J.block_start = total;
        total += nb;
    }
    if (total == 0) return 0;
    hipLaunchKernelGGL(side_tail_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)s, t);
    return check_launch("vk_side_tail");
}

extern "C" int vk_sum_slabs_bf16(void* dst, const float* src, int64_t slab_stride, int nslabs, int64_t n, const int32_t* dyn_rows, int row_len, vk_stream_t s) {
    if (n <= 0 || nslabs <= 0) return 0;
    if ((slab_stride & 3) || (n & 3) || (row_len & 3)) return set_error("vk_sum_slabs_bf16: lengths must be multiples of 4");
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(sum_slabs_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (uint16_t*)dst, src, (size_t)slab_stride, nslabs, (size_t)n, dyn_rows, row_len);
    return check_launch("vk_sum_slabs_bf16");
}

extern "C" int vk_sum_slabs_f32(float* dst, const float* src, int64_t slab_stride, int nslabs, int64_t n, vk_stream_t s) {
    if (n <= 0 || nslabs <= 0) return 0;
    if ((slab_stride & 3) || ((uintptr_t)dst & 15) || ((uintptr_t)src & 15)) return set_error("vk_sum_slabs_f32: 16-byte alignment required");
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, dst, src, (size_t)slab_stride, nslabs, (size_t)n);
    return check_launch("vk_sum_slabs_f32");
}

extern "C" int vk_grad_norm_workspace_floats(void) { return NORM_BLOCKS; }

extern "C" int vk_grad_norm_clip_masked(const float* g, int64_t n, const uint8_t* chunk_class, float pre_scale, float max_norm, float* partial, float* out, vk_stream_t s) {
    if (n % 4 || ((uintptr_t)g & 15)) return set_error("vk_grad_norm_clip: n %% 4 == 0 and 16-byte alignment required");
    if (chunk_class && (n % 1024)) return set_error("vk_grad_norm_clip_masked: a chunk mask needs an arena of whole 1024-element chunks");
    hipStream_t st = (hipStream_t)s;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, st, g, (size_t)(n / 4), partial, chunk_class);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, st, partial, NORM_BLOCKS, pre_scale, max_norm, out);
    return check_launch("vk_grad_norm_clip");
}

extern "C" int vk_grad_sqnorm_chunks(const float* g, int64_t chunk0, int64_t nchunks, const uint8_t* chunk_class, float* sums, vk_stream_t s) {
    if (chunk0 < 0 || nchunks < 0 || ((uintptr_t)g & 15)) return set_error("vk_grad_sqnorm_chunks: bad range or alignment");
    if (nchunks == 0) return 0;
    hipLaunchKernelGGL(sqnorm_chunks_kernel, dim3((unsigned)((nchunks + 3) / 4)), dim3(256), 0, (hipStream_t)s, g, (size_t)chunk0, (size_t)nchunks, chunk_class, sums);
    return check_launch("vk_grad_sqnorm_chunks");
}

extern "C" int vk_grad_norm_from_chunks(float* sums, int64_t total_chunks, float pre_scale, float max_norm, float* out, vk_stream_t s) {
    if (total_chunks <= 0) return set_error("vk_grad_norm_from_chunks: no chunks");
    if (((uintptr_t)(sums + total_chunks)) & 7) return set_error("vk_grad_norm_from_chunks: total_chunks must be even (the group sums follow the chunk sums as doubles)");
    double* group = (double*)(sums + total_chunks);           // 128 doubles behind the chunk sums (see the header)
    hipLaunchKernelGGL(sqnorm_groups_kernel, dim3(SQ_GROUPS), dim3(256), 0, (hipStream_t)s, sums, (size_t)total_chunks, group);
    hipLaunchKernelGGL(sqnorm_from_groups_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, (const double*)group, pre_scale, max_norm, out);
    return check_launch("vk_grad_norm_from_chunks");
}

extern "C" int vk_grad_norm_clip(const float* g, int64_t n, float pre_scale, float max_norm, float* partial, float* out, vk_stream_t s) {
    return vk_grad_norm_clip_masked(g, n, nullptr, pre_scale, max_norm, partial, out, s);
}

extern "C" int vk_adamw_step(const vk_adamw_args* a, vk_stream_t s) {
    if (a->n % 1024) return set_error("vk_adamw_step: arena length must be a multiple of 1024 elements");
    if (a->n == 0) return 0;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)(a->n / 1024)), dim3(256), 0, (hipStream_t)s, *a);
    return check_launch("vk_adamw_step");
}

extern "C" int vk_adamw_step_on(const vk_adamw_args* a, int ncus, vk_stream_t s) {
    if (a->n % 1024) return set_error("vk_adamw_step_on: arena length must be a multiple of 1024 elements");
    if (ncus < 1 || ncus > 256) return set_error("vk_adamw_step_on: %d compute units", ncus);
    if (a->n == 0) return 0;
    constexpr int FOOTPRINT = 96 * 1024;             // more than half a CU's LDS: one workgroup per compute unit
    static const hipError_t attr = hipFuncSetAttribute((const void*)adamw_narrow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FOOTPRINT); (void)attr;
    hipLaunchKernelGGL(adamw_narrow_kernel, dim3((unsigned)ncus), dim3(512), FOOTPRINT, (hipStream_t)s, *a);
    return check_launch("vk_adamw_step_on");
}

extern "C" int vk_axpy_f32(float* y, const float* x, float alpha, int64_t n, vk_stream_t s) {
    if (n % 4) return set_error("vk_axpy_f32: n %% 4 != 0");
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3(2048), dim3(256), 0, (hipStream_t)s, y, x, alpha, (size_t)(n / 4));
    return check_launch("vk_axpy_f32");
}

extern "C" int vk_memset_async(void* p, int value, int64_t bytes, vk_stream_t s) {
    if (bytes <= 0) return 0;
    hipError_t e = hipMemsetAsync(p, value, (size_t)bytes, (hipStream_t)s);
    if (e != hipSuccess) return set_error("vk_memset_async: %s", hipGetErrorString(e));
    return 0;
}
