// Pre-training heads and losses on LABELLED ROWS ONLY (identical losses, several times fewer rows):
//   * order-preserving compaction of labelled positions (device-side count, no host sync)
//   * row gather / scatter-add
//   * masked-LM / ITM cross entropy (CrossEntropyLoss(ignore_index=-1), volta/encoders.py:1095-1107)
//   * kl_1601 masked-region loss (volta/losses.py:16-22)
//   * pooled = dropout(pooled_t * pooled_v) (encoders.py:769-770) and the additive attention masks
//     (encoders.py:983-991)
#include "common.h"
#include "../../include/volta_hip.h"
#include "util.h"

namespace vk {

// ---- compaction ---------------------------------------------------------------------------------
// flag(i) = labels[i] != -1 (mode 0) or labels[i] == 1 (mode 1); for the j-th flagged i:
//   pos[j] = i ; rows[j] = (i / inner) * outer + (i % inner) + off ; *count = number flagged.
__global__ __launch_bounds__(1024) void select_rows_kernel(const int64_t* labels, int N, int mode, int inner, int outer, int off,
                                                           int32_t* rows, int32_t* pos, int32_t* count) {
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < N; start += 1024) {
        const int i = start + tid;
        bool f = false;
        if (i < N) { const int64_t l = labels[i]; f = mode ? (l == 1) : (l != -1); }
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int wbase = base_s;
        for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
        if (f) {
            const int j = wbase + before;
            pos[j] = i;
            rows[j] = (i / inner) * outer + (i % inner) + off;
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wave_tot[w]; base_s += t; }
        __syncthreads();
    }
    if (tid == 0) *count = base_s;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const uint16_t* src, const int32_t* rows, const int32_t* count, uint16_t* dst, int H, int maxrows) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = min(*count, maxrows);
    if (i >= n) return;
    const size_t r = (size_t)rows[i];
    for (int c = lane * 8; c < H; c += 512) *(u32x4*)(dst + (size_t)i * H + c) = *(const u32x4*)(src + r * H + c);
}

__global__ __launch_bounds__(256) void scatter_rows_add_kernel(const uint16_t* src, const int32_t* rows, const int32_t* count, uint16_t* dst, int H, int maxrows) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = min(*count, maxrows);
    if (i >= n) return;
    const size_t r = (size_t)rows[i];
    for (int c = lane * 4; c < H; c += 256) {
        const u32x2 a = *(const u32x2*)(src + (size_t)i * H + c), b = *(const u32x2*)(dst + r * H + c);
        *(u32x2*)(dst + r * H + c) = u32x2{pack2bf(bf2f(a[0] & 0xFFFF) + bf2f(b[0] & 0xFFFF), bf2f(a[0] >> 16) + bf2f(b[0] >> 16)),
                                           pack2bf(bf2f(a[1] & 0xFFFF) + bf2f(b[1] & 0xFFFF), bf2f(a[1] >> 16) + bf2f(b[1] >> 16))};
    }
}

// ---- block reductions -----------------------------------------------------------------------------
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, sh[w]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += sh[w];
    return r;
}

// ---- cross entropy ----------------------------------------------------------------------------------
// one workgroup per labelled row i: label = labels[pos ? pos[i] : i]; loss_sum += lse - logit[label].
// ONE pass over the row (122 KB of fp32 logits for the LM head): every thread keeps a running (max, sum of exp) pair over 16-byte pieces
// and rescales the sum when its maximum moves; the pairs are combined once per row.  (The two-pass form with 4-byte loads read the row
// twice and took 58 us for ~700 rows: 85 MB at 1.5 TB/s.)
__device__ __forceinline__ void online_add(float v, float& m, float& s) {
    if (v > m) { s *= __expf(m - v); m = v; }
    s += __expf(v - m);
}
__global__ __launch_bounds__(256) void xent_fwd_kernel(vk_xent_args a) {
    __shared__ float sh[4];
    const int i = blockIdx.x;
    const int n = a.count ? min(*a.count, a.max_rows) : a.max_rows;
    if (i >= n) return;
    const float* x = a.logits + (size_t)i * a.ld;
    float m = -INFINITY, s = 0.f;
    const bool vec = (((uintptr_t)x) & 15) == 0;
    const int n4 = vec ? a.V >> 2 : 0;
    for (int c = threadIdx.x; c < n4; c += 256) {
        const f32x4 v = *(const f32x4*)(x + 4 * c);
        const float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (vm > m) { s *= __expf(m - vm); m = vm; }
        s += __expf(v[0] - m) + __expf(v[1] - m) + __expf(v[2] - m) + __expf(v[3] - m);
    }
    for (int c = 4 * n4 + threadIdx.x; c < a.V; c += 256) online_add(x[c], m, s);
    const float mx = block_max(m, sh);
    s = block_sum(m == -INFINITY ? 0.f : s * __expf(m - mx), sh);
    if (threadIdx.x == 0) {
        const float lse = mx + __logf(s);
        a.lse[i] = lse;
        int64_t lab = a.labels[a.pos ? a.pos[i] : i];
        lab = lab < 0 ? 0 : (lab >= a.V ? a.V - 1 : lab);
        atomicAdd(a.loss_sum, lse - x[lab]);
    }
}
// dlogits[i][c] = (softmax - onehot) * g / count  (bf16, pad columns [V, ld) = 0)
__global__ __launch_bounds__(256) void xent_bwd_kernel(vk_xent_args a, uint16_t* dlogits, int ldd, const float* gscale) {
    const int i = blockIdx.x;
    const int n = a.count ? min(*a.count, a.max_rows) : a.max_rows;
    if (i >= n) return;
    const float* x = a.logits + (size_t)i * a.ld;
    const float lse = a.lse[i];
    const int lab = (int)a.labels[a.pos ? a.pos[i] : i];
    const float g = *gscale / (float)n;
    uint16_t* d = dlogits + (size_t)i * ldd;
    const bool vec = ((((uintptr_t)x) & 15) | (((uintptr_t)d) & 7)) == 0 && a.ld >= ((ldd + 3) & ~3);      // the logit row covers the padded width
    const int n4 = vec ? ldd >> 2 : 0;
    for (int c4 = threadIdx.x; c4 < n4; c4 += 256) {
        const int c = 4 * c4;
        const f32x4 v = *(const f32x4*)(x + c);
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = c + r < a.V ? (__expf(v[r] - lse) - (c + r == lab ? 1.f : 0.f)) * g : 0.f;
        *(u32x2*)(d + c) = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
    }
    for (int c = 4 * n4 + threadIdx.x; c < ldd; c += 256) {
        float v = 0.f;
        if (c < a.V) v = (__expf(x[c] - lse) - (c == lab ? 1.f : 0.f)) * g;
        d[c] = f2bf(v);
    }
}

// ---- kl_1601 ---------------------------------------------------------------------------------------
// row i: target = cls[pos[i]] ; loss_sum += sum_c t (log t - logp) ; saves lse and sum_c t
__global__ __launch_bounds__(256) void kl_fwd_kernel(vk_kl_args a) {
    __shared__ float sh[4];
    const int i = blockIdx.x;
    const int n = min(*a.count, a.max_rows);
    if (i >= n) return;
    const float* x = a.logits + (size_t)i * a.ld;
    const float* t = a.target + (size_t)a.pos[i] * a.V;
    // one pass: running (max, sum of exp) of the logits, sum_c t (log t - x) and sum_c t; KL = sum_c t (log t - x) + lse * sum_c t
    float m = -INFINITY, s = 0.f, acc = 0.f, ts = 0.f;
    for (int c = threadIdx.x; c < a.V; c += 256) {
        const float xv = x[c], tv = t[c];
        online_add(xv, m, s);
        ts += tv;
        if (tv > 0.f) acc += tv * (__logf(tv) - xv);
    }
    const float mx = block_max(m, sh);
    s = block_sum(m == -INFINITY ? 0.f : s * __expf(m - mx), sh);
    const float lse = mx + __logf(s);
    acc = block_sum(acc, sh);
    ts = block_sum(ts, sh);
    if (threadIdx.x == 0) { a.lse[i] = lse; a.tsum[i] = ts; atomicAdd(a.loss_sum, a.weight * (acc + lse * ts)); }      // weighted: several targets share the accumulator
}
__global__ __launch_bounds__(256) void kl_bwd_kernel(vk_kl_args a, uint16_t* dlogits, int ldd, const float* gscale) {
    const int i = blockIdx.x;
    const int n = min(*a.count, a.max_rows);
    if (i >= n) return;
    const float* x = a.logits + (size_t)i * a.ld;
    const float* t = a.target + (size_t)a.pos[i] * a.V;
    const float lse = a.lse[i], ts = a.tsum[i];
    const float g = *gscale * a.weight / (float)(n > 1 ? n : 1);
    uint16_t* d = dlogits + (size_t)i * ldd;
    for (int c = threadIdx.x; c < ldd; c += 256) {
        float v = 0.f;
        if (c < a.V) v = (__expf(x[c] - lse) * ts - t[c]) * g;
        d[c] = f2bf(v);
    }
}

// losses[0] = lm_sum / n_t (NaN when no labelled token, like the reference's mean over an empty set);
// losses[1] = w * region_sum / max(n_v, 1) (region_sum already carries the targets' weights: w = 1) ; losses[2] = itm_sum / B
__global__ void loss_finalize_kernel(const float* sums, const int32_t* n_t, const int32_t* n_v, int B, float w, float* losses) {
    if (threadIdx.x == 0) {
        losses[0] = sums[0] / (float)(*n_t);
        const int nv = *n_v;
        losses[1] = w * sums[1] / (float)(nv > 1 ? nv : 1);
        losses[2] = sums[2] / (float)B;
    }
}

// ---- pooled = dropout(pt * pv) ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_mul_fwd_kernel(const uint16_t* pt, const uint16_t* pv, uint16_t* out, int B, int P, vk_dropout dc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * P) return;
    const int row = i / P, c = i - row * P;
    float keep = 1.f;
    if (dc.threshold) {
        const uint64_t seed = *dc.seed;
        const u32x4 w = philox4((uint32_t)(c >> 2), (uint32_t)row, dc.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        keep = (w[c & 3] >= dc.threshold) ? dc.scale : 0.f;
    }
    out[i] = f2bf(bf2f(pt[i]) * bf2f(pv[i]) * keep);
}
// dyt = dp * keep * pv * [pt > 0] ; dyv = dp * keep * pt * [pv > 0]   (pt, pv are post-ReLU)
__global__ __launch_bounds__(256) void pool_mul_bwd_kernel(const uint16_t* dp, int ldp, const uint16_t* pt, const uint16_t* pv, uint16_t* dyt, uint16_t* dyv,
                                                           int B, int P, vk_dropout dc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * P) return;
    const int row = i / P, c = i - row * P;
    float keep = 1.f;
    if (dc.threshold) {
        const uint64_t seed = *dc.seed;
        const u32x4 w = philox4((uint32_t)(c >> 2), (uint32_t)row, dc.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        keep = (w[c & 3] >= dc.threshold) ? dc.scale : 0.f;
    }
    const float g = bf2f(dp[(size_t)row * ldp + c]) * keep, a = bf2f(pt[i]), b = bf2f(pv[i]);
    dyt[i] = f2bf(a > 0.f ? g * b : 0.f);
    dyv[i] = f2bf(b > 0.f ? g * a : 0.f);
}

__global__ void mask_prep_kernel(const int64_t* m, float* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (1.0f - (float)m[i]) * -10000.0f;
}

}  // namespace vk

using namespace vk;

extern "C" int vk_select_rows(const int64_t* labels, int N, int mode, int inner, int outer, int off, int32_t* rows, int32_t* pos,
                              int32_t* count, vk_stream_t s) {
    if (N < 0 || inner <= 0) return set_error("vk_select_rows: bad arguments");
    hipLaunchKernelGGL(select_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, labels, N, mode, inner, outer, off, rows, pos, count);
    return check_launch("vk_select_rows");
}
extern "C" int vk_gather_rows(const void* src, const int32_t* rows, const int32_t* count, void* dst, int H, int max_rows, vk_stream_t s) {
    if (H % 8) return set_error("vk_gather_rows: H %% 8 != 0");
    if (max_rows <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((max_rows + 3) / 4), dim3(256), 0, (hipStream_t)s, (const uint16_t*)src, rows, count, (uint16_t*)dst, H, max_rows);
    return check_launch("vk_gather_rows");
}
extern "C" int vk_scatter_rows_add(const void* src, const int32_t* rows, const int32_t* count, void* dst, int H, int max_rows, vk_stream_t s) {
    if (H % 4) return set_error("vk_scatter_rows_add: H %% 4 != 0");
    if (max_rows <= 0) return 0;
    hipLaunchKernelGGL(scatter_rows_add_kernel, dim3((max_rows + 3) / 4), dim3(256), 0, (hipStream_t)s, (const uint16_t*)src, rows, count, (uint16_t*)dst, H, max_rows);
    return check_launch("vk_scatter_rows_add");
}
extern "C" int vk_xent_fwd(const vk_xent_args* a, vk_stream_t s) {
    if (a->max_rows <= 0) return 0;
    hipLaunchKernelGGL(xent_fwd_kernel, dim3(a->max_rows), dim3(256), 0, (hipStream_t)s, *a);
    return check_launch("vk_xent_fwd");
}
extern "C" int vk_xent_bwd(const vk_xent_args* a, void* dlogits, int ldd, const float* gscale, vk_stream_t s) {
    if (a->max_rows <= 0) return 0;
    if (ldd < a->V) return set_error("vk_xent_bwd: ldd < V");
    hipLaunchKernelGGL(xent_bwd_kernel, dim3(a->max_rows), dim3(256), 0, (hipStream_t)s, *a, (uint16_t*)dlogits, ldd, gscale);
    return check_launch("vk_xent_bwd");
}
extern "C" int vk_kl_fwd(const vk_kl_args* a, vk_stream_t s) {
    if (a->max_rows <= 0) return 0;
    hipLaunchKernelGGL(kl_fwd_kernel, dim3(a->max_rows), dim3(256), 0, (hipStream_t)s, *a);
    return check_launch("vk_kl_fwd");
}
extern "C" int vk_kl_bwd(const vk_kl_args* a, void* dlogits, int ldd, const float* gscale, vk_stream_t s) {
    if (a->max_rows <= 0) return 0;
    if (ldd < a->V) return set_error("vk_kl_bwd: ldd < V");
    hipLaunchKernelGGL(kl_bwd_kernel, dim3(a->max_rows), dim3(256), 0, (hipStream_t)s, *a, (uint16_t*)dlogits, ldd, gscale);
    return check_launch("vk_kl_bwd");
}
extern "C" int vk_loss_finalize(const float* sums, const int32_t* n_t, const int32_t* n_v, int B, float kl_weight, float* losses, vk_stream_t s) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, sums, n_t, n_v, B, kl_weight, losses);
    return check_launch("vk_loss_finalize");
}
extern "C" int vk_pool_mul_fwd(const void* pt, const void* pv, void* out, int B, int P, vk_dropout drop, vk_stream_t s) {
    hipLaunchKernelGGL(pool_mul_fwd_kernel, dim3((B * P + 255) / 256), dim3(256), 0, (hipStream_t)s, (const uint16_t*)pt, (const uint16_t*)pv, (uint16_t*)out, B, P, drop);
    return check_launch("vk_pool_mul_fwd");
}
extern "C" int vk_pool_mul_bwd(const void* dp, int ldp, const void* pt, const void* pv, void* dyt, void* dyv, int B, int P, vk_dropout drop, vk_stream_t s) {
    hipLaunchKernelGGL(pool_mul_bwd_kernel, dim3((B * P + 255) / 256), dim3(256), 0, (hipStream_t)s, (const uint16_t*)dp, ldp, (const uint16_t*)pt,
                       (const uint16_t*)pv, (uint16_t*)dyt, (uint16_t*)dyv, B, P, drop);
    return check_launch("vk_pool_mul_bwd");
}
extern "C" int vk_mask_prep(const int64_t* mask, float* out, int n, vk_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(mask_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)s, mask, out, n);
    return check_launch("vk_mask_prep");
}
