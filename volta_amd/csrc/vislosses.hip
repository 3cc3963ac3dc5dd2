// The visual targets of the region head other than kl_1601, the pooled-vector fusions other than "mul", and the two small
// index kernels of VL-BERT's heads -- all on LABELLED ROWS ONLY, like heads.hip:
//   * mse_2048 / huber_2048 (volta/losses.py:25-33,105-113): regression of the region's input feature
//   * xent_1600 / xent_400 / xent_1601 (losses.py:83-102,116-124): hard detector labels, optionally x detector confidence
//   * nce_2048 (losses.py:36-80): softmax over <sample, prediction> for [own feature, 89 regions of other images, 38 of the same image]
//   * pooled = dropout(pt + pv | pt * pv | pt) (encoders.py:766-774) with the backward through the poolers' ReLUs
//   * the row VLBertTextPooler pools (encoders.py:610-623) and VL-BERT's per-region word index (embeddings.py:262-264)
// Every loss accumulates weight x row loss into *loss_sum; the image loss is loss_sum / max(#masked regions, 1), which is the
// reference's normalisation for each of them (mse / huber divide by #masked x 2048: folded into the row term).
#include "common.h"
#include "../../include/volta_hip.h"
#include "util.h"

namespace vk {

__device__ __forceinline__ float vl_block_sum(float v, float* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float vl_block_max(float v, float* sh) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// ---- regression targets: one workgroup per labelled row -------------------------------------------------------------------
template <bool HUBER>
__global__ __launch_bounds__(256) void vis_reg_fwd_kernel(vk_vis_loss_args a) {
    __shared__ float sh[4];
    const int i = blockIdx.x;
    const int n = min(*a.count, a.max_rows);
    if (i >= n) return;
    const float* x = a.logits + (size_t)i * a.ld;
    const float* t = a.target + (size_t)a.pos[i] * a.V;
    float s = 0.f;
    for (int c = threadIdx.x; c < a.V; c += 256) {
        const float d = x[c] - t[c], ad = fabsf(d);
        s += HUBER ? (ad < 1.f ? 0.5f * d * d : ad - 0.5f) : d * d;
    }
    s = vl_block_sum(s, sh);
    if (threadIdx.x == 0) atomicAdd(a.loss_sum, a.weight * s / (float)a.V);
}
template <bool HUBER>
__global__ __launch_bounds__(256) void vis_reg_bwd_kernel(vk_vis_loss_args a, uint16_t* dlogits, int ldd, const float* gscale) {
    const int i = blockIdx.x;
    const int n = min(*a.count, a.max_rows);
    if (i >= n) return;
    const float* x = a.logits + (size_t)i * a.ld;
    const float* t = a.target + (size_t)a.pos[i] * a.V;
    const float g = *gscale * a.weight / ((float)n * (float)a.V);
    uint16_t* d = dlogits + (size_t)i * ldd;
    for (int c = threadIdx.x; c < ldd; c += 256) {
        float v = 0.f;
        if (c < a.V) {
            const float e = x[c] - t[c];
            v = g * (HUBER ? __builtin_amdgcn_fmed3f(e, -1.f, 1.f) : 2.f * e);
        }
        d[c] = f2bf(v);
    }
}

// ---- hard labels (x confidence) ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vis_xent_fwd_kernel(vk_vis_loss_args a) {
    __shared__ float sh[4];
    const int i = blockIdx.x;
    const int n = min(*a.count, a.max_rows);
    if (i >= n) return;
    const float* x = a.logits + (size_t)i * a.ld;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < a.V; c += 256) mx = fmaxf(mx, x[c]);
    mx = vl_block_max(mx, sh);
    float s = 0.f;
    for (int c = threadIdx.x; c < a.V; c += 256) s += __expf(x[c] - mx);
    s = vl_block_sum(s, sh);
    if (threadIdx.x == 0) {
        const float lse = mx + __logf(s);
        a.lse[i] = lse;
        const int p = a.pos[i];
        int64_t lab = a.labels[p];
        lab = lab < 0 ? 0 : (lab >= a.V ? a.V - 1 : lab);
        atomicAdd(a.loss_sum, a.weight * (a.conf ? a.conf[p] : 1.f) * (lse - x[lab]));
    }
}
__global__ __launch_bounds__(256) void vis_xent_bwd_kernel(vk_vis_loss_args a, uint16_t* dlogits, int ldd, const float* gscale) {
    const int i = blockIdx.x;
    const int n = min(*a.count, a.max_rows);
    if (i >= n) return;
    const float* x = a.logits + (size_t)i * a.ld;
    const float lse = a.lse[i];
    const int p = a.pos[i];
    const int64_t lab = a.labels[p];
    const float g = *gscale * a.weight * (a.conf ? a.conf[p] : 1.f) / (float)n;
    uint16_t* d = dlogits + (size_t)i * ldd;
    for (int c = threadIdx.x; c < ldd; c += 256) {
        float v = 0.f;
        if (c < a.V) v = (__expf(x[c] - lse) - (c == lab ? 1.f : 0.f)) * g;
        d[c] = f2bf(v);
    }
}

// ---- nce_2048 -----------------------------------------------------------------------------------------------------------------
// sample 0 = the region's own feature, samples 1..n_neg = target[neg_index[pos * n_neg + j - 1]]; score_j = <sample_j, prediction>;
// row loss = lse(score) - score_0; the scores are kept in aux[i][0..n_neg] for the backward.
__global__ __launch_bounds__(256) void vis_nce_fwd_kernel(vk_vis_loss_args a) {
    __shared__ float sc[VK_NCE_MAX_SAMPLES];
    __shared__ float sh[4];
    const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = min(*a.count, a.max_rows);
    if (i >= n) return;
    const float* x = a.logits + (size_t)i * a.ld;
    const int p = a.pos[i], ns = a.n_neg + 1;
    for (int j = wave; j < ns; j += 4) {
        const size_t row = j == 0 ? (size_t)p : (size_t)a.neg_index[(size_t)p * a.n_neg + j - 1];
        const float* t = a.target + row * a.V;
        float s = 0.f;
        for (int c = lane * 4; c < a.V; c += 256) {
            const f32x4 tv = *(const f32x4*)(t + c), xv = *(const f32x4*)(x + c);
            s += tv[0] * xv[0] + tv[1] * xv[1] + tv[2] * xv[2] + tv[3] * xv[3];
        }
        s = wave_sum(s);
        if (lane == 0) sc[j] = s;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < ns; j += 256) mx = fmaxf(mx, sc[j]);
    mx = vl_block_max(mx, sh);
    float s = 0.f;
    for (int j = threadIdx.x; j < ns; j += 256) s += __expf(sc[j] - mx);
    s = vl_block_sum(s, sh);
    const float lse = mx + __logf(s);
    for (int j = threadIdx.x; j < ns; j += 256) a.aux[(size_t)i * VK_NCE_MAX_SAMPLES + j] = sc[j];
    if (threadIdx.x == 0) { a.lse[i] = lse; atomicAdd(a.loss_sum, a.weight * (lse - sc[0])); }
}
// d prediction[c] = g * sum_j (softmax_j - [j == 0]) sample_j[c]
__global__ __launch_bounds__(256) void vis_nce_bwd_kernel(vk_vis_loss_args a, uint16_t* dlogits, int ldd, const float* gscale) {
    __shared__ float pj[VK_NCE_MAX_SAMPLES];
    __shared__ int rowj[VK_NCE_MAX_SAMPLES];
    const int i = blockIdx.x;
    const int n = min(*a.count, a.max_rows);
    if (i >= n) return;
    const int p = a.pos[i], ns = a.n_neg + 1;
    const float lse = a.lse[i], g = *gscale * a.weight / (float)n;
    for (int j = threadIdx.x; j < ns; j += 256) {
        pj[j] = (__expf(a.aux[(size_t)i * VK_NCE_MAX_SAMPLES + j] - lse) - (j == 0 ? 1.f : 0.f)) * g;
        rowj[j] = j == 0 ? p : a.neg_index[(size_t)p * a.n_neg + j - 1];
    }
    __syncthreads();
    uint16_t* d = dlogits + (size_t)i * ldd;
    for (int c = threadIdx.x; c < ldd; c += 256) {
        float v = 0.f;
        if (c < a.V)
            for (int j = 0; j < ns; ++j) v += pj[j] * a.target[(size_t)rowj[j] * a.V + c];
        d[c] = f2bf(v);
    }
}

// Negatives of region (b, r), losses.py:47-69: word k = Philox-4x32-10 word (k & 3) at counter (k >> 2, b * R + r, site, 0);
// k in [0, 89): image = word % (B - 1), moved to B - 1 when it hits b;  k in [89, 178): region = word % R  (an "across" negative);
// k in [178, 216): region = word % (R - 1), moved to R - 1 when it hits r, in image b (an "inside" negative).
__global__ __launch_bounds__(128) void nce_negatives_kernel(vk_dropout rng, int B, int R, int32_t* out) {
    const int br = blockIdx.x, b = br / R, r = br - b * R, j = threadIdx.x;
    if (j >= VK_NCE_ACROSS + VK_NCE_INSIDE) return;
    const uint64_t seed = *rng.seed;
    auto word = [&](int k) {
        const u32x4 w = philox4_rounds<10>((uint32_t)(k >> 2), (uint32_t)br, rng.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        return w[k & 3];
    };
    int idx;
    if (j < VK_NCE_ACROSS) {
        int row = (int)(word(j) % (uint32_t)max(B - 1, 1));
        if (row == b) row = B - 1;
        idx = row * R + (int)(word(VK_NCE_ACROSS + j) % (uint32_t)R);
    } else {
        int col = (int)(word(VK_NCE_ACROSS + j) % (uint32_t)max(R - 1, 1));
        if (col == r) col = R - 1;
        idx = b * R + col;
    }
    out[(size_t)br * (VK_NCE_ACROSS + VK_NCE_INSIDE) + j] = idx;
}

// ---- pooled-vector fusion --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_fuse_fwd_kernel(const uint16_t* pt, const uint16_t* pv, uint16_t* out, int B, int P, int mode, vk_dropout dc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * P) return;
    const int row = i / P, c = i - row * P;
    float keep = 1.f;
    if (dc.threshold) {
        const uint64_t seed = *dc.seed;
        const u32x4 w = philox4((uint32_t)(c >> 2), (uint32_t)row, dc.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        keep = (w[c & 3] >= dc.threshold) ? dc.scale : 0.f;
    }
    const float a = bf2f(pt[i]);
    const float v = mode == VK_FUSE_TEXT ? a : (mode == VK_FUSE_SUM ? a + bf2f(pv[i]) : a * bf2f(pv[i]));
    out[i] = f2bf(v * keep);
}
// through the dropout, the fusion and the two poolers' ReLUs (pt, pv are post-ReLU)
__global__ __launch_bounds__(256) void pool_fuse_bwd_kernel(const uint16_t* dp, int ldp, const uint16_t* pt, const uint16_t* pv, uint16_t* dyt, uint16_t* dyv,
                                                            int B, int P, int mode, vk_dropout dc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * P) return;
    const int row = i / P, c = i - row * P;
    float keep = 1.f;
    if (dc.threshold) {
        const uint64_t seed = *dc.seed;
        const u32x4 w = philox4((uint32_t)(c >> 2), (uint32_t)row, dc.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        keep = (w[c & 3] >= dc.threshold) ? dc.scale : 0.f;
    }
    const float g = bf2f(dp[(size_t)row * ldp + c]) * keep, a = bf2f(pt[i]);
    if (mode == VK_FUSE_TEXT) { dyt[i] = f2bf(a > 0.f ? g : 0.f); return; }
    const float b = bf2f(pv[i]);
    dyt[i] = f2bf(a > 0.f ? (mode == VK_FUSE_SUM ? g : g * b) : 0.f);
    dyv[i] = f2bf(b > 0.f ? (mode == VK_FUSE_SUM ? g : g * a) : 0.f);
}

// rows[b] = b * T + max(#non-zero ids of caption b - 2, 0): the token VLBertTextPooler pools; *count = B
__global__ __launch_bounds__(64) void text_end_rows_kernel(const int64_t* ids, int B, int T, int32_t* rows, int32_t* count) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int c = 0;
    for (int t = lane; t < T; t += 64) c += ids[(size_t)b * T + t] != 0;
    c = (int)wave_sum((float)c);
    if (lane == 0) {
        rows[b] = b * T + max(c - 2, 0);
        if (b == 0) *count = B;
    }
}

// VL-BERT's word of region row m (K regions per sample): 1 = END (last region), 2 = masked region (all-zero feature), 0 = object
__global__ void vlbert_obj_ids_kernel(const int32_t* zero_flag, int64_t* ids, int M, int K) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m < M) ids[m] = (m % K == K - 1) ? 1 : (zero_flag[m] ? 2 : 0);
}

// VL-BERT's position ids (volta/embeddings.py:278-292): text_end[b] = number of non-pad tokens.  Text position t is t, plus K wherever
// ANY sample has t >= its text_end (the reference writes the shift through a stride-0 expanded view, so it lands in the one row every
// sample shares): that is t >= min_b text_end[b].  Every box of sample b sits at text_end[b], the last one at text_end[b] + 1.
// One work-group: B * (2T + K) integer operations.
__global__ __launch_bounds__(1024) void vlbert_positions_kernel(const int64_t* ids, int B, int T, int K, int64_t* tpos, int64_t* opos) {
    __shared__ int s_min[16];
    const int tid = threadIdx.x;
    int lo = T;
    for (int b = tid; b < B; b += 1024) {
        int c = 0;
        for (int t = 0; t < T; ++t) c += ids[(size_t)b * T + t] != 0;
        lo = min(lo, c);
        for (int k = 0; k < K; ++k) opos[(size_t)b * K + k] = c + (k == K - 1);
    }
    for (int o = 32; o; o >>= 1) lo = min(lo, __shfl_xor(lo, o));
    if ((tid & 63) == 0) s_min[tid >> 6] = lo;
    __syncthreads();
    lo = s_min[0];
    for (int w = 1; w < 16; ++w) lo = min(lo, s_min[w]);
    for (int i = tid; i < B * T; i += 1024) {
        const int t = i % T;
        tpos[i] = t + (t >= lo ? K : 0);
    }
}

}  // namespace vk

using namespace vk;

static int check_vis(const vk_vis_loss_args* a, const char* who) {
    if (a->kind != VK_VIS_MSE && a->kind != VK_VIS_NCE && a->kind != VK_VIS_XENT && a->kind != VK_VIS_HUBER) return set_error("%s: unknown kind %d", who, a->kind);
    if (!a->logits || !a->pos || !a->count || !a->loss_sum || a->ld < a->V) return set_error("%s: missing logits / pos / count / loss_sum, or ld < V", who);
    if (a->kind == VK_VIS_XENT && (!a->labels || !a->lse)) return set_error("%s: xent needs labels and lse", who);
    if ((a->kind == VK_VIS_MSE || a->kind == VK_VIS_HUBER || a->kind == VK_VIS_NCE) && !a->target) return set_error("%s: regression / nce need the feature matrix", who);
    if (a->kind == VK_VIS_NCE && (!a->neg_index || !a->aux || !a->lse || a->n_neg < 1 || a->n_neg + 1 > VK_NCE_MAX_SAMPLES || a->V % 4 || a->ld % 4))
        return set_error("%s: nce needs neg_index, aux, lse, 1 <= n_neg < %d and V, ld multiples of 4", who, VK_NCE_MAX_SAMPLES);
    return 0;
}

extern "C" int vk_vis_loss_fwd(const vk_vis_loss_args* a, vk_stream_t s) {
    if (check_vis(a, "vk_vis_loss_fwd")) return -1;
    if (a->max_rows <= 0) return 0;
    const dim3 grid(a->max_rows), block(256);
    hipStream_t st = (hipStream_t)s;
    switch (a->kind) {
        case VK_VIS_MSE: hipLaunchKernelGGL(vis_reg_fwd_kernel<false>, grid, block, 0, st, *a); break;
        case VK_VIS_HUBER: hipLaunchKernelGGL(vis_reg_fwd_kernel<true>, grid, block, 0, st, *a); break;
        case VK_VIS_XENT: hipLaunchKernelGGL(vis_xent_fwd_kernel, grid, block, 0, st, *a); break;
        default: hipLaunchKernelGGL(vis_nce_fwd_kernel, grid, block, 0, st, *a); break;
    }
    return check_launch("vk_vis_loss_fwd");
}

extern "C" int vk_vis_loss_bwd(const vk_vis_loss_args* a, void* dlogits, int ldd, const float* gscale, vk_stream_t s) {
    if (check_vis(a, "vk_vis_loss_bwd")) return -1;
    if (ldd < a->V) return set_error("vk_vis_loss_bwd: ldd < V");
    if (a->max_rows <= 0) return 0;
    const dim3 grid(a->max_rows), block(256);
    hipStream_t st = (hipStream_t)s;
    uint16_t* d = (uint16_t*)dlogits;
    switch (a->kind) {
        case VK_VIS_MSE: hipLaunchKernelGGL(vis_reg_bwd_kernel<false>, grid, block, 0, st, *a, d, ldd, gscale); break;
        case VK_VIS_HUBER: hipLaunchKernelGGL(vis_reg_bwd_kernel<true>, grid, block, 0, st, *a, d, ldd, gscale); break;
        case VK_VIS_XENT: hipLaunchKernelGGL(vis_xent_bwd_kernel, grid, block, 0, st, *a, d, ldd, gscale); break;
        default: hipLaunchKernelGGL(vis_nce_bwd_kernel, grid, block, 0, st, *a, d, ldd, gscale); break;
    }
    return check_launch("vk_vis_loss_bwd");
}

extern "C" int vk_nce_negatives(vk_dropout rng, int B, int R, int32_t* out, vk_stream_t s) {
    if (!rng.seed || B < 2 || R < 2) return set_error("vk_nce_negatives: needs a seed word, B >= 2 and R >= 2 (losses.py:50,59 draw from [0, B-1) and [0, R-1))");
    hipLaunchKernelGGL(nce_negatives_kernel, dim3(B * R), dim3(128), 0, (hipStream_t)s, rng, B, R, out);
    return check_launch("vk_nce_negatives");
}

extern "C" int vk_pool_fuse_fwd(const void* pt, const void* pv, void* out, int B, int P, int mode, vk_dropout drop, vk_stream_t s) {
    if (mode < VK_FUSE_MUL || mode > VK_FUSE_TEXT || (mode != VK_FUSE_TEXT && !pv)) return set_error("vk_pool_fuse_fwd: mode %d (mul 0 | sum 1 | text 2; pv required unless text)", mode);
    hipLaunchKernelGGL(pool_fuse_fwd_kernel, dim3((B * P + 255) / 256), dim3(256), 0, (hipStream_t)s, (const uint16_t*)pt, (const uint16_t*)pv, (uint16_t*)out, B, P, mode, drop);
    return check_launch("vk_pool_fuse_fwd");
}
extern "C" int vk_pool_fuse_bwd(const void* dp, int ldp, const void* pt, const void* pv, void* dyt, void* dyv, int B, int P, int mode, vk_dropout drop, vk_stream_t s) {
    if (mode < VK_FUSE_MUL || mode > VK_FUSE_TEXT || (mode != VK_FUSE_TEXT && (!pv || !dyv))) return set_error("vk_pool_fuse_bwd: mode %d (mul 0 | sum 1 | text 2; pv, dyv required unless text)", mode);
    hipLaunchKernelGGL(pool_fuse_bwd_kernel, dim3((B * P + 255) / 256), dim3(256), 0, (hipStream_t)s, (const uint16_t*)dp, ldp, (const uint16_t*)pt,
                       (const uint16_t*)pv, (uint16_t*)dyt, (uint16_t*)dyv, B, P, mode, drop);
    return check_launch("vk_pool_fuse_bwd");
}

extern "C" int vk_text_end_rows(const int64_t* ids, int B, int T, int32_t* rows, int32_t* count, vk_stream_t s) {
    if (B <= 0 || T <= 0) return set_error("vk_text_end_rows: bad arguments");
    hipLaunchKernelGGL(text_end_rows_kernel, dim3(B), dim3(64), 0, (hipStream_t)s, ids, B, T, rows, count);
    return check_launch("vk_text_end_rows");
}

extern "C" int vk_vlbert_obj_ids(const int32_t* zero_flag, int64_t* ids, int M, int K, vk_stream_t s) {
    if (M <= 0 || K <= 0) return set_error("vk_vlbert_obj_ids: bad arguments");
    hipLaunchKernelGGL(vlbert_obj_ids_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)s, zero_flag, ids, M, K);
    return check_launch("vk_vlbert_obj_ids");
}

extern "C" int vk_vlbert_positions(const int64_t* ids, int B, int T, int K, int64_t* tpos, int64_t* opos, vk_stream_t s) {
    if (B <= 0 || T <= 0 || K <= 0 || !ids || !tpos || !opos) return set_error("vk_vlbert_positions: bad arguments");
    hipLaunchKernelGGL(vlbert_positions_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, ids, B, T, K, tpos, opos);
    return check_launch("vk_vlbert_positions");
}
