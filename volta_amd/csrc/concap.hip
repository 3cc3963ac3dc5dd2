// ConceptCap batch producer on the device: raw per-pair records -> the model's input tensors, with the reference's sampling policy
// (volta/datasets/concept_cap_dataset.py): caption swap (random_cap :505-522), token masking 15 % / 80-10-10 (random_word :612-641),
// region masking 15 % / 90 % zeroing with IoU > 0.4 co-masking (random_region :643-668, iou :31-68), box normalisation + area
// (__call__ :443-462), zero padding, [CLS] / [SEP] framing (convert_example_to_features :546-610), global feature row = sum of the
// (masked) features / number of un-co-masked rows (ConceptCapLoaderTrain.__iter__ :229-286) and the objective-1 relabel of
// train_concap.py:279-284.  At the step rate of this engine the reference's Python / tensorpack loader (≈14 GB/s of fp32 features and
// targets per GPU) is the first bottleneck; this is HBM-bound byte shuffling: 8 B per feature element (one read, one write).
// Every random decision is word 0 of philox(counter = (slot, pair, stream, 0), key = seed), streams: 0 token draw, 1 replacement id,
// 2 region draw, 3 caption swap (slot 0) and replacement caption (slot 1) -- replayed by oracle/volta_ref.py:concap_words.
#include "common.h"
#include "../../include/volta_hip.h"
#include "util.h"

namespace vk {

constexpr uint32_t CC_T15 = 644245094u;      // floor(0.15 * 2^32): "prob < 0.15"
__device__ __forceinline__ uint32_t cc_word(uint64_t seed, uint32_t stream, uint32_t pair, uint32_t slot) {
    return philox4_rounds<10>(slot, pair, stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32))[0];
}
// the reference compares prob / 0.15 with 0.8 / 0.9 in double precision on prob = word / 2^32
__device__ __forceinline__ double cc_sub(uint32_t w) { return ((double)w / 4294967296.0) / 0.15; }
__device__ __forceinline__ bool cc_swap(const vk_concap_args& a, int b, int& cap) {
    cap = a.cap_index[b];
    if (a.objective != 2 && !a.visualization && (double)cc_word(a.seed, 3, b, 0) / 4294967296.0 > 0.5) {
        cap = (int)(cc_word(a.seed, 3, b, 1) % (uint32_t)a.n_caps);
        return true;
    }
    return false;
}
__device__ __forceinline__ float cc_iou(const float* bi, const float* bj) {
    const float ai = (bi[2] - bi[0] + 1.f) * (bi[3] - bi[1] + 1.f), aj = (bj[2] - bj[0] + 1.f) * (bj[3] - bj[1] + 1.f);
    float iw = fminf(bi[2], bj[2]) - fmaxf(bi[0], bj[0]) + 1.f, ih = fminf(bi[3], bj[3]) - fmaxf(bi[1], bj[1]) + 1.f;
    iw = fmaxf(iw, 0.f); ih = fmaxf(ih, 0.f);
    return iw * ih / (ai + aj - iw * ih);
}

// region decisions of pair b into LDS: sel[r] (label 1), zero[r] (feature row zeroed), returns the global-feature divisor
__device__ __forceinline__ float cc_regions(const vk_concap_args& a, int b, int n, uint8_t* sel, uint8_t* zero, int* cnt_s) {
    const int R = a.R;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const uint32_t w = r < n ? cc_word(a.seed, 2, b, r) : 0xFFFFFFFFu;
        const bool s = r < n && w < CC_T15 && !a.visualization;
        sel[r] = s;
        zero[r] = s && cc_sub(w) < 0.9;
    }
    if (threadIdx.x == 0) *cnt_s = 0;
    __syncthreads();
    int local = 0;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {          // masked_label[r] = OR_i sel[i] && IoU(i, r) > 0.4 (rows >= n: IoU 0)
        bool m = false;
        if (r < n)
            for (int i = 0; i < n; ++i)
                if (sel[i] && cc_iou(a.boxes + ((size_t)b * R + i) * 4, a.boxes + ((size_t)b * R + r) * 4) > 0.4f) m = true;
        local += m ? 0 : 1;
    }
    atomicAdd(cnt_s, local);
    __syncthreads();
    const int c = *cnt_s;
    return (float)(c == 0 ? 1 : c);
}

// ---- text, labels, masks, locations: one workgroup per pair
__global__ __launch_bounds__(128) void concap_meta_kernel(const vk_concap_args a) {
    __shared__ uint8_t sel[256], zero[256];
    __shared__ int cnt_s;
    const int b = blockIdx.x, T = a.T, R = a.R;
    const int n = min(max(a.num_boxes[b], 0), R);
    int cap;
    const bool swapped = cc_swap(a, b, cap);
    const bool drop_labels = a.objective == 1 && swapped;        // train_concap.py:279-284
    const int len = min(a.cap_len[cap], T - 2);
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        int64_t id = 0, lab = -1, m = 0;
        if (t == 0) { id = a.cls_id; m = 1; }
        else if (t == len + 1) { id = a.sep_id; m = 1; }
        else if (t <= len) {
            const int tok = a.cap_tokens[(size_t)cap * a.cap_ld + (t - 1)];
            const uint32_t w = cc_word(a.seed, 0, b, t - 1);
            id = tok; m = 1;
            if (w < CC_T15 && !a.visualization) {
                const double p = cc_sub(w);
                if (p < 0.8) id = a.mask_id;
                else if (p < 0.9) id = (int64_t)(cc_word(a.seed, 1, b, t - 1) % (uint32_t)a.vocab_size);
                lab = tok;
            }
        }
        if (a.objective == 1 && (drop_labels || lab == 0)) lab = -1;      // the relabel multiplies by (is_match == 0) and maps 0 to -1
        const size_t o = (size_t)b * T + t;
        a.input_ids[o] = id; a.input_mask[o] = m; a.segment_ids[o] = 0; a.lm_label_ids[o] = lab;
    }
    if (threadIdx.x == 0) a.is_match[b] = swapped ? 1 : 0;
    (void)cc_regions(a, b, n, sel, zero, &cnt_s);
    const int Rv = R + (a.add_global ? 1 : 0), off = a.add_global == 1 ? 1 : 0;
    const float w = a.img_wh[2 * b], h = a.img_wh[2 * b + 1];
    const float wh = (float)((double)w * (double)h);
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        a.image_label[(size_t)b * R + r] = (r < n && sel[r] && !drop_labels) ? 1 : -1;
        a.image_mask[(size_t)b * Rv + r + off] = r < n ? 1 : 0;
        float* loc = a.image_loc + ((size_t)b * Rv + r + off) * 5;
        if (r < n) {
            const float* bx = a.boxes + ((size_t)b * R + r) * 4;
            loc[0] = bx[0] / w; loc[1] = bx[1] / h; loc[2] = bx[2] / w; loc[3] = bx[3] / h;
            loc[4] = (bx[3] - bx[1]) * (bx[2] - bx[0]) / wh;
        } else {
            loc[0] = loc[1] = loc[2] = loc[3] = loc[4] = 0.f;
        }
    }
    if (a.add_global && threadIdx.x == 0) {
        const int g = a.add_global == 1 ? 0 : R;
        float* loc = a.image_loc + ((size_t)b * Rv + g) * 5;
        loc[0] = 0.f; loc[1] = 0.f; loc[2] = 1.f; loc[3] = 1.f; loc[4] = 1.f;
        a.image_mask[(size_t)b * Rv + g] = 1;
    }
}

// ---- features: one workgroup per (pair, 256 columns); a thread owns one column of all rows (coalesced along the feature axis)
__global__ __launch_bounds__(256) void concap_feat_kernel(const vk_concap_args a) {
    __shared__ uint8_t sel[256], zero[256];
    __shared__ int cnt_s;
    const int b = blockIdx.x, R = a.R, F = a.F;
    const int n = min(max(a.num_boxes[b], 0), R);
    const float cnt = cc_regions(a, b, n, sel, zero, &cnt_s);
    const int Rv = R + (a.add_global ? 1 : 0), off = a.add_global == 1 ? 1 : 0;
    const int f = blockIdx.y * 256 + threadIdx.x;
    if (f >= F) return;
    float sum = 0.f;
    for (int r = 0; r < R; ++r) {
        const float v = (r < n && !zero[r]) ? a.feat[((size_t)b * R + r) * F + f] : 0.f;
        a.image_feat[((size_t)b * Rv + r + off) * F + f] = v;
        sum += v;
    }
    if (a.add_global) a.image_feat[((size_t)b * Rv + (a.add_global == 1 ? 0 : R)) * F + f] = sum / cnt;
}

// ---- class distributions: copy with zero padding of the rows beyond num_boxes
__global__ __launch_bounds__(256) void concap_cls_kernel(const vk_concap_args a) {
    const size_t total = (size_t)a.B * a.R * a.C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / a.C;
        const int b = (int)(row / a.R), r = (int)(row % a.R);
        a.image_cls[i] = r < a.num_boxes[b] ? a.cls[i] : 0.f;
    }
}

}  // namespace vk

extern "C" int vk_concap_batch(const vk_concap_args* a, vk_stream_t stream) {
    using namespace vk;
    if (a->B <= 0) return 0;
    if (a->T < 3 || a->R < 1 || a->R > 256 || a->F < 1 || a->C < 1 || a->n_caps < 1 || a->vocab_size < 1)
        return set_error("vk_concap_batch: bad geometry T=%d R=%d F=%d C=%d n_caps=%d", a->T, a->R, a->F, a->C, a->n_caps);
    if (a->add_global < 0 || a->add_global > 2 || a->objective < 0 || a->objective > 2) return set_error("vk_concap_batch: bad add_global / objective");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(concap_meta_kernel, dim3(a->B), dim3(128), 0, s, *a);
    hipLaunchKernelGGL(concap_feat_kernel, dim3(a->B, (a->F + 255) / 256), dim3(256), 0, s, *a);
    const size_t total = (size_t)a->B * a->R * a->C;
    hipLaunchKernelGGL(concap_cls_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, s, *a);
    return check_launch("vk_concap_batch");
}
