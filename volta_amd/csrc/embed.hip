// Embedding-side kernels (HBM-bound, one wave per row, 8/16-byte accesses):
//   * text embedding sum  word[ids] + pos[t] + type[seg] (+ extra)        volta/embeddings.py:55-66
//   * its backward: scatter-add into the three tables
//   * the 5-wide box-location linear and its backward                     volta/embeddings.py:135,141
//   * y = dropout((a + b) * scale) and its backward (LXMERT image embedding, embeddings.py:169-170)
//   * column sums of a bf16 matrix (gradient of a broadcast row vector)
#include "common.h"
#include "../../include/volta_hip.h"
#include "util.h"

namespace vk {

__device__ __forceinline__ void ld4(const uint16_t* p, float (&v)[4]) {
    u32x2 r = *(const u32x2*)p;
    v[0] = bf2f(r[0] & 0xFFFF); v[1] = bf2f(r[0] >> 16); v[2] = bf2f(r[1] & 0xFFFF); v[3] = bf2f(r[1] >> 16);
}
__device__ __forceinline__ void st4(uint16_t* p, const float (&v)[4]) {
    *(u32x2*)p = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
}

__device__ __forceinline__ int64_t clampi(int64_t v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

__global__ __launch_bounds__(256) void embed_sum_fwd_kernel(vk_embed_args a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.M) return;
    const int64_t id = clampi(a.ids[row], a.V);
    const int64_t ty = a.type_ids ? clampi(a.type_ids[row], a.n_types) : 0;
    const int64_t ps = clampi(a.pos_ids ? a.pos_ids[row] : (row % a.T), a.P);
    const float* w = a.word + (size_t)id * a.H;
    const float* p = a.pos + (size_t)ps * a.H;
    const float* t = a.type + (size_t)ty * a.H;
    for (int c = lane * 4; c < a.H; c += 256) {
        const f32x4 wv = *(const f32x4*)(w + c), pv = *(const f32x4*)(p + c), tv = *(const f32x4*)(t + c);
        float o[4] = {wv[0] + pv[0] + tv[0], wv[1] + pv[1] + tv[1], wv[2] + pv[2] + tv[2], wv[3] + pv[3] + tv[3]};
        if (a.extra) {
            float e[4];
            ld4((const uint16_t*)a.extra + (size_t)row * a.H + c, e);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] += e[r];
        }
        st4((uint16_t*)a.z + (size_t)row * a.H + c, o);
    }
}

// Word table: float atomics, but only ONE per workgroup, column and distinct id.  A workgroup takes 32 rows of the SAME position t (samples
// b0 .. b0+31: row = b * T + t), because that is where ids repeat -- [CLS] at t = 0 and [SEP] at the caption's end in every sample, [MASK] on
// 15 % of the rest -- and the first row of every id inside the workgroup sums its duplicates before it touches the table.  (One atomic per
// row and column made 256-700 of them queue on the same address for the three hot ids: 110 us for 5120 x 768, on the critical path of the step.)
// Type table (<= 4 rows, every row of the batch hits them): per-workgroup partial sums first, one atomic per workgroup and column afterwards.
template <int NCH>
__global__ __launch_bounds__(256) void embed_scatter_kernel(vk_embed_bwd_args a) {
    __shared__ float red[4][4][NCH * 256];
    __shared__ int64_t id_s[32];
    __shared__ int ty_s[32], row_s[32];
    __shared__ int64_t ps_s[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = a.M / a.T;                                  // samples
    if (threadIdx.x < 32) {
        const int i = blockIdx.x * 32 + (int)threadIdx.x;      // position-major index: i = t * nb + b
        int row = -1;
        if (i < a.M) { const int t = i / nb, b = i - t * nb; row = b * a.T + t; }
        row_s[threadIdx.x] = row;
        id_s[threadIdx.x] = row >= 0 ? clampi(a.ids[row], a.V) : -1;
        ty_s[threadIdx.x] = (row >= 0 && a.type_ids) ? (int)clampi(a.type_ids[row], a.n_types) : 0;
        ps_s[threadIdx.x] = (row >= 0 && a.pos_ids) ? clampi(a.pos_ids[row], a.P) : -1;
    }
    __syncthreads();
    float ta[4][NCH][4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) ta[k][j][r] = 0.f;
    for (int it = 0; it < 8; ++it) {
        const int r0 = it * 4 + wave;
        const int row = row_s[r0];
        if (row < 0) continue;
        const int64_t id = id_s[r0];
        bool first = true;                                     // wave-uniform: is this the workgroup's first row with this id?
        for (int q = 0; q < r0; ++q) first &= id_s[q] != id;
        const int64_t ps = ps_s[r0];
        bool firstp = ps >= 0;                                 // wave-uniform: the workgroup's first row with this position id
        for (int q = 0; q < r0; ++q) firstp &= ps_s[q] != ps;
        if (firstp) {
            // explicit positions (VL-BERT): the 32 rows of a workgroup are 32 samples at ONE sequence position, and VL-BERT gives every region
            // of every sample the same position id -- one atomic per row made 25 600 rows queue on one table row (1.4 ms at 100 regions,
            // profiles/r04_vlbert_r100_kernel_stats.md); the first row of an id sums the workgroup's duplicates, as for the word table
            float pacc[NCH][4];
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) pacc[j][r] = 0.f;
            unsigned long long dupp = __ballot(lane < 32 && lane >= r0 && ps_s[lane & 31] == ps && row_s[lane & 31] >= 0);
            while (dupp) {
                int qs[4], n = 0;
                for (; n < 4 && dupp; ++n) { qs[n] = __builtin_ctzll(dupp); dupp &= dupp - 1; }
                float v[4][NCH][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < n) {
#pragma unroll
                        for (int j = 0; j < NCH; ++j) {
                            const int c = j * 256 + lane * 4;
                            if (c < a.H) ld4((const uint16_t*)a.dz + (size_t)row_s[qs[u]] * a.H + c, v[u][j]);
                        }
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (u < n) {
#pragma unroll
                        for (int j = 0; j < NCH; ++j) {
                            const int c = j * 256 + lane * 4;
                            if (c < a.H) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) pacc[j][r] += v[u][j][r];
                            }
                        }
                    }
            }
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int c = j * 256 + lane * 4;
                if (c < a.H) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) atomicAdd(a.dpos + (size_t)ps * a.H + c + r, pacc[j][r]);
                }
            }
        }
        if (!first) continue;
        float acc[NCH][4];
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
        // this row and its duplicates further down, four rows' loads in flight at a time ([CLS] at t = 0: all 32 rows of the workgroup)
        unsigned long long dup = __ballot(lane < 32 && lane >= r0 && id_s[lane & 31] == id && row_s[lane & 31] >= 0);
        while (dup) {
            int qs[4], n = 0;
            for (; n < 4 && dup; ++n) { qs[n] = __builtin_ctzll(dup); dup &= dup - 1; }
            float v[4][NCH][4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < n) {
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        const int c = j * 256 + lane * 4;
                        if (c < a.H) ld4((const uint16_t*)a.dz + (size_t)row_s[qs[u]] * a.H + c, v[u][j]);
                    }
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u < n) {
                    const int ty = ty_s[qs[u]];
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        const int c = j * 256 + lane * 4;
                        if (c < a.H) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                acc[j][r] += v[u][j][r];
#pragma unroll
                                for (int k = 0; k < 4; ++k) ta[k][j][r] += (ty == k) ? v[u][j][r] : 0.f;
                            }
                        }
                    }
                }
        }
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < a.H) {
#pragma unroll
                for (int r = 0; r < 4; ++r) atomicAdd(a.dword + (size_t)id * a.H + c + r, acc[j][r]);
            }
        }
    }
    if (!a.dtype) return;
    for (int k = 0; k < 4 && k < a.n_types; ++k) {
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][k][j * 256 + lane * 4 + r] = ta[k][j][r];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.n_types * a.H; i += 256) {
        const int k = i / a.H, c = i - k * a.H;
        const float s = red[0][k][c] + red[1][k][c] + red[2][k][c] + red[3][k][c];
        if (s != 0.f) atomicAdd(a.dtype + (size_t)k * a.H + c, s);
    }
}

// dpos[t][c] = sum_b dz[(b*T + t)*H + c]   (implicit positions: t = row % T); overwrites rows [0, T)
__global__ __launch_bounds__(64) void embed_pos_reduce_kernel(const uint16_t* dz, float* dpos, int B, int T, int H) {
    const int t = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x * 4;
    if (c >= H) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int b = 0; b < B; ++b) {
        float v[4];
        ld4(dz + ((size_t)b * T + t) * H + c, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += v[r];
    }
    *(f32x4*)(dpos + (size_t)t * H + c) = f32x4{s[0], s[1], s[2], s[3]};
}

// out[m][n] = sum_k loc[m][k] W[n][k] + b[n],  k < nloc <= 8, all fp32 in, bf16 out.  A lane keeps the weights and biases of its
// 4 * NCH columns in registers and walks LOC_ROWS rows with them (one row per wave re-read the [H, nloc] weight matrix with 72
// strided scalar loads per lane and row: 54 us for 9472 x 768, issue-bound).
constexpr int LOC_ROWS = 8;      // rows per wave
template <int NCH>
__global__ __launch_bounds__(256) void loc_linear_fwd_kernel(const float* __restrict__ loc, const float* __restrict__ W, const float* __restrict__ bias,
                                                             uint16_t* __restrict__ out, int M, int H, int nloc) {
    const int lane = threadIdx.x & 63, row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LOC_ROWS;
    if (row0 >= M) return;
    float w[NCH][4][8], b[NCH][4];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = j * 256 + lane * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            b[j][r] = c < H ? bias[c + r] : 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) w[j][r][k] = (c < H && k < nloc) ? W[(size_t)(c + r) * nloc + k] : 0.f;
        }
    }
    for (int i = 0; i < LOC_ROWS; ++i) {
        const int row = row0 + i;
        if (row >= M) break;
        float l[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) l[k] = k < nloc ? loc[(size_t)row * nloc + k] : 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < H) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float s = b[j][r];
#pragma unroll
                    for (int k = 0; k < 8; ++k) s += l[k] * w[j][r][k];
                    o[r] = s;
                }
                st4(out + (size_t)row * H + c, o);
            }
        }
    }
}

// partial[blk][k][n] = sum over the block's 32 rows of dz[m][n] * loc[m][k] (k < nloc) and of dz[m][n] (k = nloc)
template <int NCH>
__global__ __launch_bounds__(256) void loc_linear_bwd_kernel(const uint16_t* dz, const float* loc, float* partial, int M, int H, int nloc) {
    __shared__ float red[4][NCH * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[9][NCH][4];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[k][j][r] = 0.f;
    for (int it = 0; it < 8; ++it) {
        const int row = blockIdx.x * 32 + it * 4 + wave;
        if (row >= M) break;
        float l[9];
#pragma unroll
        for (int k = 0; k < 8; ++k) l[k] = k < nloc ? loc[(size_t)row * nloc + k] : 0.f;
        l[8] = 1.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < H) {
                float v[4];
                ld4(dz + (size_t)row * H + c, v);
#pragma unroll
                for (int k = 0; k < 9; ++k)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[k][j][r] += v[r] * l[k];
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * 9 * H;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][j * 256 + lane * 4 + r] = acc[k][j][r];
        __syncthreads();
        for (int c = threadIdx.x; c < H; c += 256) out[k * H + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
}
// dW[n][k] = sum_blk partial[blk][k][n] ; db[n] = sum_blk partial[blk][8][n]
__global__ __launch_bounds__(1024) void loc_linear_bwd_finalize_kernel(const float* partial, int nblk, int H, int nloc, float* dW, float* db) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (i < 9 * H)
        for (int b = ty; b < nblk; b += 16) s += partial[(size_t)b * 9 * H + i];
    red[ty][tx] = s;
    __syncthreads();
    if (ty != 0 || i >= 9 * H) return;
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][tx];
    const int k = i / H, n = i - k * H;
    if (k < 8 && k >= nloc) return;
    if (k == 8) db[n] = s; else dW[(size_t)n * nloc + k] = s;
}

// y = dropout((a + b) * scale) ; backward g = dy * keep * scale   (row = m, column = c)
__global__ __launch_bounds__(256) void add_dropout_kernel(const uint16_t* a, const uint16_t* b, uint16_t* y, int M, int H, float scale,
                                                          vk_dropout dc, int bwd) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bool don = dc.threshold != 0;
    const uint64_t seed = don ? *dc.seed : 0;
    for (int c = lane * 4; c < H; c += 256) {
        float v[4];
        ld4(a + (size_t)row * H + c, v);
        if (!bwd && b) {
            float w[4];
            ld4(b + (size_t)row * H + c, w);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += w[r];
        }
        u32x4 wd = {~0u, ~0u, ~0u, ~0u};
        if (don) wd = philox4((uint32_t)(c >> 2), (uint32_t)row, dc.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (wd[r] >= dc.threshold) ? v[r] * scale * dc.scale : 0.f;
        st4(y + (size_t)row * H + c, v);
    }
}

// partial[blk][c] = sum of the block's 32 rows
template <int NCH>
__global__ __launch_bounds__(256) void colsum_kernel(const uint16_t* src, float* partial, int M, int H) {
    __shared__ float red[4][NCH * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc[NCH][4];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < 8; ++it) {
        const int row = blockIdx.x * 32 + it * 4 + wave;
        if (row >= M) break;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < H) {
                float v[4];
                ld4(src + (size_t)row * H + c, v);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][r] += v[r];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][j * 256 + lane * 4 + r] = acc[j][r];
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) partial[(size_t)blockIdx.x * H + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* partial, int nblk, int H, float* out, int accumulate) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (c < H)
        for (int b = ty; b < nblk; b += 16) s += partial[(size_t)b * H + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty != 0 || c >= H) return;
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][tx];
    out[c] = accumulate ? out[c] + s : s;
}

// ---- VL-BERT region input (volta/embeddings.py:102-124, 243-251): one wave per region row builds the bf16 row
// [ sin/cos coordinate embedding (4 x 2*dim) | appearance feature (F) ] of the 2F-wide obj_downsample input:
//   * all-zero feature rows (masked regions) take the learned object_mask_visual_embedding instead (flag saved)
//   * box (x1,y1,x2,y2) -> (xc, yc, w, h) * 100, divided by 1000^(i/dim), sin | cos per coordinate
//   * dropout over the whole row (obj_downsample[0])
__global__ __launch_bounds__(256) void vlbert_prep_kernel(const float* loc, int nloc, const float* feat, const float* mask_emb, uint16_t* out,
                                                          int32_t* zero_flag, int M, int F, int dim, vk_dropout dc) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* f = feat + (size_t)row * F;
    bool nz = false;
    for (int c = lane * 4; c < F; c += 256) {
        const f32x4 v = *(const f32x4*)(f + c);
        nz |= (v[0] != 0.f) | (v[1] != 0.f) | (v[2] != 0.f) | (v[3] != 0.f);
    }
    const bool zero_row = __ballot(nz) == 0ull;
    if (lane == 0) zero_flag[row] = zero_row ? 1 : 0;
    const float x1 = loc[(size_t)row * nloc + 0], y1 = loc[(size_t)row * nloc + 1], x2 = loc[(size_t)row * nloc + 2], y2 = loc[(size_t)row * nloc + 3];
    const float pos[4] = {(x1 + x2) / 2 * 100, (y1 + y2) / 2 * 100, (x2 - x1) * 100, (y2 - y1) * 100};
    const int W = 8 * dim + F;
    const bool don = dc.threshold != 0;
    const uint64_t seed = don ? *dc.seed : 0;
    uint16_t* o = out + (size_t)row * W;
    for (int c = lane * 4; c < W; c += 256) {
        float v[4];
        if (c < 8 * dim) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = c + r, coord = idx / (2 * dim), j = idx % (2 * dim), i = j < dim ? j : j - dim;
                const float ang = pos[coord] / powf(1000.0f, (float)i / (float)dim);
                v[r] = j < dim ? sinf(ang) : cosf(ang);
            }
        } else {
            const int fc = c - 8 * dim;
            const f32x4 fv = zero_row ? *(const f32x4*)(mask_emb + fc) : *(const f32x4*)(f + fc);
            v[0] = fv[0]; v[1] = fv[1]; v[2] = fv[2]; v[3] = fv[3];
        }
        if (don) {
            const u32x4 w = philox4((uint32_t)(c >> 2), (uint32_t)row, dc.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (w[r] >= dc.threshold) ? v[r] * dc.scale : 0.f;
        }
        st4(o + c, v);
    }
}

// gradient of the mask embedding: partial[blk][f] = sum over the block's flagged rows of keep * dx[row][col0 + f]
template <int NCH>
__global__ __launch_bounds__(256) void vlbert_maskgrad_kernel(const uint16_t* dx, int ldx, int col0, const int32_t* zero_flag, float* partial,
                                                              int M, int F, vk_dropout dc) {
    __shared__ float red[4][NCH * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool don = dc.threshold != 0;
    const uint64_t seed = don ? *dc.seed : 0;
    float acc[NCH][4];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[j][r] = 0.f;
    for (int it = 0; it < 8; ++it) {
        const int row = blockIdx.x * 32 + it * 4 + wave;
        if (row >= M) break;
        if (!zero_flag[row]) continue;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int c = j * 256 + lane * 4;
            if (c < F) {
                float v[4];
                ld4(dx + (size_t)row * ldx + col0 + c, v);
                u32x4 w = {~0u, ~0u, ~0u, ~0u};
                if (don) w = philox4((uint32_t)((col0 + c) >> 2), (uint32_t)row, dc.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j][r] += (w[r] >= dc.threshold) ? v[r] * dc.scale : 0.f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][j * 256 + lane * 4 + r] = acc[j][r];
    __syncthreads();
    for (int c = threadIdx.x; c < F; c += 256) partial[(size_t)blockIdx.x * F + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// out[b][c] = sum_{t < T} in[(b*T + t)][c]   (gradient of a per-sample vector broadcast over its T rows)
__global__ __launch_bounds__(64) void rowgroup_sum_kernel(const uint16_t* in, uint16_t* out, int T, int H) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x * 4;
    if (c >= H) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < T; ++t) {
        float v[4];
        ld4(in + ((size_t)b * T + t) * H + c, v);
#pragma unroll
        for (int r = 0; r < 4; ++r) s[r] += v[r];
    }
    st4(out + (size_t)b * H + c, s);
}

// out = dy where y > 0 else 0  (backward of a ReLU whose output y was kept)
__global__ void relu_bwd_kernel(const uint16_t* dy, const uint16_t* y, uint16_t* out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 g = *(const u32x4*)(dy + i * 8), v = *(const u32x4*)(y + i * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = (bf2f(v[k] & 0xFFFF) > 0.f) ? (g[k] & 0xFFFFu) : 0u, hi = (bf2f(v[k] >> 16) > 0.f) ? (g[k] & 0xFFFF0000u) : 0u;
            o[k] = lo | hi;
        }
        *(u32x4*)(out + i * 8) = o;
    }
}

}  // namespace vk

using namespace vk;

extern "C" int vk_vlbert_prep_fwd(const float* loc, int nloc, const float* feat, const float* mask_emb, void* out, int32_t* zero_flag,
                                  int M, int F, int dim, vk_dropout drop, vk_stream_t s) {
    if (F % 4 || (8 * dim) % 4 || nloc < 4) return set_error("vk_vlbert_prep_fwd: F and 8*dim must be multiples of 4, nloc >= 4");
    if (M <= 0) return 0;
    hipLaunchKernelGGL(vlbert_prep_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)s, loc, nloc, feat, mask_emb, (uint16_t*)out, zero_flag, M, F, dim, drop);
    return check_launch("vk_vlbert_prep_fwd");
}

extern "C" int vk_vlbert_maskgrad(const void* dx, int ldx, int col0, const int32_t* zero_flag, float* partial, float* out, int M, int F,
                                  vk_dropout drop, vk_stream_t s) {
    if (F % 4 || F > 2048 || col0 % 4) return set_error("vk_vlbert_maskgrad: F must be a multiple of 4, <= 2048");
    if (M <= 0) return 0;
    const int nblk = (M + 31) / 32, nch = (F + 255) / 256;
    hipStream_t st = (hipStream_t)s;
    if (nch <= 4) hipLaunchKernelGGL(vlbert_maskgrad_kernel<4>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)dx, ldx, col0, zero_flag, partial, M, F, drop);
    else hipLaunchKernelGGL(vlbert_maskgrad_kernel<8>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)dx, ldx, col0, zero_flag, partial, M, F, drop);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((F + 63) / 64), dim3(1024), 0, st, partial, nblk, F, out, 0);
    return check_launch("vk_vlbert_maskgrad");
}

extern "C" int vk_rowgroup_sum_bf16(const void* in, void* out, int B, int T, int H, vk_stream_t s) {
    if (H % 4) return set_error("vk_rowgroup_sum_bf16: H %% 4 != 0");
    if (B <= 0) return 0;
    hipLaunchKernelGGL(rowgroup_sum_kernel, dim3(B, (H + 255) / 256), dim3(64), 0, (hipStream_t)s, (const uint16_t*)in, (uint16_t*)out, T, H);
    return check_launch("vk_rowgroup_sum_bf16");
}

extern "C" int vk_relu_bwd_bf16(const void* dy, const void* y, void* out, int64_t n, vk_stream_t s) {
    if (n % 8) return set_error("vk_relu_bwd_bf16: n %% 8 != 0");
    if (n <= 0) return 0;
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const uint16_t*)dy, (const uint16_t*)y, (uint16_t*)out, (size_t)(n / 8));
    return check_launch("vk_relu_bwd_bf16");
}

extern "C" int vk_copy_async(void* dst, const void* src, int64_t bytes, vk_stream_t s) {
    if (bytes <= 0) return 0;
    hipError_t e = hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)s);
    if (e != hipSuccess) return set_error("vk_copy_async: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int vk_embed_sum_fwd(const vk_embed_args* a, vk_stream_t s) {
    if (a->H % 4) return set_error("vk_embed_sum_fwd: H %% 4 != 0");
    if (a->M <= 0) return 0;
    hipLaunchKernelGGL(embed_sum_fwd_kernel, dim3((a->M + 3) / 4), dim3(256), 0, (hipStream_t)s, *a);
    return check_launch("vk_embed_sum_fwd");
}

extern "C" int vk_embed_sum_bwd(const vk_embed_bwd_args* a, vk_stream_t s) {
    if (a->H % 4 || a->H > 1024) return set_error("vk_embed_sum_bwd: H must be a multiple of 4, <= 1024");
    if (a->n_types > 4) return set_error("vk_embed_sum_bwd: at most 4 token types");
    if (a->M <= 0) return 0;
    if (a->T <= 0 || a->M % a->T) return set_error("vk_embed_sum_bwd: M must be a multiple of T (rows are b * T + t)");
    const int nch = (a->H + 255) / 256;
    dim3 grid((a->M + 31) / 32), block(256);
    hipStream_t st = (hipStream_t)s;
    switch (nch) {
        case 1: hipLaunchKernelGGL(embed_scatter_kernel<1>, grid, block, 0, st, *a); break;
        case 2: hipLaunchKernelGGL(embed_scatter_kernel<2>, grid, block, 0, st, *a); break;
        case 3: hipLaunchKernelGGL(embed_scatter_kernel<3>, grid, block, 0, st, *a); break;
        default: hipLaunchKernelGGL(embed_scatter_kernel<4>, grid, block, 0, st, *a); break;
    }
    if (!a->pos_ids && a->dpos) {
        if (a->M % a->T) return set_error("vk_embed_sum_bwd: M %% T != 0");
        hipLaunchKernelGGL(embed_pos_reduce_kernel, dim3(a->T, (a->H + 255) / 256), dim3(64), 0, st, (const uint16_t*)a->dz, a->dpos,
                           a->M / a->T, a->T, a->H);
    }
    return check_launch("vk_embed_sum_bwd");
}

extern "C" int vk_loc_linear_fwd(const float* loc, const float* W, const float* bias, void* out, int M, int H, int nloc, vk_stream_t s) {
    if (nloc > 8 || H % 4) return set_error("vk_loc_linear_fwd: nloc <= 8, H %% 4 == 0 required");
    if (M <= 0) return 0;
    if (H > 1024) return set_error("vk_loc_linear_fwd: H <= 1024 required");
    const dim3 grid((M + 4 * LOC_ROWS - 1) / (4 * LOC_ROWS)), block(256);
    hipStream_t st = (hipStream_t)s;
    switch ((H + 255) / 256) {
        case 1: hipLaunchKernelGGL(loc_linear_fwd_kernel<1>, grid, block, 0, st, loc, W, bias, (uint16_t*)out, M, H, nloc); break;
        case 2: hipLaunchKernelGGL(loc_linear_fwd_kernel<2>, grid, block, 0, st, loc, W, bias, (uint16_t*)out, M, H, nloc); break;
        case 3: hipLaunchKernelGGL(loc_linear_fwd_kernel<3>, grid, block, 0, st, loc, W, bias, (uint16_t*)out, M, H, nloc); break;
        default: hipLaunchKernelGGL(loc_linear_fwd_kernel<4>, grid, block, 0, st, loc, W, bias, (uint16_t*)out, M, H, nloc); break;
    }
    return check_launch("vk_loc_linear_fwd");
}

extern "C" int vk_rows32(int M) { return (M + 31) / 32; }

extern "C" int vk_loc_linear_bwd(const void* dz, const float* loc, float* partial, float* dW, float* db, int M, int H, int nloc, vk_stream_t s) {
    if (nloc > 8 || H % 4 || H > 1024) return set_error("vk_loc_linear_bwd: nloc <= 8, H %% 4 == 0, H <= 1024 required");
    if (M <= 0) return 0;
    const int nblk = (M + 31) / 32, nch = (H + 255) / 256;
    hipStream_t st = (hipStream_t)s;
    switch (nch) {
        case 1: hipLaunchKernelGGL(loc_linear_bwd_kernel<1>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)dz, loc, partial, M, H, nloc); break;
        case 2: hipLaunchKernelGGL(loc_linear_bwd_kernel<2>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)dz, loc, partial, M, H, nloc); break;
        case 3: hipLaunchKernelGGL(loc_linear_bwd_kernel<3>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)dz, loc, partial, M, H, nloc); break;
        default: hipLaunchKernelGGL(loc_linear_bwd_kernel<4>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)dz, loc, partial, M, H, nloc); break;
    }
    hipLaunchKernelGGL(loc_linear_bwd_finalize_kernel, dim3((9 * H + 63) / 64), dim3(1024), 0, st, partial, nblk, H, nloc, dW, db);
    return check_launch("vk_loc_linear_bwd");
}

extern "C" int vk_add_dropout(const void* a, const void* b, void* y, int M, int H, float scale, vk_dropout drop, int backward, vk_stream_t s) {
    if (H % 4) return set_error("vk_add_dropout: H %% 4 != 0");
    if (M <= 0) return 0;
    hipLaunchKernelGGL(add_dropout_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)s, (const uint16_t*)a, (const uint16_t*)b,
                       (uint16_t*)y, M, H, scale, drop, backward);
    return check_launch("vk_add_dropout");
}

extern "C" int vk_colsum_bf16(const void* src, float* partial, float* out, int M, int H, int accumulate, vk_stream_t s) {
    if (H % 4 || H > 1024) return set_error("vk_colsum_bf16: H must be a multiple of 4, <= 1024");
    if (M <= 0) return 0;
    const int nblk = (M + 31) / 32, nch = (H + 255) / 256;
    hipStream_t st = (hipStream_t)s;
    switch (nch) {
        case 1: hipLaunchKernelGGL(colsum_kernel<1>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)src, partial, M, H); break;
        case 2: hipLaunchKernelGGL(colsum_kernel<2>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)src, partial, M, H); break;
        case 3: hipLaunchKernelGGL(colsum_kernel<3>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)src, partial, M, H); break;
        default: hipLaunchKernelGGL(colsum_kernel<4>, dim3(nblk), dim3(256), 0, st, (const uint16_t*)src, partial, M, H); break;
    }
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((H + 63) / 64), dim3(1024), 0, st, partial, nblk, H, out, accumulate);
    return check_launch("vk_colsum_bf16");
}
