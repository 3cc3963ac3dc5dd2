/* libvolta_hip.so -- C ABI of the MI355X (gfx950) kernels behind volta's BertForVLPreTraining.
 *
 * Conventions: raw device pointers + explicit sizes, hipStream_t last, int return
 * (0 = ok, <0 = error, text via vk_last_error()).  Nothing here allocates, frees or synchronises; every
 * buffer (including workspaces) belongs to the caller (PyTorch).  All activations are bf16 (raw
 * uint16 bits) row-major, statistics / losses / gradients of parameters are fp32.
 *
 * Each entry names the reference interface it replaces (paths relative to the volta repo).
 */
#ifndef VOLTA_HIP_H
#define VOLTA_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vk_stream_t; /* hipStream_t */

int vk_version(void);
const char* vk_device_arch(void);          /* "gfx950" (compile target) */
const char* vk_last_error(void);           /* thread-local message of the last failing call */

/* ------------------------------------------------------------------------------------------------
 * Dropout stream.  Replaces torch.nn.Dropout's global generator (volta/encoders.py:207,218,377,391,
 * 519,534; volta/embeddings.py:53,137,...) with a counter-based Philox-4x32-10 stream: element
 * (row, c) of dropout site `site` takes word c&3 of philox(counter=(c>>2,row,site,0), key=*seed).
 * `seed` lives in device memory so that a captured hipGraph replays with fresh randomness. */
typedef struct vk_dropout {
    const uint64_t* seed; /* device pointer */
    uint32_t site;
    uint32_t threshold;   /* floor(p * 2^32); 0 disables */
    float scale;          /* 1/(1-p) */
} vk_dropout;

/* Writes *seed_dev = seed (tiny kernel; once per training step). */
int vk_set_seed(uint64_t* seed_dev, uint64_t seed, vk_stream_t s);

/* fp32 -> bf16 (round to nearest even).  The reference feeds fp32 region features / weights to fp32
 * GEMMs; this engine's MFMA operands are bf16. */
int vk_cast_f32_bf16(const float* src, void* dst, int64_t n, vk_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * GEMM family (bf16 MFMA 16x16x32, fp32 accumulate).  Replaces torch.nn.functional.linear and its
 * autograd (every nn.Linear in volta/encoders.py:204-217,376,390,463,477,518,533,599,629,646,663,
 * 686,727,747 and volta/embeddings.py:134-135).
 *   layout NT: C[M,N] = A[M,K] * B[N,K]^T          (forward:  y = x W^T)
 *   layout NN: C[M,N] = A[M,K] * B[K,N]            (dgrad:    dx = dy W)
 *   layout TN: C[M,N] = A[K,M]^T * B[K,N]          (wgrad:    dW = dy^T x ; K = number of rows)
 * Requirements: lda/ldb multiples of 8 elements, 16-byte aligned bases, K % 64 == 0 for NT/NN
 * (or zero / finite padding up to the next multiple of 64); M, N arbitrary.
 * `dyn` (device int32, may be NULL) replaces M (NT/NN) or K (TN) at run time: used for the heads
 * that run on labelled rows only, whose count is known on the device only. */
enum { VK_NT = 0, VK_NN = 1, VK_TN = 2 };
enum {
    VK_EPI_BF16 = 0,  /* C(bf16) = acc + bias                                              */
    VK_EPI_GELU = 1,  /* u = acc + bias ; C = gelu(u) ; C2 = gelu'(u)      (encoders.py:130) */
    VK_EPI_MULR = 2,  /* C(bf16) = acc * R                                  (gelu backward)  */
    VK_EPI_ADDR = 3,  /* C(bf16) = acc + bias + R                           (grad accumulate)*/
    VK_EPI_F32 = 4,   /* C(fp32) = acc + bias, columns [N, n_store) written as 0           */
    VK_EPI_RELU = 5   /* C(bf16) = max(acc + bias, 0)                       (poolers)        */
};
typedef struct vk_gemm_problem {
    const void* A;
    const void* B;
    void* C;
    void* C2;              /* VK_EPI_GELU: derivative output, same ld as C */
    const float* bias;     /* [N] fp32 or NULL */
    const void* R;         /* bf16 [M, ldr] for MULR / ADDR */
    float* bias_grad;      /* TN only: if non-NULL receives sum over the K rows of A -> [M] fp32 */
    const int32_t* dyn;    /* see above */
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldr;
    int32_t n_store;       /* VK_EPI_F32: zero-fill columns up to here (>= N), else 0 */
} vk_gemm_problem;
#define VK_GEMM_MAX_GROUP 8
/* One launch for up to VK_GEMM_MAX_GROUP independent problems of the same layout / epilogue. */
int vk_gemm_grouped(int layout, int epilogue, const vk_gemm_problem* probs, int nprob, vk_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Fused (dropout +) residual + LayerNorm.  Replaces apex FusedLayerNormAffineFunction
 * (apex/csrc/layer_norm_cuda.cpp:121-240: forward_affine / backward_affine) and the python fallback
 * BertLayerNorm (volta/encoders.py:48-61) together with the dropout and residual add that precede
 * it (volta/encoders.py:410-423, 552-565).  eps = 1e-12 inside the sqrt, biased variance, fp32 stats.
 *   z = drop_pre(d) + x ;  y = drop_post(gamma * (z - mean) * rstd + beta)
 * Rows [0, split_row) use drop.site, rows >= split_row use drop.site + 1 with row index restarting at
 * 0 (the reference draws separate masks for the text and vision tensors of a shared sub-layer). */
typedef struct vk_ln_args {
    const void* d;         /* bf16 [M, H]  dense output (bias already added)                    */
    const void* x;         /* bf16 [M, H]  residual input or NULL                                */
    const float* gamma;    /* [H] */
    const float* beta;     /* [H] */
    void* y;               /* bf16 [M, H]                                                       */
    void* z;               /* bf16 [M, H]  pre-LN activations saved for backward (may alias d)  */
    float* mean;           /* [M] */
    float* rstd;           /* [M] */
    int32_t M, H;
    int32_t split_row;     /* >= M when unused */
    int32_t post;          /* 0: dropout on d before the add; 1: dropout on the LN output        */
    float out_scale;       /* y is multiplied by this after LN (LXMERT's (a+b)/2); normally 1    */
    vk_dropout drop;
} vk_ln_args;
int vk_ln_fwd(const vk_ln_args* a, vk_stream_t s);

typedef struct vk_ln_bwd_args {
    const void* dy;        /* bf16 [M, H] */
    const void* z;         /* bf16 [M, H] saved by forward */
    const float* mean;
    const float* rstd;
    const float* gamma;
    void* dz;              /* bf16 [M, H]: gradient w.r.t. z (= gradient of the residual branch)  */
    void* dd;              /* bf16 [M, H]: gradient w.r.t. d (dropout mask re-applied); may be NULL when no dropout */
    float* partial;        /* workspace fp32 [vk_ln_bwd_partial_rows(M), 2, H] */
    float* dgamma;         /* [H] */
    float* dbeta;          /* [H] */
    int32_t M, H;
    int32_t split_row;
    int32_t post;
    float out_scale;
    vk_dropout drop;
} vk_ln_bwd_args;
int vk_ln_bwd_partial_rows(int M);
int vk_ln_bwd(const vk_ln_bwd_args* a, vk_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Gated bimodal attention.  Replaces the body of BertGatedSelfAttention.forward after the Q/K/V
 * projections (volta/encoders.py:258-340: up to four score matmuls, scaling, additive masks, the
 * joint softmax over [text keys | vision keys], per-block dropout, up to four context matmuls, head
 * merge and the tt+tv / vv+vt sums) and its autograd.  Index 0 = text, 1 = vision.
 *   gate[mq][mk] != 0  <=>  queries of modality mq attend keys of modality mk
 *                           (gate[0][0]=has_tt, gate[0][1]=has_tv, gate[1][0]=has_vt, gate[1][1]=has_vv)
 * q/k/v[m] point at the Q / K / V column block of modality m's projection output (row stride ld[m],
 * head h at columns h*64..h*64+63); ctx[m] receives the merged heads [B*L[m], ldo[m]].  mask[m] is the
 * additive key mask [B, L[m]] (0 / -10000).  lse[m] ([B, nh, L[m]] fp32) is saved for the backward, which
 * recomputes the probabilities instead of storing them.  head size is 64; L[0] <= 64, L[1] <= 128. */
typedef struct vk_attn_args {
    const void* q[2];
    const void* k[2];
    const void* v[2];
    int32_t ld[2];
    int32_t L[2];
    const float* mask[2];
    void* ctx[2];
    int32_t ldo[2];
    float* lse[2];
    int32_t B, nh;
    int32_t gate[2][2];
    vk_dropout drop[2][2];   /* one dropout site per score block, as in the reference */
    float scale;             /* 1/sqrt(head size) */
} vk_attn_args;
typedef struct vk_attn_bwd_args {
    const void* dctx[2];     /* bf16 [B*L[m], ldo[m]] gradient of ctx */
    void* dq[2];             /* bf16, row stride ldg[m], same column convention as q/k/v */
    void* dk[2];
    void* dv[2];
    int32_t ldg[2];
} vk_attn_bwd_args;
int vk_gated_attn_fwd(const vk_attn_args* a, vk_stream_t s);
/* `a` must be the forward call's arguments (ctx and lse now inputs). */
int vk_gated_attn_bwd(const vk_attn_args* a, const vk_attn_bwd_args* b, vk_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
