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
#include <stddef.h>
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
 * 519,534; volta/embeddings.py:53,137,...) with a counter-based Philox-4x32-7 stream: element
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
    VK_EPI_RELU = 5,  /* C(bf16) = max(acc + bias, 0)                       (poolers)        */
    VK_EPI_F32_ACC = 6 /* C(fp32) += acc, bias_grad += ...  (wgrad of weights shared by two modalities) */
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
    /* Split accumulation (nparts >= 2; 0 / 1 = off): `nparts` problems of ONE launch with the same M, N, C, epilogue operands, `ws` and
       `cnt` and part = 0 .. nparts-1 are K-slices (or row chunks, TN) of one product.  Every workgroup leaves its fp32 partial tile in
       `ws`; the workgroup that arrives LAST at a tile (ticket on cnt[tile]) sums the partials in part order -- the result does not depend
       on the arrival order -- and runs the epilogue.  No spinning: a launch completes under any residency.  Needs the 256-row tile
       geometries (258 / 259 of vk_gemm_grouped_ex, one tile per workgroup) and no `dyn`.
       ws:  vk_gemm_split_workspace_bytes(layout, M, N, nparts, geometry) bytes, 16-byte aligned, contents irrelevant;
       cnt: one int32 per output tile (same query, second result), ZERO before the first launch; every launch leaves it zero. */
    void* ws;
    int32_t* cnt;
    int32_t part, nparts;
    /* Row-block hand-off between two launches of ONE stream (NULL / 0 = off; see VK_GEMM_SOFT_START below).  A row block is 256 rows.
       sig: int32 per row block of C.  Every tile adds 1 to sig[its row block] once its part of C is in memory: C leaves the epilogue as
            write-through stores, every wave drains them, one lane adds (agent scope).  After the launch sig[rb] = column tiles per row block.
       dep / dep_need: int32 per row block of A.  A tile asks for its A rows only after dep[its row block] >= dep_need (one lane polls,
            the workgroup waits at a barrier; the A rows are then read with cache-bypassing loads).  Everything ELSE the problem reads
            (B, bias, R) must be complete when the launch is enqueued.
       err: raised to 1 when a poll gives up (> 2^20 rounds: a mis-planned dependency; the tile then runs on whatever is there), or NULL.
       The counters are the caller's: zero before the PRODUCER is enqueued, read-only for the consumer.  NT / NN, geometries 258 / 259,
       whole tiles (M % 256 == 0; a signalling problem: geometry 258, N % 256 == 0, C rows 128-byte aligned), no dyn, no split. */
    int32_t* sig;
    const int32_t* dep;
    int32_t* err;
    int32_t dep_need;
    int32_t reserved_;
    /* Problem 0 of a launch only (NULL = off): every workgroup of the launch, as it retires, stores *retire_stamp (device uint64) to
       *retire_flag (device uint64, agent scope).  A vk_gate_wait on ANOTHER stream that polls the flag for that stamp opens when the launch
       begins to hand CUs back -- how the engine starts a sub-layer's weight-gradient block behind its last dgrad without a stream event,
       whose cross-queue wake-up latency is not under the program's control (volta_amd/engine.py, profiles/r04_experiments.md). */
    uint64_t* retire_flag;
    const uint64_t* retire_stamp;
} vk_gemm_problem;
#define VK_GEMM_MAX_GROUP 32
/* One launch for up to VK_GEMM_MAX_GROUP independent problems of the same layout / epilogue. */
int vk_gemm_grouped(int layout, int epilogue, const vk_gemm_problem* probs, int nprob, vk_stream_t s);
/* Same with the tile geometry chosen by the caller instead of the shape heuristic (0 = heuristic): 128 (128 x 128 tiles, 4 waves),
 * 258 / 259 / 260 (256 x 256 / 192 / 128 tiles, 8 waves, LDS-DMA ring), optionally OR-ed with VK_GEMM_PERSISTENT (one workgroup per CU
 * walks the tile list; NT / NN without `dyn`) or VK_GEMM_ONE_TILE_PER_WG; 261 (256 x 128 tiles, 4 waves, 72 KiB of LDS: two
 * workgroups per CU, csrc/gemm4w.hip); 262 (128 x 128 tiles, 4 waves, ring of 6 K-steps: launches too small for 256-row tiles).  A per-call argument, no library state: re-entrant. */
#define VK_GEMM_PERSISTENT 0x1000
#define VK_GEMM_ONE_TILE_PER_WG 0x2000
/* Soft boundary: enqueue the launch WITHOUT the stream-order barrier (the AQL packet's barrier bit is cleared, hipExtAnyOrderLaunch -- honoured
 * on gfx950, tools/micro/anyorder.hip): its workgroups take CUs as the launch in front of it frees them, and its tiles start as the row
 * blocks they read are signalled (`dep`), instead of after that launch's last tile, its end-of-kernel write-back and the dispatch ramp.
 * The caller guarantees: every input is complete at enqueue time or guarded by `dep`; nothing the launch writes is read or written by a
 * launch that may still run.  The next launch WITHOUT this flag waits for all earlier ones, as always.  (The workgroups of one queue are
 * dispatched in enqueue order -- tools/micro/dispatch_order.hip -- so a polling workgroup never holds a CU that a producer still needs.) */
#define VK_GEMM_SOFT_START 0x4000
int vk_gemm_grouped_ex(int layout, int epilogue, const vk_gemm_problem* probs, int nprob, int geometry, vk_stream_t s);
/* Chain: a producer group and the consumer group that reads its outputs in ONE persistent launch (the reference's BertGatedIntermediate ->
 * BertGatedOutput pair, volta/encoders.py:486-501 -> 541-566, and its backward: FFN-down dgrad x gelu' -> FFN-up dgrad).  Producers run on
 * 256 x 256 tiles with epilogue `epi_p` and signal row blocks (`sig`); consumers run on 256 x 192 tiles with epilogue `epi_c`, each reads
 * ONE producer's C as its A operand (A = that C, M, K = its N, lda = its ldc) and names that producer's counters in `dep`, dep_need = N / 256.
 * One workgroup per CU walks its producer tiles, then its consumer tiles behind the row-block polls: no launch boundary between the two
 * products, early finishers start on consumer tiles while the others finish.  Whole tiles only; counters zero before the launch; err as
 * for the hand-off.  Built pairs: NT with GELU -> BF16, NN with MULR -> ADDR. */
int vk_gemm_chain(int layout, int epi_p, const vk_gemm_problem* producers, int np, int epi_c, const vk_gemm_problem* consumers, int nc, vk_stream_t s);
/* Workspace of a split accumulation (see vk_gemm_problem::ws): bytes of `ws` for one product [M, N] cut into `nparts` parts under tile
 * geometry 258 / 259; *tiles receives the number of int32 counters `cnt` needs.  Host-side arithmetic, no device work. */
size_t vk_gemm_split_workspace_bytes(int layout, int M, int N, int nparts, int geometry, int* tiles);

/* fp8 (OCP e4m3) forward projections on v_mfma_scale_f32_16x16x128_f8f6f4 (BASELINE.json configs[4]; the reference is fp32, the
 * sites are the nn.Linear forwards of volta/encoders.py:242-255, 495-499, 552-565).  Layout NT only: C[M,N] = (A8[M,K] . B8[N,K]^T)
 * * scale_a[m] * scale_b[n] (+ bias, epilogue): A8 / B8 are e4m3 bytes, lda / ldb in BYTES (multiples of 16), K padded to a multiple
 * of 128 by the leading dimensions; scale vectors are the per-row de-quantisation factors written by vk_quant_rows_fp8 (NULL = 1).
 * Epilogues BF16, GELU, F32, RELU; geometry 0 (heuristic) / 128 / 256.  Gradients stay on the bf16 kernels. */
#define VK_GEMM_FP8_MAX_GROUP 4
typedef struct vk_gemm_fp8_problem {
    vk_gemm_problem p;        /* R, bias_grad unused */
    const float* scale_a;     /* [M] or NULL */
    const float* scale_b;     /* [N] or NULL */
    void* c8;                 /* VK_EPI_GELU only: e4m3 copy of C, c8[m, n] = saturate(C[m, n] * c8_mul), row stride ldc8 bytes; or NULL */
    float c8_mul;
    int32_t ldc8;
} vk_gemm_fp8_problem;
int vk_gemm_fp8_grouped(int epilogue, const vk_gemm_fp8_problem* probs, int nprob, int geometry, vk_stream_t s);
/* Row-wise e4m3 quantisation: scale[m] = max|x[m,:]| / 448 (1 for a zero row), dst[m,k] = rne(x[m,k] / scale[m]); src bf16 (or fp32),
 * ld / ldq in elements / bytes, K a multiple of 8 and <= 4096; dyn (device int32 or NULL) limits the rows. */
int vk_quant_rows_fp8(const void* src, int src_is_f32, int64_t ld, void* dst, int64_t ldq, float* scale, int M, int K, const int32_t* dyn, vk_stream_t s);
/* dst[i] = e4m3(saturate(src[i] * mul)), src bf16: static-scale quantisation of a whole tensor (the GELU output). */
int vk_cast_bf16_fp8(const void* src, void* dst, int64_t n, float mul, vk_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Fused (dropout +) residual + LayerNorm.  Replaces apex FusedLayerNormAffineFunction
 * (apex/csrc/layer_norm_cuda.cpp:121-240: forward_affine / backward_affine) and the python fallback
 * BertLayerNorm (volta/encoders.py:48-61) together with the dropout and residual add that precede
 * it (volta/encoders.py:410-423, 552-565).  eps = 1e-12 inside the sqrt, biased variance, fp32 stats.
 *   z = drop_pre(d) + x + addvec ;  y = drop_post(gamma * (z - mean) * rstd + beta) * out_scale
 * Dropout row index: rows [0, split_row) belong to segment 0, the rest to segment 1 (the reference
 * draws separate masks for the text and vision tensors of a shared sub-layer, or one mask over their
 * per-sample concatenation in the single-stream embeddings).  With r' = row - segment start the Philox
 * row is (r' / div) * mul + (r' % div) + off  (div == 0: r' itself). */
typedef struct vk_drop_rows {
    uint32_t site;
    int32_t div, mul, off;
} vk_drop_rows;
typedef struct vk_ln_args {
    const void* d;         /* bf16 [M, H]  dense output (bias already added)                    */
    const void* x;         /* bf16 [M, H]  residual input or NULL                                */
    const float* addvec;   /* fp32 [H] added to every row before the statistics, or NULL         */
    const float* gamma;    /* [H] */
    const float* beta;     /* [H] */
    void* y;               /* bf16 [M, H]                                                       */
    void* z;               /* bf16 [M, H]  pre-LN activations saved for backward (may alias d)  */
    float* mean;           /* [M] */
    float* rstd;           /* [M] */
    const int32_t* dyn;    /* device row count overriding M when non-NULL                        */
    int32_t M, H;
    int32_t split_row;     /* >= M when unused */
    int32_t post;          /* 0: dropout on d before the add; 1: dropout on the LN output        */
    float out_scale;       /* normally 1 (LXMERT image embedding: 0.5)                            */
    vk_dropout drop;       /* seed / threshold / scale; drop.site is ignored, see seg[]           */
    vk_drop_rows seg[2];
    void* y8;              /* optional e4m3 copy of y, quantised per row (fp8 projection path): [M, ld8] bytes, or NULL */
    float* y8_scale;       /* [M] de-quantisation factors: y[m, :] ~ y8[m, :] * y8_scale[m]                              */
    int64_t ld8;           /* row stride of y8 in bytes (multiple of 8)                                                   */
} vk_ln_args;
int vk_ln_fwd(const vk_ln_args* a, vk_stream_t s);
/* Two independent jobs of equal H in ONE launch (text and vision stream of a sub-layer); b may be NULL. */
int vk_ln_fwd_pair(const vk_ln_args* a, const vk_ln_args* b, vk_stream_t s);

typedef struct vk_ln_bwd_args {
    const void* dy;        /* bf16 [M, H] */
    const void* z;         /* bf16 [M, H] saved by forward */
    const float* mean;
    const float* rstd;
    const float* gamma;
    void* dz;              /* bf16 [M, H]: gradient w.r.t. z (= gradient of the residual branch and of addvec rows) */
    void* dd;              /* bf16 [M, H]: gradient w.r.t. d (pre-dropout mask re-applied); NULL: not needed */
    float* partial;        /* workspace fp32 [vk_ln_bwd_partial_rows(M), 2, H] */
    float* dgamma;         /* [H] */
    float* dbeta;          /* [H] */
    const int32_t* dyn;
    int32_t M, H;
    int32_t split_row;
    int32_t post;
    float out_scale;
    int32_t accumulate;    /* bit 0: dgamma / dbeta += (shared sub-layers: one LayerNorm, two modalities);
                              bit 1: leave the column reduction of `partial` to a later vk_ln_bwd_finalize (same args) */
    vk_dropout drop;
    vk_drop_rows seg[2];
} vk_ln_bwd_args;
int vk_ln_bwd_partial_rows(int M);
int vk_ln_bwd(const vk_ln_bwd_args* a, vk_stream_t s);
int vk_ln_bwd_pair(const vk_ln_bwd_args* a, const vk_ln_bwd_args* b, vk_stream_t s);
/* dgamma / dbeta from the per-workgroup partial records of a vk_ln_bwd issued with accumulate bit 1 set: off the
   critical path of the backward pass (the parameters' gradients are only needed by the optimizer / all-reduce). */
int vk_ln_bwd_finalize(const vk_ln_bwd_args* a, vk_stream_t s);

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
    int32_t dh;              /* head size: 0 or 64 -> the MFMA kernels (every ctrl_* config); 32, 96, 128 -> the generic kernels
                                (config/vilbert_base.json: 8 heads of 128), column h * dh of q / k / v / ctx is head h */
    float* probs[2][2];      /* NULL, or for block (query modality i, key modality j) fp32 [B, nh, L[i], L[j]]: the attention
                                probabilities after dropout, as BertGatedSelfAttention returns them under config.visualization
                                (volta/encoders.py:342-358).  Any non-NULL entry routes the forward to the generic kernels. */
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
/* Host arithmetic, no device work: bytes of LDS the launch would ask for per workgroup when it runs on the generic kernels (rows beyond
 * the MFMA tiles, head sizes 32 / 96, attention maps), 0 when the MFMA kernels serve it.  More than 160 KiB cannot run: a planner checks
 * BEFORE the first step (the backward needs about twice the forward's: ~491 keys at head size 64, ~258 at 128). */
size_t vk_gated_attn_lds_bytes(const vk_attn_args* a, int backward);

/* ------------------------------------------------------------------------------------------------
 * Embeddings.  vk_embed_sum_fwd replaces the three nn.Embedding lookups and their sum in
 * BertEmbeddings.forward (volta/embeddings.py:55-66; also :345-348, :442-445); the LayerNorm +
 * dropout that follow are vk_ln_fwd (post = 1).  pos_ids == NULL means position = row % T. */
typedef struct vk_embed_args {
    const int64_t* ids;       /* [M] */
    const int64_t* type_ids;  /* [M] or NULL (all 0) */
    const int64_t* pos_ids;   /* [M] or NULL */
    const float* word;        /* [V, H] fp32 master weights */
    const float* pos;         /* [P, H] */
    const float* type;        /* [n_types, H] */
    const void* extra;        /* bf16 [M, H] added per row, or NULL (VL-BERT visual part) */
    void* z;                  /* bf16 [M, H] out */
    int32_t M, T, H;
    int32_t V, P, n_types;    /* table heights: indices are clamped into range (no device fault on bad ids) */
} vk_embed_args;
int vk_embed_sum_fwd(const vk_embed_args* a, vk_stream_t s);
/* Backward of the lookups (autograd of nn.Embedding): dword / dtype (and dpos when pos_ids != NULL) are
 * ACCUMULATED with atomics (caller zeroes them, or lets the tied LM-decoder wgrad write dword first);
 * with implicit positions dpos rows [0, T) are overwritten. */
typedef struct vk_embed_bwd_args {
    const void* dz;           /* bf16 [M, H] */
    const int64_t* ids;
    const int64_t* type_ids;
    const int64_t* pos_ids;
    float* dword;
    float* dpos;
    float* dtype;             /* may be NULL */
    int32_t M, T, H, n_types;
    int32_t V, P;
} vk_embed_bwd_args;
int vk_embed_sum_bwd(const vk_embed_bwd_args* a, vk_stream_t s);

/* Box-location linear (nn.Linear(num_locs=5, H), volta/embeddings.py:135,141,157,164,420,450): fp32 in,
 * bf16 out; backward gives dW [H, nloc] and db [H] (partial: fp32 [vk_rows32(M), 9, H] workspace). */
int vk_rows32(int M);
int vk_loc_linear_fwd(const float* loc, const float* W, const float* bias, void* out, int M, int H, int nloc, vk_stream_t s);
int vk_loc_linear_bwd(const void* dz, const float* loc, float* partial, float* dW, float* db, int M, int H, int nloc, vk_stream_t s);
/* y = dropout((a + b) * scale)  (LXMERT image embedding, embeddings.py:169-170); backward != 0:
 * y = a * keep * scale (b ignored). */
int vk_add_dropout(const void* a, const void* b, void* y, int M, int H, float scale, vk_dropout drop, int backward, vk_stream_t s);
/* out[c] (+)= sum_m src[m][c]; partial: fp32 [vk_rows32(M), H]. */
int vk_colsum_bf16(const void* src, float* partial, float* out, int M, int H, int accumulate, vk_stream_t s);
/* VL-BERT region input (volta/embeddings.py:102-124 coordinate_embeddings, :243-251): builds the bf16 row
 * [sin|cos box embedding (4 x 2*dim) | appearance feature (F)] of the obj_downsample input with dropout;
 * all-zero feature rows take mask_emb (object_mask_visual_embedding) and are flagged in zero_flag[M]. */
int vk_vlbert_prep_fwd(const float* loc, int nloc, const float* feat, const float* mask_emb, void* out, int32_t* zero_flag,
                       int M, int F, int dim, vk_dropout drop, vk_stream_t s);
/* d(mask_emb)[f] = sum over flagged rows of dropout-masked dx[row][col0 + f]; partial: fp32 [vk_rows32(M), F]. */
int vk_vlbert_maskgrad(const void* dx, int ldx, int col0, const int32_t* zero_flag, float* partial, float* out, int M, int F,
                       vk_dropout drop, vk_stream_t s);
/* out[b] = sum of the T consecutive rows of sample b (gradient of a per-sample broadcast), bf16 [B, H]. */
int vk_rowgroup_sum_bf16(const void* in, void* out, int B, int T, int H, vk_stream_t s);
/* out = dy where y > 0 else 0 (ReLU backward from the kept output), n bf16 elements. */
int vk_relu_bwd_bf16(const void* dy, const void* y, void* out, int64_t n, vk_stream_t s);
int vk_copy_async(void* dst, const void* src, int64_t bytes, vk_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Heads and losses, evaluated on labelled rows only.  Replaces BertPreTrainingHeads + the loss code of
 * BertForVLPreTraining.forward (volta/encoders.py:766-784, 1079-1109) and kl_1601 (volta/losses.py:16-22),
 * including the two host synchronisations at encoders.py:1089 and :1111 (counts stay on the device). */
/* flag(i) = labels[i] != -1 (mode 0) | labels[i] == 1 (mode 1).  For the j-th flagged i:
 * pos[j] = i, rows[j] = (i / inner) * outer + (i % inner) + off; *count = number flagged. */
int vk_select_rows(const int64_t* labels, int N, int mode, int inner, int outer, int off, int32_t* rows, int32_t* pos,
                   int32_t* count, vk_stream_t s);
int vk_gather_rows(const void* src, const int32_t* rows, const int32_t* count, void* dst, int H, int max_rows, vk_stream_t s);
int vk_scatter_rows_add(const void* src, const int32_t* rows, const int32_t* count, void* dst, int H, int max_rows, vk_stream_t s);
typedef struct vk_xent_args {
    const float* logits;      /* fp32 [rows, ld] */
    const int64_t* labels;    /* label of row i = labels[pos ? pos[i] : i] */
    const int32_t* pos;       /* or NULL */
    const int32_t* count;     /* device row count, or NULL: max_rows rows */
    float* lse;               /* [rows] saved */
    float* loss_sum;          /* accumulated (atomicAdd): zero it first */
    int32_t V, ld, max_rows;
} vk_xent_args;
int vk_xent_fwd(const vk_xent_args* a, vk_stream_t s);
/* dlogits(bf16)[i][c] = (softmax - onehot) * (*gscale) / count; columns [V, ldd) are written as 0 */
int vk_xent_bwd(const vk_xent_args* a, void* dlogits, int ldd, const float* gscale, vk_stream_t s);
typedef struct vk_kl_args {
    const float* logits;      /* fp32 [rows, ld] */
    const float* target;      /* fp32 [*, V]: row pos[i] is the target distribution of row i */
    const int32_t* pos;
    const int32_t* count;
    float* lse;
    float* tsum;
    float* loss_sum;
    float weight;             /* visual_target_weights["0"] */
    int32_t V, ld, max_rows;
} vk_kl_args;
int vk_kl_fwd(const vk_kl_args* a, vk_stream_t s);
int vk_kl_bwd(const vk_kl_args* a, void* dlogits, int ldd, const float* gscale, vk_stream_t s);
/* losses[0] = sums[0] / *n_t ; losses[1] = w * sums[1] / max(*n_v, 1) ; losses[2] = sums[2] / B.  vk_kl_fwd and vk_vis_loss_fwd
   accumulate WEIGHTED row losses (several targets share sums[1]), so the engine passes w = 1. */
int vk_loss_finalize(const float* sums, const int32_t* n_t, const int32_t* n_v, int B, float kl_weight, float* losses, vk_stream_t s);
/* pooled = dropout(pooled_t * pooled_v) (encoders.py:769-770) and its backward through the two ReLUs */
int vk_pool_mul_fwd(const void* pt, const void* pv, void* out, int B, int P, vk_dropout drop, vk_stream_t s);
int vk_pool_mul_bwd(const void* dp, int ldp, const void* pt, const void* pv, void* dyt, void* dyv, int B, int P, vk_dropout drop, vk_stream_t s);
/* The other fusions of the two pooled vectors (BertPreTrainingHeads.forward / BertForVLTasks.forward, encoders.py:766-778,1184-1195):
   out = dropout(pt * pv | pt + pv | pt); "text" and "vl-bert_vqa" have no vision pooler (pv, dyv NULL).  vk_pool_mul_* = mode VK_FUSE_MUL. */
enum { VK_FUSE_MUL = 0, VK_FUSE_SUM = 1, VK_FUSE_TEXT = 2 };
int vk_pool_fuse_fwd(const void* pt, const void* pv, void* out, int B, int P, int mode, vk_dropout drop, vk_stream_t s);
int vk_pool_fuse_bwd(const void* dp, int ldp, const void* pt, const void* pv, void* dyt, void* dyv, int B, int P, int mode, vk_dropout drop, vk_stream_t s);
/* VLBertTextPooler (encoders.py:610-623) pools the token two places before the end of the caption:
   rows[b] = b * T + max(#non-zero ids of caption b - 2, 0), *count = B (feeds vk_gather_rows / vk_scatter_rows_add). */
int vk_text_end_rows(const int64_t* ids, int B, int T, int32_t* rows, int32_t* count, vk_stream_t s);
/* VL-BERT with the xent_1601 target gives masked regions (all-zero feature rows) a word of their own (embeddings.py:191,262-264):
   ids[m] = 1 for the last region of a sample (END), 2 for a masked region, 0 otherwise -- the row of the 3-row word table. */
int vk_vlbert_obj_ids(const int32_t* zero_flag, int64_t* ids, int M, int K, vk_stream_t s);
/* VL-BERT's position ids (embeddings.py:278-292) from input_ids [B, T] (0 = pad), K boxes per sample:
   text_end[b] = #non-zero ids; tpos[b, t] = t + (t >= min_b text_end[b] ? K : 0) (the reference's shift goes through a stride-0
   expanded view, so it lands in the row all samples share); opos[b, k] = text_end[b] + (k == K - 1). */
int vk_vlbert_positions(const int64_t* ids, int B, int T, int K, int64_t* tpos, int64_t* opos, vk_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * The visual targets other than kl_1601 (volta/losses.py:25-126), on the labelled regions only.  Row i of `logits` is the
 * decoder's prediction for the i-th masked region; pos[i] = its index in the [B, R] label grid (= row of target / labels / conf).
 * Every kernel adds weight x row loss to *loss_sum; the image loss is loss_sum / max(*count, 1):
 *   VK_VIS_MSE    mse_2048   (:25-33)     mean_c (x - feat)^2
 *   VK_VIS_HUBER  huber_2048 (:105-113)   mean_c smooth_l1(x - feat)
 *   VK_VIS_XENT   xent_1600 / xent_400 (:83-102, conf = detector confidence) and xent_1601 (:116-124, conf NULL)
 *   VK_VIS_NCE    nce_2048   (:36-80)     lse_j <sample_j, x> - <sample_0, x>, sample_0 = the region's own feature,
 *                                         samples 1..n_neg = target rows neg_index[pos * n_neg + j - 1] (vk_nce_negatives)
 * The backward writes dlogits (bf16, columns [V, ldd) = 0) = d(loss_sum / max(count,1)) / d logits x *gscale.
 * ---------------------------------------------------------------------------------------------- */
enum { VK_VIS_MSE = 1, VK_VIS_NCE = 2, VK_VIS_XENT = 3, VK_VIS_HUBER = 5 };
#define VK_NCE_ACROSS 89          /* int(128 * 0.7) negatives from other images of the batch (losses.py:45) */
#define VK_NCE_INSIDE 38          /* int(128 * 0.3) negatives from the same image (losses.py:46) */
#define VK_NCE_MAX_SAMPLES 128
typedef struct vk_vis_loss_args {
    const float* logits;       /* fp32 [rows, ld] */
    const float* target;       /* regression / nce: the region features the model was given, fp32 [B*R, V] */
    const int64_t* labels;     /* xent: [B*R] */
    const float* conf;         /* xent: [B*R] or NULL */
    const int32_t* pos;        /* [rows] */
    const int32_t* count;      /* device row count */
    const int32_t* neg_index;  /* nce: [B*R, n_neg] */
    float* lse;                /* [rows] saved (xent, nce) */
    float* aux;                /* nce: [rows, VK_NCE_MAX_SAMPLES] saved scores */
    float* loss_sum;
    float weight;              /* visual_target_weights[ix] */
    int32_t V, ld, max_rows, kind, n_neg;
} vk_vis_loss_args;
int vk_vis_loss_fwd(const vk_vis_loss_args* a, vk_stream_t s);
int vk_vis_loss_bwd(const vk_vis_loss_args* a, void* dlogits, int ldd, const float* gscale, vk_stream_t s);
/* The 127 negatives of every region as losses.py:47-69 draws them, from the counter-based stream (rng.seed, rng.site):
   out[(b*R + r) * 127 + j]; j < 89: a region of another image, j >= 89: another region of the same image.  B, R >= 2. */
int vk_nce_negatives(vk_dropout rng, int B, int R, int32_t* out, vk_stream_t s);

/* additive attention mask (1 - m) * -10000 (encoders.py:983-991) */
int vk_mask_prep(const int64_t* mask, float* out, int n, vk_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Optimizer over flat fp32 arenas.  vk_grad_norm_clip replaces torch.nn.utils.clip_grad_norm_
 * (train_concap.py:307-308): out[0] = ||g|| * pre_scale, out[1] = min(1, max_norm / (norm + 1e-6)) stay on
 * the device.  vk_adamw_step replaces pytorch_transformers.optimization.AdamW.step (train_concap.py:227,310;
 * package pinned at 1.1.0 in requirements.txt:39, not vendored):
 *   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_c * step_mult * m / (sqrt(v) + eps) ; p -= lr_c * wd_c * p
 * with g pre-multiplied by grad_scale * clip[1]; step_mult = sqrt(1-b2^t)/(1-b1^t) (correct_bias) or 1.
 * Arena length is a multiple of 1024; chunk_class[i] selects (lr multiplier, weight decay) of chunk i.  Class
 * VK_CHUNK_SKIP marks chunks that take no part: frozen parameters (config.fixed_layers, volta/train_utils.py:250-255;
 * the driver builds its groups from requires_grad parameters only, train_concap.py:200,213) and parameters without a
 * gradient this step -- pytorch_transformers' AdamW skips `p.grad is None`, torch's clip_grad_norm_ likewise: no moment
 * update, no decay, left out of the norm. */
#define VK_CHUNK_SKIP 255
int vk_grad_norm_workspace_floats(void);
int vk_grad_norm_clip(const float* g, int64_t n, float pre_scale, float max_norm, float* partial, float* out, vk_stream_t s);
/* same, leaving out the 1024-element chunks whose chunk_class is VK_CHUNK_SKIP (chunk_class NULL: all chunks) */
int vk_grad_norm_clip_masked(const float* g, int64_t n, const uint8_t* chunk_class, float pre_scale, float max_norm, float* partial, float* out, vk_stream_t s);
/* Shard-decomposable form of the same norm, for data-parallel ranks that each own a part of the gradient arena (volta_amd/parallel.py,
 * mode "zero1"): vk_grad_sqnorm_chunks writes sums[c] = sum of squares of the 1024-element chunk c of g for c in [chunk0, chunk0 + nchunks)
 * (0 for VK_CHUNK_SKIP chunks) -- a function of that chunk's data alone -- and vk_grad_norm_from_chunks adds all `total_chunks` sums in a
 * fixed order (double) and leaves out[0..1] as vk_grad_norm_clip does.  A rank that computed only its own chunks and received the other
 * ranks' sums obtains exactly the bits of a rank that computed them all. */
int vk_grad_sqnorm_chunks(const float* g, int64_t chunk0, int64_t nchunks, const uint8_t* chunk_class, float* sums, vk_stream_t s);
/* `sums`: total_chunks floats (total_chunks even, buffer 8-byte aligned) followed by 256 floats of scratch for the first level of the sum. */
int vk_grad_norm_from_chunks(float* sums, int64_t total_chunks, float pre_scale, float max_norm, float* out, vk_stream_t s);
typedef struct vk_adamw_args {
    float* p;
    const float* g;
    float* m;
    float* v;
    void* shadow;               /* bf16 copy of p, refreshed in the same pass, or NULL */
    const uint8_t* chunk_class; /* [n / 1024] or NULL (class 0) */
    const float* clip;          /* device float[2] from vk_grad_norm_clip or NULL */
    int64_t n;
    float cls_lr_mult[8];       /* classes 0..7; chunks of class VK_CHUNK_SKIP are not touched */
    float cls_wd[8];
    float lr, beta1, beta2, eps, step_mult, grad_scale;
} vk_adamw_args;
int vk_adamw_step(const vk_adamw_args* a, vk_stream_t s);
/* The same update (same bits) by `ncus` resident workgroups, each keeping one compute unit to itself and striding over the arena: the form
 * for a step that runs on a stream of its own under the next forward pass, beside GEMM launches that claim the rest of the chip. */
int vk_adamw_step_on(const vk_adamw_args* a, int ncus, vk_stream_t s);
int vk_axpy_f32(float* y, const float* x, float alpha, int64_t n, vk_stream_t s);
/* dst[i] = sum_{s < nslabs} src[s * slab_stride + i], i < n (fp32).  Combines the partial weight gradients of a
 * split-K wgrad: each K-chunk is an ordinary problem of the grouped TN launch writing its own slab. */
int vk_sum_slabs_f32(float* dst, const float* src, int64_t slab_stride, int nslabs, int64_t n, vk_stream_t s);
/* Same with a bf16 destination and a device-side row count: rows = min(*dyn_rows, n / row_len) rows of row_len. */
int vk_sum_slabs_bf16(void* dst, const float* src, int64_t slab_stride, int nslabs, int64_t n, const int32_t* dyn_rows, int row_len, vk_stream_t s);
int vk_memset_async(void* p, int value, int64_t bytes, vk_stream_t s);
/* No counterpart in the reference: `nwg` (<= 256) one-wave workgroups that stay resident for `usec` (<= 100000) microseconds.
 * whole_cu != 0: each declares all 160 KiB of a CU's LDS, i.e. owns a CU -- the uneven load under which the row-block hand-offs are
 * tested and the stream probe of volta_amd/streams.py; whole_cu == 0: a wave slot and nothing else -- a pure DELAY on its stream
 * (VK_FN_HOLD in a command list: the head of a weight-gradient side block, see volta_amd/engine.py). */
int vk_hold_cus(int nwg, int usec, int whole_cu, vk_stream_t s);
/* One wave that stays on stream `s` until *flag == *stamp (device uint64 words, polled with cache-bypassing loads; gives up after
 * timeout_us, <= 1000000, and raises *err (may be NULL) to 2): everything enqueued on `s` behind it starts when another stream's
 * launch has stored the stamp (vk_gemm_problem::retire_flag).  The stamp must already hold its value when the gate is ENQUEUED-and-reached:
 * the engine orders its side stream behind the launch that bumps the epoch word with one event per backward pass (vk_bump_u64). */
int vk_gate_wait(const uint64_t* flag, const uint64_t* stamp, int timeout_us, int32_t* err, vk_stream_t s);
/* *word += 1 (device uint64; one lane): the epoch of the gates of one backward pass. */
int vk_bump_u64(uint64_t* word, vk_stream_t s);
/* Stream-ordered flag and the gate that waits for it, with the value known to the HOST (a step counter): `vk_store_u64` stores `value` to
 * *word when stream `s` gets there; `vk_gate_value` holds its stream (one spinning wave) until *flag == want.  Together they order one
 * stream behind a point of another WITHOUT hipStreamWaitEvent: on gfx950 an AQL barrier packet that sits unsatisfied at the head of a
 * queue stalls the other queues of its command-processor pipe until it is satisfied (profiles/r04_experiments.md: a reducer stream's
 * early-enqueued waits cost 2-7 ms on a 16.9 ms step), a running one-wave kernel does not.  volta_amd/parallel.py, volta_amd/optimization.py. */
int vk_store_u64(uint64_t* word, uint64_t value, vk_stream_t s);
int vk_gate_value(const uint64_t* flag, uint64_t want, int timeout_us, int32_t* err, vk_stream_t s);
/* Measurement aid: the CU footprint of a collective library's channel kernels without the library -- `nwg` workgroups of 256 threads copy
 * `bytes` (multiple of 16) from src to dst and stay resident until `min_usec` have passed since the first of them started (a channel
 * kernel lives as long as its transfer, at link rate).  stamps: NULL or 2 uint64 {~0, 0} that receive first start / last end in 100 MHz ticks.
 * Stands where apex's bucket all-reduce would be launched (apex/apex/parallel/distributed.py:425-475) in tools/comm_footprint.py. */
int vk_comm_standin(const void* src, void* dst, int64_t bytes, int nwg, int min_usec, uint64_t* stamps, vk_stream_t s);
/* The persistent GEMM launches (one workgroup per CU, geometries 258 / 259 | VK_GEMM_PERSISTENT) leave `n` CUs (0 .. 128) unclaimed from now
 * on: room for the channel kernels of a collective that runs beside the backward pass.  The ONE piece of process-wide state in the
 * library -- a launch geometry shared by every stream, set by volta_amd.parallel.DistributedDataParallel (VK_COMM_CUS); returns the
 * previous value.  n < 0 only queries. */
int vk_gemm_reserve_cus(int n);
/* The tail of a sub-layer's weight-gradient block in ONE launch: every split-K slab sum (kind 0, as vk_sum_slabs_f32) and every
 * deferred LayerNorm dgamma / dbeta column reduction (kind 1, as vk_ln_bwd_finalize) of the sub-layer.  The reference has no
 * counterpart (autograd accumulates each of these tensors with its own kernels); njobs <= VK_TAIL_MAX_JOBS. */
#define VK_TAIL_MAX_JOBS 16
typedef struct vk_tail_job {
    float* dst;             /* kind 0: destination [n]; kind 1: dgamma [H] */
    float* dst2;            /* kind 1: dbeta [H] */
    const float* src;       /* kind 0: slabs, slab s at src + s * stride; kind 1: partial records [count][2][H] */
    const float* src2;      /* kind 1: a second set of partial records [count2][2][H] summed into the same dgamma / dbeta (one LayerNorm
                               shared by the text and the vision stream of a sub-layer), or NULL */
    int64_t stride;         /* kind 0: elements between slabs (multiple of 4); kind 1: count2 */
    int64_t n;              /* kind 0: elements; kind 1: H */
    int32_t kind, count;    /* count: slabs (kind 0) / partial records (kind 1) */
    int32_t accumulate;     /* kind 1: dgamma / dbeta += */
    int32_t block_start;    /* filled by vk_side_tail */
} vk_tail_job;
int vk_side_tail(const vk_tail_job* jobs, int njobs, vk_stream_t s);

/* out[i] = a[i] * b[i] (bf16); processes rows * row_len elements where rows = min(*dyn_rows, n / row_len)
 * when dyn_rows != NULL.  (d gelu(u) = dz * gelu'(u) in the prediction-head transforms.) */
int vk_mul_bf16(const void* a, const void* b, void* out, int64_t n, const int32_t* dyn_rows, int row_len, vk_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Command lists.  A plan is an array of vk_op whose argument structs live in caller memory; vk_run_ops
 * issues them in order on one stream (stops at the first error).  There is no reference counterpart:
 * the reference walks a Python module tree and autograd graph every step (volta/encoders.py:868-881). */
enum {
    VK_OP_GEMM = 1,      /* a = vk_gemm_problem[i2], i0 = layout, i1 = epilogue */
    VK_OP_LN_FWD, VK_OP_LN_BWD,   /* a = args, b = args of a second job sharing the launch or NULL */
    VK_OP_ATTN_FWD,
    VK_OP_ATTN_BWD,      /* a = vk_attn_args, b = vk_attn_bwd_args */
    VK_OP_EMBED_FWD, VK_OP_EMBED_BWD, VK_OP_XENT_FWD,
    VK_OP_XENT_BWD,      /* a = vk_xent_args, b = dlogits, i0 = ldd, c = gscale */
    VK_OP_KL_FWD, VK_OP_KL_BWD,
    VK_OP_GENERIC,       /* a = vk_generic_args */
    /* Side-stream control.  Weight gradients are off the critical path of the backward pass (the reference computes
       them inside autograd's serial order, volta/encoders.py backward of every nn.Linear): the ops between SIDE_BEGIN
       and SIDE_END run on the library's side stream, which first waits for everything issued so far on the caller's
       stream; SIDE_END records event i0 (0..15) there, WAIT_SIDE makes the caller's stream wait for event i0 (no-op
       if never recorded), JOIN makes it wait for everything issued on the side stream. */
    VK_OP_SIDE_BEGIN, VK_OP_SIDE_END, VK_OP_WAIT_SIDE, VK_OP_JOIN,
    VK_OP_LN_FINALIZE,   /* a = vk_ln_bwd_args of the deferred vk_ln_bwd */
    VK_OP_GEMM_FP8,      /* a = vk_gemm_fp8_problem[i2], i1 = epilogue, i0 = geometry */
    VK_OP_GEMM_CHAIN     /* vk_gemm_chain: a = producers[i2 & 0xFF], b = consumers[i2 >> 8], i0 = layout, i1 = epi_p | epi_c << 8 */
};
enum {
    VK_FN_CAST = 1, VK_FN_MEMSET, VK_FN_LOC_FWD, VK_FN_LOC_BWD, VK_FN_ADD_DROPOUT, VK_FN_COLSUM, VK_FN_SELECT,
    VK_FN_GATHER, VK_FN_SCATTER_ADD, VK_FN_LOSS_FINAL, VK_FN_POOL_FWD, VK_FN_POOL_BWD, VK_FN_MASK_PREP, VK_FN_MUL,
    VK_FN_VLBERT_PREP, VK_FN_VLBERT_MASKGRAD, VK_FN_ROWGROUP_SUM, VK_FN_RELU_BWD, VK_FN_COPY, VK_FN_SUM_SLABS, VK_FN_SUM_SLABS_BF16,
    VK_FN_SIDE_TAIL,     /* p[0] = vk_tail_job[n[0]] */
    VK_FN_QUANT_ROWS,    /* vk_quant_rows_fp8(p[0], n[4], n[2], p[1], n[3], p[2], n[0], n[1], p[3]) */
    VK_FN_CAST_FP8,      /* vk_cast_bf16_fp8(p[0], p[1], n[0], f[0]) */
    VK_FN_VIS_LOSS_FWD,  /* vk_vis_loss_fwd(p[0]) */
    VK_FN_VIS_LOSS_BWD,  /* vk_vis_loss_bwd(p[0], p[1], n[0], p[2]) */
    VK_FN_NCE_NEG,       /* vk_nce_negatives(drop, n[0], n[1], p[0]) */
    VK_FN_TEXT_END_ROWS, /* vk_text_end_rows(p[0], n[0], n[1], p[1], p[2]) */
    VK_FN_VLBERT_OBJ_IDS, /* vk_vlbert_obj_ids(p[0], p[1], n[0], n[1]) */
    VK_FN_VLBERT_POSITIONS, /* vk_vlbert_positions(p[0], n[0], n[1], n[2], p[1], p[2]) */
    VK_FN_HOLD,          /* vk_hold_cus(n[0], n[1], n[2]) */
    VK_FN_GATE,          /* vk_gate_wait(p[0], p[1], n[0], p[2]) */
    VK_FN_BUMP           /* vk_bump_u64(p[0]) */
};                       /* VK_FN_POOL_FWD / VK_FN_POOL_BWD: n[3] = fusion mode (VK_FUSE_*) */
typedef struct vk_generic_args {   /* positional arguments of the small entry points, see executor.cpp */
    int32_t fn;
    void* p[6];
    int64_t n[6];
    float f[2];
    vk_dropout drop;
} vk_generic_args;
typedef struct vk_op {
    int32_t kind, i0, i1, i2;
    const void* a;
    const void* b;
    const void* c;
} vk_op;
int vk_run_ops(const vk_op* ops, int n, vk_stream_t s);
/* Same, bracketing each op with HIP events on `s`; synchronises `s` and adds elapsed ms per op to ms[0..n).
   Side-stream blocks run inline on `s` here, so that every op's duration is attributed to it. */
int vk_run_ops_timed(const vk_op* ops, int n, vk_stream_t s, float* ms);
/* Make `s` wait for everything issued so far on ITS side stream (what VK_OP_JOIN does).  Every caller stream has a side stream and
   events of its own (created on first use), so command lists replayed on different streams / from different threads share nothing. */
int vk_side_join(vk_stream_t s);
/* Make `waiter` (e.g. a communication stream) wait for the side stream that belongs to `owner` (the stream the lists run on). */
int vk_side_join_from(vk_stream_t owner, vk_stream_t waiter);
/* The side stream that belongs to `owner` (created on first use; NULL + vk_last_error() on failure): host code that opens further
   streams beside a command list (communication, a pipelined optimizer) probes them against this one -- HIP streams share a handful of
   hardware queues, and a stream that shares the side stream's queue serialises the weight gradients behind its own waits
   (volta_amd/streams.py, profiles/r04_experiments.md). */
vk_stream_t vk_side_stream(vk_stream_t owner);
/* 0: run side-stream blocks inline on the caller's stream (serial schedule); 1 (default): concurrently. */
void vk_side_enable(int on);

/* ------------------------------------------------------------------------------------------------
 * ConceptCap batch producer: raw records -> the model's input tensors with the reference's sampling policy.
 * Replaces BertPreprocessBatch.__call__ / convert_example_to_features / random_word / random_region / iou and the global-feature
 * code of ConceptCapLoaderTrain.__iter__ (volta/datasets/concept_cap_dataset.py:31-68, 229-286, 429-668) plus the objective-1
 * relabel of train_concap.py:279-284.  Decisions are words of Philox streams of `seed` (see csrc/concap.hip). */
typedef struct vk_concap_args {
    const int32_t* cap_tokens;   /* [n_caps, cap_ld] caption token ids without [CLS] / [SEP]            */
    const int32_t* cap_len;      /* [n_caps]                                                            */
    const int32_t* cap_index;    /* [B] caption of each pair                                            */
    const float* feat;           /* [B, R, F] region features (rows >= num_boxes are ignored)           */
    const float* cls;            /* [B, R, C] class distributions                                       */
    const float* boxes;          /* [B, R, 4] x1, y1, x2, y2 in pixels                                  */
    const int32_t* num_boxes;    /* [B]                                                                 */
    const float* img_wh;         /* [B, 2] image width, height                                          */
    int64_t* input_ids;          /* [B, T]                                                              */
    int64_t* input_mask;         /* [B, T]                                                              */
    int64_t* segment_ids;        /* [B, T]                                                              */
    int64_t* lm_label_ids;       /* [B, T]                                                              */
    int64_t* is_match;           /* [B]  1 = caption replaced                                           */
    float* image_feat;           /* [B, R + (add_global != 0), F]                                       */
    float* image_loc;            /* [B, R + (add_global != 0), 5]                                       */
    float* image_cls;            /* [B, R, C]                                                           */
    int64_t* image_label;        /* [B, R]                                                              */
    int64_t* image_mask;         /* [B, R + (add_global != 0)]                                          */
    uint64_t seed;
    int32_t B, T, R, F, C, n_caps, cap_ld, vocab_size, cls_id, sep_id, mask_id;
    int32_t add_global;          /* 0 none, 1 first, 2 last                                             */
    int32_t objective;           /* 0, 1 (mismatched pairs lose their MLM / region labels), 2 (no swaps) */
    int32_t visualization;       /* != 0: no caption swap, no token / region masking (BertPreprocessBatch(visualization=True), :514,622,652) */
} vk_concap_args;
int vk_concap_batch(const vk_concap_args* a, vk_stream_t s);

/* ------------------------------------------------------------------------------------------------
 * Record readers in front of the batch producer.  Host code (no stream argument): files are memory-mapped and fields
 * are decoded straight into the caller's staging slot -- use pinned memory and one cudaMemcpyAsync per batch.
 *
 * vk_lmdb_*: read-only access to an LMDB data file, replacing `lmdb.open(path, readonly=True, lock=False)` + `txn.get` / cursor iteration
 * (volta/datasets/_image_features_reader.py:46-58,83; tensorpack's LMDBSerializer.load at concept_cap_dataset.py:117-121,305-309).
 * `path` is the data file or the directory holding `data.mdb`.  Returned key / value pointers point into the mapping and stay valid
 * until vk_lmdb_close.  Main database only; DUPSORT / named sub-databases are refused. */
typedef struct vk_lmdb vk_lmdb;
int vk_lmdb_open(const char* path, vk_lmdb** out);
void vk_lmdb_close(vk_lmdb* db);
int64_t vk_lmdb_entries(const vk_lmdb* db);
int vk_lmdb_first(vk_lmdb* db);                                   /* rewind the cursor to the smallest key */
/* 1 = record returned and cursor advanced (keys in memcmp order), 0 = end, -1 = error */
int vk_lmdb_next(vk_lmdb* db, const void** key, size_t* klen, const void** val, size_t* vlen);
/* 1 = found, 0 = no such key, -1 = error */
int vk_lmdb_get(const vk_lmdb* db, const void* key, size_t klen, const void** val, size_t* vlen);

/* One Conceptual Captions datapoint = msgpack array of the 13 fields BertPreprocessBatch.__call__ unpacks (concept_cap_dataset.py:430-431):
 * features [nb, F], cls_prob [nb, C], obj_labels [nb], obj_confs [nb], attr_labels [nb], attr_confs [nb], attr_scores [nb, A], boxes [nb, 4],
 * num_boxes, img_h, img_w, img_id, caption; ndarrays in msgpack_numpy's encoding ({nd, type, kind, shape, data}).  Decodes into one slot of
 * the staging arrays vk_concap_batch reads: rows >= num_boxes are zero-filled (the reference stages into np.zeros, :433-446); a record with
 * more than R boxes is an error (the reference's assignment raises).  NULL destinations are skipped. */
typedef struct vk_concap_record {
    float* feat;            /* [R, F] */
    float* cls;             /* [R, C] */
    float* attr;            /* [R, A] or NULL */
    float* boxes;           /* [R, 4] pixels */
    int64_t* obj_labels;    /* [R] or NULL */
    float* obj_confs;       /* [R] or NULL */
    int64_t* attr_labels;   /* [R] or NULL */
    float* attr_confs;      /* [R] or NULL */
    int32_t R, F, C, A;
    int32_t num_boxes;      /* out */
    float img_w, img_h;     /* out */
    int32_t caption_len;    /* out */
    const char* caption;    /* out: UTF-8 bytes inside the record (not NUL-terminated) */
    char image_id[64];      /* out: NUL-terminated */
} vk_concap_record;
int vk_concap_record_decode(const void* rec, size_t len, vk_concap_record* r);
/* n datapoints into n slots on `threads` host threads (records are independent).  -1 with the first failing record's message;
   *failed (optional) = its index. */
int vk_concap_records_decode(const void* const* recs, const size_t* lens, vk_concap_record* slots, int n, int threads, int* failed);

/* BERT tokenisation of captions (basic tokenizer + WordPiece), replacing `self.tokenizer.encode(caption)` of BertPreprocessBatch.__call__
 * (concept_cap_dataset.py:465; tokenizer = pytorch-transformers 1.1 BertTokenizer, requirements.txt:39).  `vocab_path`: BERT's vocab.txt, one
 * token per line, id = line number.  Ids come without [CLS] / [SEP]. */
typedef struct vk_wordpiece vk_wordpiece;
int vk_wordpiece_open(const char* vocab_path, int lowercase, vk_wordpiece** out);
void vk_wordpiece_close(vk_wordpiece* t);
int vk_wordpiece_vocab_size(const vk_wordpiece* t);
int vk_wordpiece_token_id(const vk_wordpiece* t, const char* token);            /* -1 if absent */
/* returns the number of ids of `text` (UTF-8, len bytes); the first min(that, cap) are written */
int vk_wordpiece_encode(const vk_wordpiece* t, const char* text, size_t len, int32_t* ids, int cap);
/* n texts -> ids [n, ld] (zero-padded, truncated at ld), counts[i] = min(#ids, ld), on `threads` host threads */
int vk_wordpiece_encode_batch(const vk_wordpiece* t, const char* const* texts, const size_t* lens, int n, int32_t* ids, int ld, int32_t* counts, int threads);

/* base64 text (standard or url-safe alphabet, padding optional, line breaks skipped) -> bytes: the `boxes` / `features` / `cls_prob`
 * columns of the extraction TSV (data/conceptual_captions/preprocess_cc_train.py:66-68) and of the task feature stores
 * (_image_features_reader.py:87-88). */
int vk_b64_decode(const char* src, size_t n, void* dst, size_t cap, size_t* out_len);

#ifdef __cplusplus
}
#endif
#endif
