// Command-list executor: a training step is a static list of kernel launches with pre-bound arguments
// (all activation / weight / gradient buffers are fixed for a given batch shape), built once by the host
// plan builder and replayed here with no per-launch interpreter or autograd overhead (~3 us per launch on
// the host instead of ~15 us through an eager framework op).  The same entry points are exported one by
// one (include/volta_hip.h), so every list element can also be issued and checked in isolation.
#include "util.h"
#include <vector>
#include "../../include/volta_hip.h"

static int run_one(const vk_op& o, int i, vk_stream_t s);

// ---- side streams: one (with its events) per CALLER stream, created on first use and kept for the life of the process.  Two host
// threads replaying command lists on two streams therefore never share a side stream or an event (re-entrant across streams); the
// map itself is guarded by a mutex.  vk_side_enable() is the only process-wide switch (profiling: run side blocks inline).
#include <cstdlib>
#include <mutex>
#include <unordered_map>
namespace {
constexpr int N_SIDE_EVENTS = 16;
struct Side {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t done[N_SIDE_EVENTS] = {};
    bool recorded[N_SIDE_EVENTS] = {};
    bool ok = false;
};
std::mutex g_side_mutex;
std::unordered_map<void*, Side*> g_sides;
int g_side_enabled = 1;

Side* side_lookup(vk_stream_t caller) {
    std::lock_guard<std::mutex> lock(g_side_mutex);
    auto it = g_sides.find((void*)caller);
    return it == g_sides.end() ? nullptr : it->second;
}

Side* side_for(vk_stream_t caller) {
    std::lock_guard<std::mutex> lock(g_side_mutex);
    Side*& sp = g_sides[(void*)caller];
    if (sp && sp->ok) return sp;
    if (!sp) sp = new Side();
    Side& g = *sp;
    // (default priority: the device's lowest priority for this stream measured no better -- 17.89 / 17.98 against 17.80 / 17.91 ms per step)
    {
        // VK_SIDE_PRIORITY=low: the side stream's hardware queue at the device's lowest priority (an experiment knob; see profiles/r04_experiments.md)
        const char* pr = getenv("VK_SIDE_PRIORITY");
        int least = 0, greatest = 0;
        hipError_t e;
        if (pr && pr[0] == 'l' && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) e = hipStreamCreateWithPriority(&g.stream, hipStreamNonBlocking, least);
        else e = hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking);
        if (e != hipSuccess) { vk::set_error("side stream: hipStreamCreate failed"); return nullptr; }
    }
    if (hipEventCreateWithFlags(&g.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g.join, hipEventDisableTiming) != hipSuccess) {
        vk::set_error("side stream: hipEventCreate failed");
        return nullptr;
    }
    for (int i = 0; i < N_SIDE_EVENTS; ++i)
        if (hipEventCreateWithFlags(&g.done[i], hipEventDisableTiming) != hipSuccess) { vk::set_error("side stream: hipEventCreate failed"); return nullptr; }
    g.ok = true;
    return sp;
}
}  // namespace

extern "C" void vk_side_enable(int on) { g_side_enabled = on; }

extern "C" int vk_side_join_from(vk_stream_t owner, vk_stream_t waiter) {
    Side* g = side_lookup(owner);
    if (!g || !g->ok) return 0;           // nothing was ever issued on this stream's side stream
    if (hipEventRecord(g->join, g->stream) != hipSuccess) return vk::set_error("vk_side_join: hipEventRecord failed");
    if (hipStreamWaitEvent((hipStream_t)waiter, g->join, 0) != hipSuccess) return vk::set_error("vk_side_join: hipStreamWaitEvent failed");
    return 0;
}

extern "C" int vk_side_join(vk_stream_t s) { return vk_side_join_from(s, s); }

extern "C" vk_stream_t vk_side_stream(vk_stream_t owner) {
    Side* g = side_for(owner);
    return g ? (vk_stream_t)g->stream : nullptr;
}

extern "C" int vk_run_ops(const vk_op* ops, int n, vk_stream_t s) {
    vk_stream_t cur = s;
    Side* g = nullptr;
    for (int i = 0; i < n; ++i) {
        const vk_op& o = ops[i];
        if (o.kind >= VK_OP_SIDE_BEGIN && o.kind <= VK_OP_JOIN) {
            if (!g_side_enabled) continue;
            if (!g && !(g = side_for(s))) return -1;
            hipError_t e = hipSuccess;
            switch (o.kind) {
                case VK_OP_SIDE_BEGIN:
                    if (o.i0 != 1) {          // i0 == 1: no event -- the block opens with a gate (VK_FN_GATE) that a launch of the caller's stream releases
                        e = hipEventRecord(g->fork, (hipStream_t)s);
                        if (e == hipSuccess) e = hipStreamWaitEvent(g->stream, g->fork, 0);
                    }
                    cur = (vk_stream_t)g->stream;
                    break;
                case VK_OP_SIDE_END:
                    if (o.i0 < 0 || o.i0 >= N_SIDE_EVENTS) return vk::set_error("vk_run_ops: side event %d out of range", o.i0);
                    e = hipEventRecord(g->done[o.i0], g->stream);
                    g->recorded[o.i0] = true;
                    cur = s;
                    break;
                case VK_OP_WAIT_SIDE:
                    if (o.i0 < 0 || o.i0 >= N_SIDE_EVENTS) return vk::set_error("vk_run_ops: side event %d out of range", o.i0);
                    if (g->recorded[o.i0]) e = hipStreamWaitEvent((hipStream_t)s, g->done[o.i0], 0);
                    break;
                default:
                    if (vk_side_join(s) != 0) return -1;
            }
            if (e != hipSuccess) return vk::set_error("vk_run_ops: side-stream control op %d failed at index %d: %s", o.kind, i, hipGetErrorString(e));
            continue;
        }
        // a gate holds a SIDE stream until a launch of the caller's stream releases it: run inline (serial schedule) it would wait for a launch
        // that is listed behind it on the same stream
        if (o.kind == VK_OP_GENERIC && o.a && ((const vk_generic_args*)o.a)->fn == VK_FN_GATE && cur == s) continue;
        int rc = run_one(o, i, cur);
        if (rc != 0) return rc;
    }
    if (cur != s) return vk::set_error("vk_run_ops: op list ends inside a side-stream block");
    return 0;
}

// Profiling variant: brackets every op with HIP events on the launch stream, synchronises once at the end and
// ADDS each op's elapsed milliseconds to ms[i].  Used by bench.py for the live per-kernel-class timings.
extern "C" int vk_run_ops_timed(const vk_op* ops, int n, vk_stream_t s, float* ms) {
    static thread_local std::vector<hipEvent_t> ev;      // profiling helper: events are reused by the calling thread
    while ((int)ev.size() < n + 1) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return vk::set_error("vk_run_ops_timed: hipEventCreate failed");
        ev.push_back(e);
    }
    hipStream_t st = (hipStream_t)s;
    (void)hipEventRecord(ev[0], st);
    for (int i = 0; i < n; ++i) {
        const bool gate = ops[i].kind == VK_OP_GENERIC && ops[i].a && ((const vk_generic_args*)ops[i].a)->fn == VK_FN_GATE;
        if (!(ops[i].kind >= VK_OP_SIDE_BEGIN && ops[i].kind <= VK_OP_JOIN) && !gate) {     // control ops and gates: nothing to launch, block runs inline
            int rc = run_one(ops[i], i, s);
            if (rc != 0) return rc;
        }
        (void)hipEventRecord(ev[i + 1], st);
    }
    if (hipStreamSynchronize(st) != hipSuccess) return vk::set_error("vk_run_ops_timed: stream sync failed");
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, ev[i], ev[i + 1]);
        ms[i] += t;
    }
    return 0;
}

static int run_one(const vk_op& o, int i, vk_stream_t s) {
    {
        int rc = 0;
        switch (o.kind) {
            case VK_OP_GEMM: rc = vk_gemm_grouped_ex(o.i0 & 0xFF, o.i1, (const vk_gemm_problem*)o.a, o.i2, o.i0 >> 8, s); break;      /* i0: layout | tile geometry << 8 (0 = heuristic) */
            case VK_OP_GEMM_FP8: rc = vk_gemm_fp8_grouped(o.i1, (const vk_gemm_fp8_problem*)o.a, o.i2, o.i0, s); break;
            case VK_OP_GEMM_CHAIN: rc = vk_gemm_chain(o.i0, o.i1 & 0xFF, (const vk_gemm_problem*)o.a, o.i2 & 0xFF, o.i1 >> 8, (const vk_gemm_problem*)o.b, o.i2 >> 8, s); break;
            case VK_OP_LN_FWD: rc = vk_ln_fwd_pair((const vk_ln_args*)o.a, (const vk_ln_args*)o.b, s); break;
            case VK_OP_LN_BWD: rc = vk_ln_bwd_pair((const vk_ln_bwd_args*)o.a, (const vk_ln_bwd_args*)o.b, s); break;
            case VK_OP_LN_FINALIZE: rc = vk_ln_bwd_finalize((const vk_ln_bwd_args*)o.a, s); break;
            case VK_OP_ATTN_FWD: rc = vk_gated_attn_fwd((const vk_attn_args*)o.a, s); break;
            case VK_OP_ATTN_BWD: rc = vk_gated_attn_bwd((const vk_attn_args*)o.a, (const vk_attn_bwd_args*)o.b, s); break;
            case VK_OP_EMBED_FWD: rc = vk_embed_sum_fwd((const vk_embed_args*)o.a, s); break;
            case VK_OP_EMBED_BWD: rc = vk_embed_sum_bwd((const vk_embed_bwd_args*)o.a, s); break;
            case VK_OP_XENT_FWD: rc = vk_xent_fwd((const vk_xent_args*)o.a, s); break;
            case VK_OP_XENT_BWD: rc = vk_xent_bwd((const vk_xent_args*)o.a, (void*)o.b, o.i0, (const float*)o.c, s); break;
            case VK_OP_KL_FWD: rc = vk_kl_fwd((const vk_kl_args*)o.a, s); break;
            case VK_OP_KL_BWD: rc = vk_kl_bwd((const vk_kl_args*)o.a, (void*)o.b, o.i0, (const float*)o.c, s); break;
            case VK_OP_GENERIC: {
                const vk_generic_args* g = (const vk_generic_args*)o.a;
                switch (g->fn) {
                    case VK_FN_CAST: rc = vk_cast_f32_bf16((const float*)g->p[0], g->p[1], g->n[0], s); break;
                    case VK_FN_MEMSET: rc = vk_memset_async(g->p[0], (int)g->n[1], g->n[0], s); break;
                    case VK_FN_LOC_FWD: rc = vk_loc_linear_fwd((const float*)g->p[0], (const float*)g->p[1], (const float*)g->p[2], g->p[3], (int)g->n[0], (int)g->n[1], (int)g->n[2], s); break;
                    case VK_FN_LOC_BWD: rc = vk_loc_linear_bwd(g->p[0], (const float*)g->p[1], (float*)g->p[2], (float*)g->p[3], (float*)g->p[4], (int)g->n[0], (int)g->n[1], (int)g->n[2], s); break;
                    case VK_FN_ADD_DROPOUT: rc = vk_add_dropout(g->p[0], g->p[1], g->p[2], (int)g->n[0], (int)g->n[1], g->f[0], g->drop, (int)g->n[2], s); break;
                    case VK_FN_COLSUM: rc = vk_colsum_bf16(g->p[0], (float*)g->p[1], (float*)g->p[2], (int)g->n[0], (int)g->n[1], (int)g->n[2], s); break;
                    case VK_FN_SELECT: rc = vk_select_rows((const int64_t*)g->p[0], (int)g->n[0], (int)g->n[1], (int)g->n[2], (int)g->n[3], (int)g->n[4], (int32_t*)g->p[1], (int32_t*)g->p[2], (int32_t*)g->p[3], s); break;
                    case VK_FN_GATHER: rc = vk_gather_rows(g->p[0], (const int32_t*)g->p[1], (const int32_t*)g->p[2], g->p[3], (int)g->n[0], (int)g->n[1], s); break;
                    case VK_FN_SCATTER_ADD: rc = vk_scatter_rows_add(g->p[0], (const int32_t*)g->p[1], (const int32_t*)g->p[2], g->p[3], (int)g->n[0], (int)g->n[1], s); break;
                    case VK_FN_LOSS_FINAL: rc = vk_loss_finalize((const float*)g->p[0], (const int32_t*)g->p[1], (const int32_t*)g->p[2], (int)g->n[0], g->f[0], (float*)g->p[3], s); break;
                    case VK_FN_POOL_FWD: rc = vk_pool_fuse_fwd(g->p[0], g->p[1], g->p[2], (int)g->n[0], (int)g->n[1], (int)g->n[3], g->drop, s); break;
                    case VK_FN_POOL_BWD: rc = vk_pool_fuse_bwd(g->p[0], (int)g->n[2], g->p[1], g->p[2], g->p[3], g->p[4], (int)g->n[0], (int)g->n[1], (int)g->n[3], g->drop, s); break;
                    case VK_FN_VIS_LOSS_FWD: rc = vk_vis_loss_fwd((const vk_vis_loss_args*)g->p[0], s); break;
                    case VK_FN_VIS_LOSS_BWD: rc = vk_vis_loss_bwd((const vk_vis_loss_args*)g->p[0], g->p[1], (int)g->n[0], (const float*)g->p[2], s); break;
                    case VK_FN_NCE_NEG: rc = vk_nce_negatives(g->drop, (int)g->n[0], (int)g->n[1], (int32_t*)g->p[0], s); break;
                    case VK_FN_TEXT_END_ROWS: rc = vk_text_end_rows((const int64_t*)g->p[0], (int)g->n[0], (int)g->n[1], (int32_t*)g->p[1], (int32_t*)g->p[2], s); break;
                    case VK_FN_VLBERT_OBJ_IDS: rc = vk_vlbert_obj_ids((const int32_t*)g->p[0], (int64_t*)g->p[1], (int)g->n[0], (int)g->n[1], s); break;
                    case VK_FN_VLBERT_POSITIONS: rc = vk_vlbert_positions((const int64_t*)g->p[0], (int)g->n[0], (int)g->n[1], (int)g->n[2], (int64_t*)g->p[1], (int64_t*)g->p[2], s); break;
                    case VK_FN_HOLD: rc = vk_hold_cus((int)g->n[0], (int)g->n[1], (int)g->n[2], s); break;
                    case VK_FN_GATE: rc = vk_gate_wait((const uint64_t*)g->p[0], (const uint64_t*)g->p[1], (int)g->n[0], (int32_t*)g->p[2], s); break;
                    case VK_FN_BUMP: rc = vk_bump_u64((uint64_t*)g->p[0], s); break;
                    case VK_FN_MASK_PREP: rc = vk_mask_prep((const int64_t*)g->p[0], (float*)g->p[1], (int)g->n[0], s); break;
                    case VK_FN_MUL: rc = vk_mul_bf16(g->p[0], g->p[1], g->p[2], g->n[0], (const int32_t*)g->p[3], (int)g->n[1], s); break;
                    case VK_FN_VLBERT_PREP: rc = vk_vlbert_prep_fwd((const float*)g->p[0], (int)g->n[3], (const float*)g->p[1], (const float*)g->p[2], g->p[3], (int32_t*)g->p[4], (int)g->n[0], (int)g->n[1], (int)g->n[2], g->drop, s); break;
                    case VK_FN_VLBERT_MASKGRAD: rc = vk_vlbert_maskgrad(g->p[0], (int)g->n[2], (int)g->n[3], (const int32_t*)g->p[1], (float*)g->p[2], (float*)g->p[3], (int)g->n[0], (int)g->n[1], g->drop, s); break;
                    case VK_FN_ROWGROUP_SUM: rc = vk_rowgroup_sum_bf16(g->p[0], g->p[1], (int)g->n[0], (int)g->n[1], (int)g->n[2], s); break;
                    case VK_FN_RELU_BWD: rc = vk_relu_bwd_bf16(g->p[0], g->p[1], g->p[2], g->n[0], s); break;
                    case VK_FN_COPY: rc = vk_copy_async(g->p[0], g->p[1], g->n[0], s); break;
                    case VK_FN_SUM_SLABS_BF16: rc = vk_sum_slabs_bf16(g->p[0], (const float*)g->p[1], g->n[0], (int)g->n[1], g->n[2], (const int32_t*)g->p[2], (int)g->n[3], s); break;
                    case VK_FN_QUANT_ROWS: rc = vk_quant_rows_fp8(g->p[0], (int)g->n[4], g->n[2], g->p[1], g->n[3], (float*)g->p[2], (int)g->n[0], (int)g->n[1], (const int32_t*)g->p[3], s); break;
                    case VK_FN_CAST_FP8: rc = vk_cast_bf16_fp8(g->p[0], g->p[1], g->n[0], g->f[0], s); break;
                    case VK_FN_SIDE_TAIL: rc = vk_side_tail((const vk_tail_job*)g->p[0], (int)g->n[0], s); break;
                    case VK_FN_SUM_SLABS: rc = vk_sum_slabs_f32((float*)g->p[0], (const float*)g->p[1], g->n[0], (int)g->n[1], g->n[2], s); break;
                    default: rc = vk::set_error("vk_run_ops: unknown generic fn %d at op %d", g->fn, i);
                }
                break;
            }
            default: rc = vk::set_error("vk_run_ops: unknown op kind %d at index %d", o.kind, i);
        }
        return rc;
    }
}
