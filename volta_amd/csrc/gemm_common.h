// Pieces shared by the GEMM kernels of libvolta_hip.so (gemm.hip: 128^2 tiles and the dispatcher,
// gemm256.hip: the 256^2 8-phase kernel).
#pragma once
#include "common.h"
#include "../../include/volta_hip.h"
#include "util.h"

namespace vk {

constexpr int BK = 64;

struct KProb {
    const char* A; const char* B; char* C; char* C2; const float* bias; const char* R; float* bias_grad;
    const int32_t* dyn;
    int32_t M, N, K, lda, ldb, ldc, ldr, n_store;
    int32_t tiles_n, tile_start;
    char* C8; float c8_mul; int32_t ldc8;      // GELU epilogue (fp8 path): e4m3 copy of C, q = saturate(C * c8_mul); NULL otherwise
    char* ws; int32_t* cnt; int32_t part, nparts;   // split accumulation (vk_gemm_problem): nparts <= 1 = off
    int32_t* sig; const int32_t* dep; int32_t* err; int32_t dep_need;      // row-block hand-off between launches (vk_gemm_problem::sig / dep)
};
constexpr int GROUP_PLAIN_ORDER = 1 << 16;    // KGroup::stagger flag: workgroup i takes tile i (no XCD chunking)
struct KGroup {
    int32_t nprob;
    int32_t stagger;   // bits 0-7: DMA stagger of the legacy geometries, 8-15: ablation switches (study kernel), 16: GROUP_PLAIN_ORDER
    unsigned long long* retire_flag;            // vk_gemm_problem::retire_flag of problem 0: every workgroup stores *retire_stamp there as it retires
    const unsigned long long* retire_stamp;
    KProb p[VK_GEMM_MAX_GROUP];
};
// Last statement of a GEMM workgroup: tell a gate on another stream (vk_gate_wait) that this launch has begun to hand CUs back.
__device__ __forceinline__ void retire_mark(const KGroup& g) {
    if (g.retire_flag && threadIdx.x == 0) __hip_atomic_store(g.retire_flag, *g.retire_stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// XOR applied to the 16-byte chunk index of a row of a transposed ([k][cols]) image (low 4 bits only)
__device__ __forceinline__ int tswz(int row) { return ((row & 3) << 1) ^ (((row >> 3) & 1) << 3); }

// fragment for the 16 rows [r0, r0+16) of a K-contiguous image (128-byte rows), k-substep ks (32 wide)
__device__ __forceinline__ bf16x8 frag_rows(uint32_t tile, int r0, int ks, int lane) {
    const int r = r0 + (lane & 15);
    const int c = (ks * 4 + (lane >> 4)) ^ (r & 7);
    return *(const bf16x8 VK_LDS*)(uintptr_t)(tile + r * 128 + c * 16);
}
// fragment for the 16 columns [c0, c0+16) of a transposed image ([k][EXT], ROWB bytes per row); natural k order
template <int ROWB>
__device__ __forceinline__ bf16x8 frag_cols(uint32_t tile, int c0, int ks, int lane) {
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int row_a = ks * 32 + g * 8 + q;
    const int chunk = (c0 >> 3) + (p >> 1);
    const uint32_t a0 = tile + row_a * ROWB + ((chunk ^ tswz(row_a)) << 4) + ((p & 1) << 3);
    const int row_b = row_a + 4;
    const uint32_t a1 = tile + row_b * ROWB + ((chunk ^ tswz(row_b)) << 4) + ((p & 1) << 3);
    bf16x4 lo = lds_read_tr16(a0);
    bf16x4 hi = lds_read_tr16(a1);
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// XCD-aware tile order (cdna guide T1): workgroups are dealt round-robin over the 8 XCDs, so give XCD x the
// contiguous chunk x of the tile list -- neighbouring tiles (same A row panel) then share one L2.  Bijective for
// any grid size; placement only affects speed.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Epilogue shared by every geometry.  acc[i][j] is the 16 x 16 MFMA tile at rows m_base + 16 i, columns
// n_base + 16 j; with the swapped operands (D = B-frag x A-frag) a lane owns row (lane & 15) and the 4 consecutive
// columns 4 * (lane >> 4) .. +3 of each tile: 8-byte bf16 / 16-byte fp32 stores.
// Every global READ of the epilogue (bias, residual / multiplier R, accumulate-into C) is an unconditional
// bounds-checked buffer load -- absent operands and out-of-range rows / columns read as 0 -- so the compiler can
// issue them all up front instead of one dependent L2 round trip per element; only the stores are predicated.
// Extent of a K-contiguous operand row for the buffer bounds: raw buffer loads are range-checked per DWORD, so with an odd K the dword that
// holds element K-1 of the LAST row would count as out of range and read as zero (found with the one-column region-logit head: K = 1 lost
// the whole last row).  The row is extended to an even element count inside its leading dimension; the extra element is a pad column, which
// is zero by the ragged-K contract (lda padded to a multiple of 64, pad written as 0) or meets an out-of-range (zero) row of the other operand.
// 16-byte write-through (sc1) store: the line goes to the memory side at once and is dropped from the XCD's L2, so it is not among the
// dirty lines the end-of-kernel release has to write back (MI355X_MICROARCH.md, "stores of each flavour" / boundary row)
__device__ __forceinline__ void store16_wt(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// ---- row-block hand-off between two launches of one stream (include/volta_hip.h, vk_gemm_problem::sig / dep; VK_GEMM_SOFT_START) ----
// The form is MI355X_MICROARCH.md's "valid forms" table, third row: the producer's stores of the handed-off bytes are 16-byte write-through
// (sc1) stores of whole lines, every storing wave drains them (s_waitcnt vmcnt(0)), the workgroup meets at a barrier and ONE lane adds 1 to
// the row block's counter (agent scope); the consumer polls that counter with sc1 loads from ONE lane, the workgroup meets at a barrier
// behind the poll, and every load of the handed-off bytes is an sc1 load (the A operand's LDS-DMA loads carry aux = sc1).
// A poll that does not see its count within ~2^20 rounds (> 0.3 s: a mis-planned dependency) raises *err and lets the tile run on what is
// there -- a wrong result that the host can see, never a hung queue.
constexpr int AUX_SC1 = 16;
__device__ __forceinline__ void soft_wait(const int32_t* dep, int rb, int need, int32_t* err) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(dep + rb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1 << 20)) {
                if (err) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    // agent-scope acquire as the compiler's memory model spells it for this part (poll, then invalidate): no later load of this wave may be
    // served from a line that a cache of this XCD fetched before the producer's write-through stores landed
    asm volatile("buffer_inv sc1" ::: "memory");
}
__device__ __forceinline__ void soft_signal(int32_t* sig, int rb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave, before the barrier the adding lane passes
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(sig + rb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int even_up(int K, int ld) { const int e = (K + 1) & ~1; return e < ld ? e : ld; }

#ifndef VK_EPI_ABLATE      // A/B builds (tools/runs): 1 = GELU epilogue without the gelu' stores, 2 = with trivial arithmetic, 3 = both; 4 = x R / + R epilogues without the loads of R
#define VK_EPI_ABLATE 0
#endif
template <bool AT, int EPI, int TI, int TJ, int REGION = 16384>       // REGION: bytes of the wave-private LDS staging region
__device__ __forceinline__ void gemm_epilogue(const KProb& P, f32x4 (&acc)[TI][TJ], f32x4 (&accb)[TI], bool do_bias_grad,
                                              int m_base, int n_base, int M, int lane, uint32_t lds_region = 0) {
    const int gq = lane >> 4, lr = lane & 15;
    const int Mout = AT ? P.M : M;     // TN: M is the output row count and is never dynamic
    const int N = P.N;
    constexpr bool F32OUT = (EPI == VK_EPI_F32 || EPI == VK_EPI_F32_ACC);
    constexpr bool USE_R = (EPI == VK_EPI_MULR || EPI == VK_EPI_ADDR);
    const int nlim = (F32OUT && P.n_store > N) ? P.n_store : N;

    // Fast path: the wave's whole TI x TJ block of tiles lies inside the output -- straight-line code, no predicates.
    // With a wave-private 16 KiB LDS region the results are written as full rows: the MFMA layout gives a store
    // instruction 16 rows x 32 (64) bytes, which measured 5x below the write rate of a plain fill; staged through an
    // XOR-swizzled LDS image every store instruction covers whole 128-byte lines (8 / 4 rows x 128 / 256 bytes).
    if (m_base + 16 * TI <= Mout && n_base + 16 * TJ <= N) {
        char* const Cp = P.C;
        char* const C2p = P.C2;
        const char* const Rp = P.R;
        const int ldc = P.ldc, ldr = P.ldr;
        constexpr int ES = F32OUT ? 4 : 2;                     // bytes per output element
        constexpr int ROWB = TJ * 16 * ES;                     // bytes per image row
        constexpr bool POW2 = (TJ & (TJ - 1)) == 0;            // XOR swizzle needs 2^k chunks per row; otherwise rows are padded by 16 bytes
        constexpr int PITCH = POW2 ? ROWB : ROWB + 16;
        constexpr int NIMG = (EPI == VK_EPI_GELU) ? 2 : 1;
        constexpr int RPH0 = REGION / (PITCH * NIMG);          // rows per pass through the staging region
        constexpr int TIH = RPH0 >= TI * 16 ? TI : (RPH0 >= TI * 8 ? TI / 2 : TI / 4);    // row tiles per pass (divides TI)
        constexpr int RPH = TIH * 16;
        constexpr int CPR = ROWB / 16;                         // 16-byte chunks per row
        constexpr int RPI = 64 / CPR;                          // rows per store instruction
        constexpr int SWZ = POW2 ? (CPR < 8 ? CPR - 1 : 7) : 0; // XOR mask of the chunk swizzle (stays inside the row: 4 chunks for 256 x 128 bf16 tiles)
        static_assert(TIH >= 1 && RPH * PITCH * NIMG <= REGION, "epilogue staging does not fit its region");
        const bool via_lds = lds_region != 0 && ((ldc * ES) & 15) == 0 && (((uintptr_t)Cp | (uintptr_t)C2p) & 15) == 0;
        const bool wt = P.sig != nullptr;
        const uint32_t img2 = lds_region + RPH * PITCH;
        f32x4 b4[TJ];
        {
            const bool has_bias = (EPI != VK_EPI_MULR) && (P.bias != nullptr);
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(P.bias, has_bias ? (uint32_t)N * 4u : 0u);
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                b4[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (uint32_t)(n_base + j * 16 + gq * 4) * 4u, 0, 0));
        }
        // the residual / multiplier operand of the whole wave tile is requested up front: one L2 round trip instead of
        // one per row of tiles (2 VGPRs per tile; the K loop's fragment registers are free by now)
        // (the persistent kernel keeps the next tile's staging offsets live beside the accumulators: its 256-wide tiles request R in
        // two batches of TI / 2 rows of tiles instead of all at once, 32 instead of 64 VGPRs)
        constexpr int RB = (REGION < 16384 && TJ >= 4) ? TI / 2 : TI;
        u32x2 rall[USE_R ? RB : 1][TJ];
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if (USE_R && (i % RB) == 0) {
#pragma unroll
                for (int ii = 0; ii < RB; ++ii) {
                    const char* rrow = Rp + ((size_t)(m_base + (i + ii) * 16 + lr) * ldr + (size_t)(n_base + gq * 4)) * 2;
#pragma unroll
                    for (int j = 0; j < TJ; ++j) rall[ii][j] = (VK_EPI_ABLATE & 4) ? u32x2{0x3F803F80u, 0x3F803F80u} : *(const u32x2*)(rrow + j * 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int m = m_base + i * 16 + lr;
            const int lrow = (i % TIH) * 16 + lr;              // row inside the LDS image
            const size_t rowc = (size_t)m * ldc + (size_t)(n_base + gq * 4);
            u32x2 rv[TJ];
            f32x4 cv[TJ];
            if (USE_R) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) rv[j] = rall[i % RB][j];
            }
            if (EPI == VK_EPI_F32_ACC) {
#pragma unroll
                for (int j = 0; j < TJ; ++j) cv[j] = *(const f32x4*)(Cp + (rowc + j * 16) * 4);
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                f32x4 v = acc[i][j] + b4[j];
                if (F32OUT) {
                    if (EPI == VK_EPI_F32_ACC) v += cv[j];
                    if (via_lds) *(f32x4 VK_LDS*)(uintptr_t)(lds_region + lrow * PITCH + (((j * 4 + gq) ^ (lrow & SWZ)) << 4)) = v;
                    else *(f32x4*)(Cp + (rowc + j * 16) * 4) = v;
                    continue;
                }
                float w[4] = {0.f, 0.f, 0.f, 0.f};
                if (USE_R) { w[0] = bf2f(rv[j][0] & 0xFFFF); w[1] = bf2f(rv[j][0] >> 16); w[2] = bf2f(rv[j][1] & 0xFFFF); w[3] = bf2f(rv[j][1] >> 16); }
                float o[4], o2[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (EPI == VK_EPI_BF16) o[r] = v[r];
                    else if (EPI == VK_EPI_GELU && (VK_EPI_ABLATE & 2)) { o[r] = v[r]; o2[r] = 0.5f * v[r]; }
                    else if (EPI == VK_EPI_GELU) { gelu_both(v[r], o[r], o2[r]); }
                    else if (EPI == VK_EPI_MULR) o[r] = v[r] * w[r];
                    else if (EPI == VK_EPI_ADDR) o[r] = v[r] + w[r];
                    else o[r] = fmaxf(v[r], 0.f);
                }
                const u32x2 pk = u32x2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
                if (EPI == VK_EPI_GELU && P.C8)        // fp8 copy for the FFN-down projection: 4 bytes per lane, straight from the registers
                    *(uint32_t*)(P.C8 + (size_t)m * P.ldc8 + (size_t)(n_base + j * 16 + gq * 4)) = pack4_fp8(o[0] * P.c8_mul, o[1] * P.c8_mul, o[2] * P.c8_mul, o[3] * P.c8_mul);
                if (via_lds) {
                    const uint32_t a = lrow * PITCH + (((j * 2 + (gq >> 1)) ^ (lrow & SWZ)) << 4) + ((gq & 1) << 3);
                    *(u32x2 VK_LDS*)(uintptr_t)(lds_region + a) = pk;
                    if (EPI == VK_EPI_GELU) *(u32x2 VK_LDS*)(uintptr_t)(img2 + a) = u32x2{pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3])};
                } else {
                    *(u32x2*)(Cp + (rowc + j * 16) * 2) = pk;
                    if (EPI == VK_EPI_GELU) *(u32x2*)(C2p + (rowc + j * 16) * 2) = u32x2{pack2bf(o2[0], o2[1]), pack2bf(o2[2], o2[3])};
                }
            }
            if (via_lds && (i % TIH) == TIH - 1) {
                // read the image(s) back row-major and store whole lines
                const int rr0 = lane / CPR, ch = lane % CPR;
                const int mrow0 = m_base + (i / TIH) * RPH;
#pragma unroll
                for (int q = 0; q < (RPH + RPI - 1) / RPI; ++q) {
                    const int row = q * RPI + rr0;
                    if ((RPI * CPR < 64 && rr0 >= RPI) || ((RPH % RPI) && row >= RPH)) continue;
                    const uint32_t a = row * PITCH + ((ch ^ (row & SWZ)) << 4);
                    const size_t g = ((size_t)(mrow0 + row) * ldc + (size_t)n_base) * ES + (size_t)ch * 16;
                    if (AT && EPI == VK_EPI_F32) __builtin_nontemporal_store(*(const u32x4 VK_LDS*)(uintptr_t)(lds_region + a), (u32x4*)(Cp + g));      // weight-gradient slab: read once, by the tail launch
                    else if (wt) store16_wt(Cp + g, *(const u32x4 VK_LDS*)(uintptr_t)(lds_region + a));      // handed to a consumer tile by counter (soft_signal): write-through
                    else *(u32x4*)(Cp + g) = *(const u32x4 VK_LDS*)(uintptr_t)(lds_region + a);      // (write-through for EVERY output measured neutral: profiles/r04_experiments.md)
                    // gelu'(u) is read again only by the backward pass: stored non-temporally so that it does not push the activation beside
                    // it -- the next GEMM's A operand -- out of the Infinity Cache
                    if (VK_EPI_ABLATE & 1) continue;
                    if (EPI == VK_EPI_GELU) __builtin_nontemporal_store(*(const u32x4 VK_LDS*)(uintptr_t)(img2 + a), (u32x4*)(C2p + g));
                }
            }
        }
        if (do_bias_grad && gq == 0) {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                const int m = m_base + i * 16 + lr;
                P.bias_grad[m] = (EPI == VK_EPI_F32_ACC ? P.bias_grad[m] : 0.f) + accb[i][0];
            }
        }
        return;
    }

    float bv[TJ][4];
    {
        const bool has_bias = (EPI != VK_EPI_MULR) && (P.bias != nullptr);
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(P.bias, has_bias ? (uint32_t)N * 4u : 0u);
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                bv[j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (uint32_t)(n_base + j * 16 + gq * 4 + r) * 4u, 0, 0));
    }
    const __amdgpu_buffer_rsrc_t rr = make_rsrc(P.R, (USE_R && Mout > 0) ? (uint32_t)(((uint32_t)(Mout - 1) * P.ldr + N) * 2u) : 0u);
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(P.C, (EPI == VK_EPI_F32_ACC && Mout > 0) ? (uint32_t)(((uint32_t)(Mout - 1) * P.ldc + nlim) * 4u) : 0u);

#pragma unroll
    for (int i = 0; i < TI; ++i) {
        __builtin_amdgcn_sched_barrier(0);     // one row of tiles at a time: keeps the epilogue's live registers below the K loop's
        const int m = m_base + i * 16 + lr;
        const bool row_ok = m < Mout;
        u32x2 rv[TJ];
        u32x4 cv[TJ];
        if (USE_R) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int n = n_base + j * 16 + gq * 4;
                const uint32_t off = (row_ok && n < N) ? ((uint32_t)m * (uint32_t)P.ldr + (uint32_t)n) * 2u : 0xFFFFFFF0u;
                rv[j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rr, off, 0, 0));
            }
        }
        if (EPI == VK_EPI_F32_ACC) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int n = n_base + j * 16 + gq * 4;
                const uint32_t off = (row_ok && n < nlim) ? ((uint32_t)m * (uint32_t)P.ldc + (uint32_t)n) * 4u : 0xFFFFFFF0u;
                if (n + 3 < nlim || !row_ok) cv[j] = __builtin_amdgcn_raw_buffer_load_b128(rc, off, 0, 0);
                else {                                     // ragged last columns: per-dword bounds
#pragma unroll
                    for (int r = 0; r < 4; ++r) cv[j][r] = (n + r < nlim) ? __builtin_amdgcn_raw_buffer_load_b32(rc, off + 4u * r, 0, 0) : 0u;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int n = n_base + j * 16 + gq * 4;
            float v[4] = {acc[i][j][0] + bv[j][0], acc[i][j][1] + bv[j][1], acc[i][j][2] + bv[j][2], acc[i][j][3] + bv[j][3]};
            const size_t off = (size_t)m * P.ldc + n;
            const bool full = (n + 3 < nlim);
            if (F32OUT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (n + r >= N) v[r] = 0.f;
                if (EPI == VK_EPI_F32_ACC) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += __builtin_bit_cast(float, cv[j][r]);
                }
                if (!row_ok || n >= nlim) continue;
                float* c = (float*)P.C + off;
                if (full) *(f32x4*)c = f32x4{v[0], v[1], v[2], v[3]};
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < nlim) c[r] = v[r];
                continue;
            }
            float w[4] = {0.f, 0.f, 0.f, 0.f};
            if (USE_R) { w[0] = bf2f(rv[j][0] & 0xFFFF); w[1] = bf2f(rv[j][0] >> 16); w[2] = bf2f(rv[j][1] & 0xFFFF); w[3] = bf2f(rv[j][1] >> 16); }
            float o[4], o2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (EPI == VK_EPI_BF16) o[r] = v[r];
                else if (EPI == VK_EPI_GELU) { gelu_both(v[r], o[r], o2[r]); }
                else if (EPI == VK_EPI_MULR) o[r] = v[r] * w[r];
                else if (EPI == VK_EPI_ADDR) o[r] = v[r] + w[r];
                else o[r] = fmaxf(v[r], 0.f);
            }
            if (!row_ok || n >= nlim) continue;
            uint16_t* c = (uint16_t*)P.C + off;
            if (EPI == VK_EPI_GELU && P.C8) {
                uint8_t* c8 = (uint8_t*)P.C8 + (size_t)m * P.ldc8 + n;
                const uint32_t w8 = pack4_fp8(o[0] * P.c8_mul, o[1] * P.c8_mul, o[2] * P.c8_mul, o[3] * P.c8_mul);
                if (full) *(uint32_t*)c8 = w8;
                else
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < N) c8[r] = (uint8_t)(w8 >> (8 * r));
            }
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
        for (int i = 0; i < TI; ++i) {
            const int m = m_base + i * 16 + lr;
            if (m < Mout) P.bias_grad[m] = (EPI == VK_EPI_F32_ACC ? P.bias_grad[m] : 0.f) + accb[i][0];
        }
    }
}

// ---- split accumulation: nparts workgroups hold K-slices of one output tile (include/volta_hip.h, vk_gemm_problem::ws) -------------
// Every workgroup stores its accumulators into its slab of `ws` with write-through (sc1) 16-byte stores, drains them
// (s_waitcnt vmcnt(0) in every wave, workgroup barrier) and ONE lane draws a ticket from the tile's counter with an agent-scope
// atomic add; the workgroup whose add came last (ticket = nparts - 1) reads all slabs back with sc1 loads -- in part order, its own
// included, so that the fp32 sum does not depend on who arrived last -- resets the counter and returns true: it runs the epilogue.
// (The hand-off form is MI355X_MICROARCH.md's "valid forms" table, first row: one lane per storing workgroup adds to one counter, the
// adder that came last is told by the value its add returned, its other waves load behind a workgroup barrier; sc1 stores and loads of
// 16 bytes, hipMalloc memory, one workgroup per CU.)  Nobody waits for anybody: the launch completes under any residency.
// Slab layout: [wave][value index][lane] f32x4, value index = i * TJ + j (accumulator tiles), then TI bias-gradient tiles when BG.
template <int TI, int TJ, bool BG> constexpr uint32_t split_slab_bytes(int waves) { return (uint32_t)waves * (uint32_t)(TI * TJ + (BG ? TI : 0)) * 1024u; }

template <int TI, int TJ, bool BG, int WAVES>
__device__ __forceinline__ bool split_combine(const KProb& P, int tile, f32x4 (&acc)[TI][TJ], f32x4 (&accb)[TI], int wave, int lane, uint32_t lds_word) {
    constexpr int NV = TI * TJ + (BG ? TI : 0);
    constexpr uint32_t WAVE_BYTES = NV * 1024u, SLAB = WAVES * WAVE_BYTES;
    char* const tile_ws = P.ws + (size_t)tile * (size_t)P.nparts * SLAB;
    const uint32_t base = (uint32_t)wave * WAVE_BYTES + (uint32_t)lane * 16u;
    {
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(tile_ws + (size_t)P.part * SLAB, SLAB);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rw, base + (uint32_t)(i * TJ + j) * 1024u, 0, 16 /* sc1: write-through */);
        if (BG) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, accb[i]), rw, base + (uint32_t)(TI * TJ + i) * 1024u, 0, 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave, before the barrier the ticket lane passes
    __syncthreads();
    volatile uint32_t VK_LDS* const word = (volatile uint32_t VK_LDS*)(uintptr_t)lds_word;
    if (threadIdx.x == 0) {
        const int ticket = __hip_atomic_fetch_add(P.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *word = (uint32_t)ticket;                             // the add has returned: its value is used
    }
    __syncthreads();
    const int ticket = (int)*word;
    if (ticket != P.nparts - 1) return false;
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (BG) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma nounroll
    for (int q = 0; q < P.nparts; ++q) {
        const __amdgpu_buffer_rsrc_t rq = make_rsrc(tile_ws + (size_t)q * SLAB, SLAB);
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            u32x4 v[TJ];
#pragma unroll
            for (int j = 0; j < TJ; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rq, base + (uint32_t)(i * TJ + j) * 1024u, 0, 16 /* sc1 */);
#pragma unroll
            for (int j = 0; j < TJ; ++j) acc[i][j] += __builtin_bit_cast(f32x4, v[j]);
        }
        if (BG) {
#pragma unroll
            for (int i = 0; i < TI; ++i)
                accb[i] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rq, base + (uint32_t)(TI * TJ + i) * 1024u, 0, 16));
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(P.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    return true;
}

// 256-row tiles, 8 waves, LDS-DMA ring (gemm256.hip).  variant 4 / 3 / 2: K-split kernel with 256 / 192 / 128 columns; 0: the 4-phase
// study kernel (VK_STUDY builds only).  persistent: one workgroup per CU walks the tile list (NT / NN, no device-side row counts).
constexpr int NUM_CU = 256;       // MI355X
int launch_gemm256(int layout, int epilogue, const KGroup& g, int total, hipStream_t s, int variant, bool persistent, bool soft = false);
// one persistent launch for a producer group (256 x 256 tiles) and the consumer group that reads its outputs (256 x 192 tiles)
int launch_gemm256_chain(int layout, int epi_p, int epi_c, const KGroup& g, int nprod, int ncons, hipStream_t s);
// 4-wave ring kernels (gemm4w.hip): bm = 256: 256 x 128 tiles, 72 KiB LDS, two workgroups per CU; bm = 128: 128 x 128 tiles, ring of 6 K-steps
int launch_gemm4w(int layout, int epilogue, const KGroup& g, int total, hipStream_t s, int bm);

}  // namespace vk
