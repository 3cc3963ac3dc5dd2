// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of libvolta_hip.so.
// wave = 64 lanes everywhere; bf16 is carried as raw uint16 bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vk {

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one 16x16x32 / 32x32x16 MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define VK_LDS __attribute__((address_space(3)))

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {      // one v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2_t));
}

// four fp32 -> four OCP e4m3 bytes (round to nearest even, saturating at +-448)
__device__ __forceinline__ uint32_t pack4_fp8(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.0f, 448.0f); b = __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f);
    c = __builtin_amdgcn_fmed3f(c, -448.0f, 448.0f); d = __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f);
    uint32_t w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return w;
}

// ---- wave reductions (64 lanes) -------------------------------------------------------------
// DPP forms: four in-row steps (quad xor 1, quad xor 2, half-row mirror, row mirror: every lane of a 16-lane row ends with the row's
// result), two cross-row broadcasts (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 holds the wave's result) and
// one v_readlane -- ~8 VALU instructions instead of six dependent ds_bpermute round trips through the LDS crossbar (what __shfl_xor
// compiles to), which is most of a LayerNorm row's latency.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float v, float old) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1, 0xF>(v, v);        // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xF>(v, v);        // quad_perm [2,3,0,1]
    v += dpp_move<0x141, 0xF>(v, v);       // row_half_mirror
    v += dpp_move<0x140, 0xF>(v, v);       // row_mirror
    v += dpp_move<0x142, 0xA>(v, 0.f);     // row_bcast:15 -> rows 1, 3
    v += dpp_move<0x143, 0xC>(v, 0.f);     // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x142, 0xA>(v, v));      // unmasked rows keep their own value: max is idempotent
    v = fmaxf(v, dpp_move<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---- Philox-4x32: the engine's counter-based random streams ------------------------------------
// Dropout contract (mirrored by oracle/volta_ref.py:philox_u32): a dropout site sees its tensor as
// [rows, C]; element (row, c) takes word (c & 3) of Philox-4x32-7(counter = (c >> 2, row, site, 0),
// key = (seed_lo, seed_hi)); it is KEPT iff word >= floor(p * 2^32) and scaled by 1/(1-p).
// 7 rounds is the smallest Crush-resistant Philox-4x32 (Salmon et al., SC'11, table 2); the dropout kernels
// (LayerNorm, attention) are VALU-bound on this generator, and the 10-round form cost them 40-75 % of their
// instructions.  The sampling policy of the batch producer (concap.hip) keeps the 10-round default form.
// The products are written as v_mul_hi_u32 + v_mul_lo_u32 pairs on purpose: the 64-bit form (v_mad_u64_u32) needs
// aligned register pairs, which pushed the attention forward kernel from 121 to 134 VGPRs (3 instead of 4 waves
// per SIMD) and made it 25 % slower.
template <int ROUNDS>
__device__ __forceinline__ u32x4 philox4_rounds(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u32x4 r = {c0, c1, c2, c3};
    return r;
}
constexpr int DROPOUT_PHILOX_ROUNDS = 7;
// the dropout stream
__device__ __forceinline__ u32x4 philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    return philox4_rounds<DROPOUT_PHILOX_ROUNDS>(c0, c1, c2, c3, k0, k1);
}

struct DropCfg {              // by-value kernel argument
    const uint64_t* seed;     // device word, bumped once per step by vk_step_begin (graph-replay safe)
    uint32_t site;            // dropout site id (position in the reference's forward order)
    uint32_t thr;             // floor(p * 2^32); 0 => dropout off
    float scale;              // 1/(1-p)
};

__device__ __forceinline__ u32x4 drop_words(const DropCfg& d, uint64_t seed, uint32_t row, uint32_t c4) {
    return philox4(c4, row, d.site, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// ---- buffer resources (bounds-checked loads: out-of-range reads return 0) ---------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// erf-GELU and its derivative (volta/encoders.py:130-136).  erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7,
// far below the bf16 output rounding); exp(-x^2/2) is shared between erf and the Gaussian density of the
// derivative, so value + derivative cost one exp, one rcp and ~12 FMAs instead of two libm erff calls.
__device__ __forceinline__ void gelu_both(float x, float& y, float& dy) {
    // constants folded by hand (the compiler may not reassociate): exp(-x^2 / 2) = exp2(x^2 * (-log2(e) / 2)) -- v_exp_f32 IS exp2 --, and the
    // polynomial carries the 1/2 of cdf = 1/2 + 1/2 erf: 3 multiplies and 2 adds fewer per element of a VALU-bound epilogue (r04_experiments.md 8)
    const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170f);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(fabsf(x), 0.3275911f * 0.70710678118654752f, 1.0f));   // bare v_rcp_f32 (1 ulp): __frcp_rn expands to the 10-instruction IEEE division
    const float poly = ((((0.5f * 1.061405429f * t - 0.5f * 1.453152027f) * t + 0.5f * 1.421413741f) * t - 0.5f * 0.284496736f) * t + 0.5f * 0.254829592f) * t;
    const float half_erf = __builtin_fmaf(-poly, e, 0.5f);       // erf(|x| / sqrt 2) / 2
    const float cdf = 0.5f + copysignf(half_erf, x);
    y = x * cdf;
    dy = __builtin_fmaf(x * 0.39894228040143268f, e, cdf);
}
__device__ __forceinline__ float gelu_f(float x) { float y, d; gelu_both(x, y, d); return y; }
__device__ __forceinline__ float gelu_grad_f(float x) { float y, d; gelu_both(x, y, d); return d; }

// ds_read_b64_tr_b16: within each group of 16 lanes, lane 4q+p supplies the address of row q,
// elements 4p..4p+3 of a 4x16 block of 16-bit values; lane i receives column i (rows 0..3).
__device__ __forceinline__ bf16x4 lds_read_tr16(uint32_t byte_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((VK_LDS bf16x4*)(uintptr_t)byte_addr);
}

}  // namespace vk
