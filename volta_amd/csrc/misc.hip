// Small utility kernels: seed word, dtype casts, fills.
#include "common.h"
#include "../../include/volta_hip.h"
#include "util.h"

namespace vk {

__global__ void set_seed_kernel(uint64_t* p, uint64_t v) { *p = v; }

// fp32 -> bf16, 8 elements per lane (32-B loads, 16-B stores); n8 = n / 8, scalar tail after it
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n) {
    const size_t n8 = n >> 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 a = __builtin_nontemporal_load((const f32x4*)(src + i * 8)), b = __builtin_nontemporal_load((const f32x4*)(src + i * 8 + 4));
        *(u32x4*)(dst + i * 8) = u32x4{pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[n8 * 8 + threadIdx.x] = f2bf(src[n8 * 8 + threadIdx.x]);
}

// out = a * b over rows * row_len bf16 elements (8 per lane)
__global__ void mul_bf16_kernel(const uint16_t* a, const uint16_t* b, uint16_t* out, size_t n, const int32_t* dyn_rows, int row_len) {
    if (dyn_rows) { const size_t lim = (size_t)(*dyn_rows) * row_len; if (lim < n) n = lim; }
    const size_t n8 = n >> 3;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 x = *(const u32x4*)(a + i * 8), y = *(const u32x4*)(b + i * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack2bf(bf2f(x[k] & 0xFFFF) * bf2f(y[k] & 0xFFFF), bf2f(x[k] >> 16) * bf2f(y[k] >> 16));
        *(u32x4*)(out + i * 8) = o;
    }
}

}  // namespace vk

extern "C" int vk_mul_bf16(const void* a, const void* b, void* out, int64_t n, const int32_t* dyn_rows, int row_len, vk_stream_t s) {
    if (n <= 0) return 0;
    if (n % 8 || row_len % 8) return vk::set_error("vk_mul_bf16: n and row_len must be multiples of 8");
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vk::mul_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const uint16_t*)a, (const uint16_t*)b,
                       (uint16_t*)out, (size_t)n, dyn_rows, row_len);
    return vk::check_launch("vk_mul_bf16");
}

extern "C" int vk_set_seed(uint64_t* seed_dev, uint64_t seed, vk_stream_t s) {
    hipLaunchKernelGGL(vk::set_seed_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, seed_dev, seed);
    return vk::check_launch("vk_set_seed");
}

extern "C" int vk_cast_f32_bf16(const float* src, void* dst, int64_t n, vk_stream_t s) {
    if (n <= 0) return 0;
    if (((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return vk::set_error("vk_cast_f32_bf16: 16-byte alignment required");
    int64_t blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(vk::cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, src, (uint16_t*)dst, (size_t)n);
    return vk::check_launch("vk_cast_f32_bf16");
}

// ---- measurement aid (tools/bench_gemm.py peak): sustained rate of v_mfma_f32_16x16x32_bf16 on register operands,
// 8 waves per CU, 16 independent accumulators per wave -- the ceiling any GEMM main loop on this chip can approach.
namespace vk {
__global__ __launch_bounds__(512) void mfma_peak_kernel(const uint32_t* seed, float* out, int iters) {
    typedef __attribute__((ext_vector_type(8))) short v8;
    typedef __attribute__((ext_vector_type(4))) float v4;
    v8 a[4], b[4];
    const uint32_t s = seed[threadIdx.x & 63] + blockIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a[i][k] = (short)(0x3C00 + ((s * (i * 8 + k + 1) * 2654435761u) >> 22));      // bf16 values of varying mantissa
            b[i][k] = (short)(0xBC00 + ((s * (i * 8 + k + 77) * 40503u) >> 22));
        }
    v4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = v4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 12345.678f) out[0] = t;
}
}  // namespace vk

extern "C" int vk_mfma_peak(const uint32_t* seed64, float* out, int iters, int blocks, vk_stream_t stream) {
    hipLaunchKernelGGL(vk::mfma_peak_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, seed64, out, iters);
    return vk::check_launch("vk_mfma_peak");
}

// ---- `nwg` one-wave workgroups that stay resident for `usec` microseconds; whole_cu: each declares all 160 KiB of LDS, i.e. owns a CU.  The
// hand-off tests (tests/test_gemm_gpu.py: soft boundaries under UNEVEN load) and tools/comm_footprint.py run it on a second stream.
namespace vk {
__global__ void hold_cus_kernel(unsigned ticks) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        smem[0] = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    }
}
}  // namespace vk

extern "C" int vk_hold_cus(int nwg, int usec, int whole_cu, vk_stream_t stream) {
    if (nwg <= 0 || usec <= 0) return 0;
    if (nwg > 256 || usec > 100000) return vk::set_error("vk_hold_cus: at most 256 workgroups for at most 0.1 s (got %d, %d us)", nwg, usec);
    constexpr int LDS = 160 * 1024;
    static const hipError_t attr = hipFuncSetAttribute((const void*)vk::hold_cus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); (void)attr;
    hipLaunchKernelGGL(vk::hold_cus_kernel, dim3(nwg), dim3(64), whole_cu ? LDS : 64, (hipStream_t)stream, (unsigned)usec * 100u);
    return vk::check_launch("vk_hold_cus");
}

// ---- measurement aid: the CU footprint of a communication library's channel kernels, without the library (one-GPU boxes cannot run RCCL with
// more than one rank).  `nwg` workgroups of 256 threads stream `bytes` from src to dst (16-byte non-temporal accesses, each workgroup a
// contiguous slice) and then stay resident until `min_usec` have passed since the first of them started -- a channel kernel lives as long
// as its transfer does, at link rate, not at HBM rate.  stamps (optional, 2 x uint64): first start / last end, s_memrealtime (100 MHz).
namespace vk {
__global__ __launch_bounds__(256) void comm_standin_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16, unsigned min_ticks,
                                                           unsigned long long* stamps) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (stamps && threadIdx.x == 0) atomicMin(stamps, t0);
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    if (threadIdx.x == 0) {
        const unsigned long long first = stamps ? __hip_atomic_load(stamps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : t0;
        while (__builtin_amdgcn_s_memrealtime() - first < min_ticks) __builtin_amdgcn_s_sleep(32);
        if (stamps) atomicMax(stamps + 1, __builtin_amdgcn_s_memrealtime());
    }
}
}  // namespace vk

extern "C" int vk_comm_standin(const void* src, void* dst, int64_t bytes, int nwg, int min_usec, uint64_t* stamps, vk_stream_t stream) {
    if (bytes <= 0 || nwg <= 0) return 0;
    if ((bytes & 15) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15)) return vk::set_error("vk_comm_standin: 16-byte granularity");
    if (nwg > 1024 || min_usec < 0 || min_usec > 100000) return vk::set_error("vk_comm_standin: at most 1024 workgroups, at most 0.1 s");
    hipLaunchKernelGGL(vk::comm_standin_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, (const vk::u32x4*)src, (vk::u32x4*)dst, (size_t)(bytes / 16),
                       (unsigned)min_usec * 100u, (unsigned long long*)stamps);
    return vk::check_launch("vk_comm_standin");
}

// ---- gate: one wave that holds its stream until another stream's launch has stored the stamp (include/volta_hip.h, vk_gate_wait)
namespace vk {
__global__ void gate_wait_kernel(const unsigned long long* flag, const unsigned long long* stamp, unsigned timeout_ticks, int32_t* err) {
    if (threadIdx.x != 0) return;
    const unsigned long long want = __hip_atomic_load(stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(16);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
            if (err) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}
__global__ void bump_u64_kernel(unsigned long long* w) { *w += 1ull; }
__global__ void store_u64_kernel(unsigned long long* w, unsigned long long v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void gate_value_kernel(const unsigned long long* flag, unsigned long long want, unsigned timeout_ticks, int32_t* err) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(16);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
            if (err) __hip_atomic_store(err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}
}  // namespace vk

extern "C" int vk_store_u64(uint64_t* word, uint64_t value, vk_stream_t stream) {
    if (!word) return vk::set_error("vk_store_u64: NULL");
    hipLaunchKernelGGL(vk::store_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)word, (unsigned long long)value);
    return vk::check_launch("vk_store_u64");
}

extern "C" int vk_gate_value(const uint64_t* flag, uint64_t want, int timeout_us, int32_t* err, vk_stream_t stream) {
    if (!flag || timeout_us <= 0 || timeout_us > 10000000) return vk::set_error("vk_gate_value: flag and 0 < timeout_us <= 10000000");
    hipLaunchKernelGGL(vk::gate_value_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned long long*)flag, (unsigned long long)want,
                       (unsigned)(timeout_us > 40000000 ? 4000000000u : (unsigned)timeout_us * 100u), err);
    return vk::check_launch("vk_gate_value");
}

extern "C" int vk_gate_wait(const uint64_t* flag, const uint64_t* stamp, int timeout_us, int32_t* err, vk_stream_t stream) {
    if (!flag || !stamp || timeout_us <= 0 || timeout_us > 1000000) return vk::set_error("vk_gate_wait: flag, stamp and 0 < timeout_us <= 1000000");
    hipLaunchKernelGGL(vk::gate_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned long long*)flag, (const unsigned long long*)stamp,
                       (unsigned)timeout_us * 100u, err);
    return vk::check_launch("vk_gate_wait");
}

extern "C" int vk_bump_u64(uint64_t* word, vk_stream_t stream) {
    if (!word) return vk::set_error("vk_bump_u64: NULL");
    hipLaunchKernelGGL(vk::bump_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)word);
    return vk::check_launch("vk_bump_u64");
}
