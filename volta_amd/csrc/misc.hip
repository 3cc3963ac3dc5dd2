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

// ---- test / measurement aid: `nwg` workgroups that own a CU each (they declare all 160 KiB of LDS) for `usec` microseconds.  The
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

extern "C" int vk_hold_cus(int nwg, int usec, vk_stream_t stream) {
    if (nwg <= 0 || usec <= 0) return 0;
    if (nwg > 256 || usec > 100000) return vk::set_error("vk_hold_cus: at most 256 workgroups for at most 0.1 s (got %d, %d us)", nwg, usec);
    constexpr int LDS = 160 * 1024;
    static const hipError_t attr = hipFuncSetAttribute((const void*)vk::hold_cus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); (void)attr;
    hipLaunchKernelGGL(vk::hold_cus_kernel, dim3(nwg), dim3(64), LDS, (hipStream_t)stream, (unsigned)usec * 100u);
    return vk::check_launch("vk_hold_cus");
}
