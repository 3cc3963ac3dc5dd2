// LDS-DMA fill rate of one CU by access pattern: the GEMM K loop stages 256-row strips of a K-contiguous operand.  With 32-deep K-steps a
// row contributes 64 bytes per step (one wave instruction = 16 rows x 64 B: sixteen HALF cache lines), with 64-deep K-steps 128 bytes
// (8 rows x 128 B: eight whole lines).  Same bytes per CU, same bytes in flight (64 KiB), no MFMA, no LDS reads: what does the path deliver?
//   hipcc --offload-arch=gfx950 -O2 dma_pattern.hip -o dma_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
#define LDSP __attribute__((address_space(3)))

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000); }

// DEEP = 32: pattern of the shipped kernel; 64: whole lines.  Each workgroup (8 waves) stages `strips` strips of 256 rows x DEEP k per step
// (2 strips = an A and a B strip of a 256 x 256 tile) into a ring of LDS slots and keeps two steps' worth (64 KiB at DEEP 32 x 2 strips) in flight.
#define FILL_KERNEL(NAME, DEEP, LPR, PIECES, STRIP, SLOTS)                                                                                  \
__global__ __launch_bounds__(512) void NAME(const char* base, int ld_bytes, int ksteps, int reps, int rows_per_wg, unsigned long long* cycles, int shared) { \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                                                             \
    const uint32_t lds0 = (uint32_t)(uintptr_t)(LDSP char*)smem;                                                                            \
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                           \
    const char* mine = base + (size_t)(shared ? (blockIdx.x & 7) : blockIdx.x) * rows_per_wg * ld_bytes;                                                                  \
    const __amdgpu_buffer_rsrc_t rs = rsrc(mine, (uint32_t)rows_per_wg * ld_bytes);                                                         \
    uint32_t off[PIECES];                                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < PIECES; ++i) {                                                                                    \
        const int lin = i * 512 + tid, r = lin / LPR, c = lin % LPR;                                                                        \
        off[i] = (uint32_t)r * (uint32_t)ld_bytes + (uint32_t)c * 16u;                                                                      \
    }                                                                                                                                       \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                                             \
    for (int rep = 0; rep < reps; ++rep) {                                                                                                  \
        int slot = 0;                                                                                                                       \
        for (int p = 0; p < ksteps; ++p) {                                                                                                  \
            _Pragma("unroll") for (int s = 0; s < 2; ++s)          /* two strips per step: rows [0, 256) and [256, 512) of the workgroup's block */ \
                _Pragma("unroll") for (int i = 0; i < PIECES; ++i) {                                                                        \
                    const uint32_t dst = lds0 + (uint32_t)slot * 2u * STRIP + s * STRIP + (uint32_t)(i * 512 + wave * 64) * 16u;            \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDSP void*)(uintptr_t)dst, 16, off[i] + (uint32_t)p * (DEEP * 2u) + (uint32_t)s * 256u * (uint32_t)ld_bytes, 0, 0, 0); \
                }                                                                                                                           \
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       /* 64 KiB stay in flight: two steps of 32, one step of 64 */               \
            slot = slot == SLOTS - 1 ? 0 : slot + 1;                                                                                        \
        }                                                                                                                                   \
    }                                                                                                                                       \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                                        \
    if (tid == 0) cycles[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;                                                                   \
}
FILL_KERNEL(fill32, 32, 4, 2, 16384u, 5)
FILL_KERNEL(fill64, 64, 8, 4, 32768u, 2)

int main() {
    const int G = 256, ld = 768 * 2, rows = 512;       // K = 768 bf16 rows; every workgroup its own 512 rows (A-like: 100 MB over the chip at 256 rows used)
    char* buf; unsigned long long* cyc;
    CK(hipMalloc(&buf, (size_t)G * rows * ld)); CK(hipMemset(buf, 1, (size_t)G * rows * ld)); CK(hipMalloc(&cyc, G * 8));
    CK(hipFuncSetAttribute((const void*)fill32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)fill64, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {             // 0: every byte once (HBM / Infinity Cache); 1: each workgroup's own 768 KiB 40 times (201 MB in all: Infinity Cache); 2: the 32 workgroups of an XCD share one 768 KiB block (L2)
        const int reps = mode == 0 ? 1 : 40, shared = mode == 2;
        for (int deep : {32, 64, 32, 64}) {
            const int ksteps = 768 / deep;
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (deep == 32) hipLaunchKernelGGL(fill32, dim3(G), dim3(512), 160 * 1024, 0, buf, ld, ksteps, reps, rows, cyc, shared);
            else hipLaunchKernelGGL(fill64, dim3(G), dim3(512), 160 * 1024, 0, buf, ld, ksteps, reps, rows, cyc, shared);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)G * reps * ksteps * 2.0 * 256 * deep * 2;
            printf("%s, K-step %2d deep (%3d B per row and step), %2d passes: %7.1f us, %6.1f GB/s per CU, %5.2f TB/s chip\n",
                   mode == 0 ? "cold, own block  " : mode == 1 ? "own 768 KiB block" : "block per XCD    ", deep, deep * 2, reps, ms * 1e3, bytes / G / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
