// Is hipExtAnyOrderLaunch honoured on gfx950?  K1: 256 workgroups that own a CU each (160 KiB of LDS) and spin for 20..90 us, then raise a
// flag; K2: the same footprint, records when each workgroup starts and when it sees K1's flag (bounded poll).  With the barrier bit cleared
// K2's first workgroups start while K1's long ones still run.   hipcc --offload-arch=gfx950 -O2 anyorder.hip -o anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

__global__ void k1(unsigned* flag, unsigned long long* t_end) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        smem[0] = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long span = 2000ull + 1000ull * (blockIdx.x & 7);       // 100 MHz ticks: 20 .. 90 us
        while (__builtin_amdgcn_s_memrealtime() - t0 < span) __builtin_amdgcn_s_sleep(8);
        t_end[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_store(flag + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void k2(const unsigned* flag, unsigned long long* t_start, unsigned long long* t_seen) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        smem[0] = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        t_start[blockIdx.x] = t0;
        unsigned long long t = t0;
        while (__hip_atomic_load(flag + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(8);
            t = __builtin_amdgcn_s_memrealtime();
            if (t - t0 > 200000ull) break;          // 2 ms: never hang
        }
        t_seen[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
}

int main() {
    const int G = 256, LDS = 160 * 1024;
    unsigned* flag; unsigned long long *te, *ts, *tv;
    CK(hipMalloc(&flag, G * 4)); CK(hipMalloc(&te, G * 8)); CK(hipMalloc(&ts, G * 8)); CK(hipMalloc(&tv, G * 8));
    CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipStream_t created, nonblocking; CK(hipStreamCreate(&created)); CK(hipStreamCreateWithFlags(&nonblocking, hipStreamNonBlocking));
    const char* snames[3] = {"created stream", "non-blocking stream", "NULL stream"};
    for (int si = 0; si < 3; ++si) {
    hipStream_t s = si == 0 ? created : si == 1 ? nonblocking : (hipStream_t)0;
    printf("-- %s\n", snames[si]);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(flag, 0, G * 4, s));
            CK(hipStreamSynchronize(s));
            hipLaunchKernelGGL(k1, dim3(G), dim3(64), LDS, s, flag, te);
            if (mode == 0) hipLaunchKernelGGL(k2, dim3(G), dim3(64), LDS, s, (const unsigned*)flag, ts, tv);
            else hipExtLaunchKernelGGL(k2, dim3(G), dim3(64), LDS, s, nullptr, nullptr, hipExtAnyOrderLaunch, (const unsigned*)flag, ts, tv);
            CK(hipStreamSynchronize(s));
            std::vector<unsigned long long> e(G), a(G), v(G);
            CK(hipMemcpy(e.data(), te, G * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(a.data(), ts, G * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(v.data(), tv, G * 8, hipMemcpyDeviceToHost));
            const unsigned long long e0 = *std::min_element(e.begin(), e.end()), e1 = *std::max_element(e.begin(), e.end());
            const unsigned long long a0 = *std::min_element(a.begin(), a.end()), a1 = *std::max_element(a.begin(), a.end());
            const unsigned long long v1 = *std::max_element(v.begin(), v.end());
            printf("%s rep %d: K1 ends %.1f .. %.1f us | K2 starts %.1f .. %.1f us, last flag seen %.1f us (all relative to K1's first end)\n",
                   mode ? "any-order" : "in-order ", rep, 0.0, (e1 - e0) * 0.01, ((double)a0 - (double)e0) * 0.01, ((double)a1 - (double)e0) * 0.01, ((double)v1 - (double)e0) * 0.01);
        }
    }
    }
    return 0;
}
