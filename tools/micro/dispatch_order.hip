// Does a kernel launched with hipExtAnyOrderLaunch ever get a workgroup onto a CU before EVERY workgroup of the kernel in front of it (same
// stream) has started?  K1: 512 workgroups that own a CU each (160 KiB LDS) -> two rounds on 256 CUs, each spins 30 us.  K2 (any-order):
// 1024 small workgroups (no LDS) that would fit beside K1's at any time.  In-order workgroup dispatch <=> min(K2 start) >= max(K1 start).
// Also the uneven case: a third stream holds 64 CUs for 200 us while K1 is launched.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

__global__ void spin_k(unsigned long long* t_start, unsigned long long* t_end, unsigned span) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        smem[0] = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        t_start[blockIdx.x] = t0;
        while (__builtin_amdgcn_s_memrealtime() - t0 < span) __builtin_amdgcn_s_sleep(8);
        t_end[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
}

int main() {
    const int LDS = 160 * 1024, G1 = 512, G2 = 1024, G3 = 64;
    unsigned long long *s1, *e1, *s2, *e2, *s3, *e3;
    CK(hipMalloc(&s1, G1 * 8)); CK(hipMalloc(&e1, G1 * 8)); CK(hipMalloc(&s2, G2 * 8)); CK(hipMalloc(&e2, G2 * 8)); CK(hipMalloc(&s3, G3 * 8)); CK(hipMalloc(&e3, G3 * 8));
    CK(hipFuncSetAttribute((const void*)spin_k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipStream_t s, side; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&side));
    for (int uneven = 0; uneven < 2; ++uneven)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            if (uneven) hipLaunchKernelGGL(spin_k, dim3(G3), dim3(64), LDS, side, s3, e3, 20000u);      // 64 CUs held for 200 us
            hipLaunchKernelGGL(spin_k, dim3(G1), dim3(64), LDS, s, s1, e1, 3000u);
            hipExtLaunchKernelGGL(spin_k, dim3(G2), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, s2, e2, 500u);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> a(G1), b(G1), c(G2);
            CK(hipMemcpy(a.data(), s1, G1 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), e1, G1 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), s2, G2 * 8, hipMemcpyDeviceToHost));
            const unsigned long long a0 = *std::min_element(a.begin(), a.end()), a1 = *std::max_element(a.begin(), a.end()), b1 = *std::max_element(b.begin(), b.end());
            const unsigned long long c0 = *std::min_element(c.begin(), c.end()), c1 = *std::max_element(c.begin(), c.end());
            int early = 0;
            for (int i = 0; i < G2; ++i) early += c[i] < a1;
            printf("%s rep %d: K1 starts 0.0 .. %.1f us, last end %.1f | K2 starts %.1f .. %.1f us | K2 workgroups started before K1's last start: %d\n",
                   uneven ? "64 CUs held" : "idle chip  ", rep, (a1 - a0) * 0.01, (b1 - a0) * 0.01, ((double)c0 - (double)a0) * 0.01, ((double)c1 - (double)a0) * 0.01, early);
        }
    return 0;
}
