// How early does an any-order kernel's first workgroup start when the kernel in front of it finishes UNEVENLY inside every XCD?
// K1: 256 CU-owning workgroups (160 KiB LDS); `nlong` workgroups per XCD spin 80 us, the others 20 us.  K2: 256 CU-owning workgroups, any-order.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

__global__ void k1(unsigned long long* t_end, int nlong, unsigned* xcc) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        smem[0] = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long span = ((int)(blockIdx.x >> 3) < nlong) ? 8000ull : 2000ull;
        while (__builtin_amdgcn_s_memrealtime() - t0 < span) __builtin_amdgcn_s_sleep(8);
        t_end[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id & 0xF;
    }
}
__global__ void k2(unsigned long long* t_start, unsigned* xcc) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        smem[0] = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        t_start[blockIdx.x] = t0;
        unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = id & 0xF;
        while (__builtin_amdgcn_s_memrealtime() - t0 < 500ull) __builtin_amdgcn_s_sleep(8);
    }
}

int main() {
    const int G = 256, LDS = 160 * 1024;
    unsigned long long *te, *ts; unsigned *x1, *x2;
    CK(hipMalloc(&te, G * 8)); CK(hipMalloc(&ts, G * 8)); CK(hipMalloc(&x1, G * 4)); CK(hipMalloc(&x2, G * 4));
    CK(hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int k2lds = 0; k2lds < 2; ++k2lds)
    for (int nlong : {0, 1, 4, 16, 31}) {
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k1, dim3(G), dim3(64), LDS, s, te, nlong, x1);
        hipExtLaunchKernelGGL(k2, dim3(G), dim3(64), k2lds ? LDS : 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ts, x2);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> e(G), a(G); std::vector<unsigned> xa(G), xb(G);
        CK(hipMemcpy(e.data(), te, G * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(a.data(), ts, G * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(xa.data(), x1, G * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(xb.data(), x2, G * 4, hipMemcpyDeviceToHost));
        const unsigned long long e0 = *std::min_element(e.begin(), e.end());
        std::vector<double> st(G);
        for (int i = 0; i < G; ++i) st[i] = ((double)a[i] - (double)e0) * 0.01;
        std::sort(st.begin(), st.end());
        int same = 0;
        for (int i = 0; i < G; ++i) same += xa[i] == (unsigned)(xa[0] + i) % 8 ? 1 : 0;
        printf("K2 %s LDS, %2d long (80 us) K1 workgroups per XCD of 32: K1 ends 0 .. %.1f us | K2 starts (sorted) #0 %.1f  #32 %.1f  #128 %.1f  #224 %.1f  #255 %.1f | K1 wg i on XCD (x0+i)%%8: %d/256\n",
               k2lds ? "160K" : "no  ", nlong, ((double)*std::max_element(e.begin(), e.end()) - (double)e0) * 0.01, st[0], st[32], st[128], st[224], st[255], same);
    }
    return 0;
}
