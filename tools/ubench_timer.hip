// calibrates s_memtime: against s_memrealtime (constant 100 MHz) and against a full-rate VALU stream (4 shader cycles per wave64 fp32 op)
#include <hip/hip_runtime.h>
#include <cstdio>
#define TICK(x) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
#define RTICK(x) asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
__global__ void k(float* out, long long* cyc, int n) {
    long long t0, t1, r0, r1;
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; const float b = 1.0001f;
    TICK(t0); RTICK(r0);
    for (int i = 0; i < n; ++i) { a0 = fmaf(a0, b, 1e-6f); a1 = fmaf(a1, b, 1e-6f); a2 = fmaf(a2, b, 1e-6f); a3 = fmaf(a3, b, 1e-6f); a4 = fmaf(a4, b, 1e-6f); a5 = fmaf(a5, b, 1e-6f); a6 = fmaf(a6, b, 1e-6f); a7 = fmaf(a7, b, 1e-6f); }
    TICK(t1); RTICK(r1);
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}
int main() {
    float* o; long long* cyc; (void)hipMalloc(&o, 4 * 1024); (void)hipMalloc(&cyc, 16);
    long long h[2];
    for (int n : {20000, 200000, 2000000, 20000000}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, cyc, n);
        (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        const double real_us = h[1] / 100.0;
        printf("n=%9d: %.0f us (realtime), %.0f us (events): s_memtime %.1f MHz; one wave64 fp32 fma every %.2f memtime ticks = %.2f ns -> shader clock %.0f MHz if full rate (4 cycles)\n",
               n, real_us, ms * 1e3, h[0] / real_us, (double)h[0] / n / 8, real_us * 1e3 / n / 8, 4.0 / (real_us * 1e3 / n / 8) * 1e3);
    }
    return 0;
}
