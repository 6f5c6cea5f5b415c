// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on one SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(double* out, long long* cyc, int n) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < n; ++i) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_fma(double* out, long long* cyc, int n) {
    double a0 = 0, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-9;
    long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < n; ++i) {
        a0 = fma(a0, y, x); a1 = fma(a1, y, x); a2 = fma(a2, y, x); a3 = fma(a3, y, x);
        a4 = fma(a4, y, x); a5 = fma(a5, y, x); a6 = fma(a6, y, x); a7 = fma(a7, y, x);
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 1024 * 64); hipMalloc(&cyc, 8 * 64);
    long long h[4];
    const int n = 2000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n); hipDeviceSynchronize();
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n); hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
        printf("mfma_f64_16x16x4: %d waves/CU: %.1f cycles per MFMA per wave (s_memtime ticks)\n", waves, (double)h[0] / (4.0 * n));
        hipLaunchKernelGGL(k_fma, dim3(1), dim3(64 * waves), 0, 0, out, cyc, n); hipDeviceSynchronize();
        hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
        printf("v_fma_f64       : %d waves/CU: %.2f cycles per wave-FMA\n", waves, (double)h[0] / (8.0 * n));
    }
    // wall-clock check of the s_memtime unit
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, out, cyc, 200000); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("s_memtime ticks per us: %.1f\n", (double)h[0] / (ms * 1e3));
    return 0;
}
