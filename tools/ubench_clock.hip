// does a busy chip clock higher?  time a fixed dependent-FMA chain (ns via s_memtime, a constant 2.4 GHz-ish timer)
// alone and while a burner kernel keeps the other CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#define TICK(x) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
__global__ void probe(double* out, long long* cyc, int n) {
    double a = 1.0 + threadIdx.x * 1e-9; const double b = 1.0000001; long long t0, t1;
    TICK(t0); for (int i = 0; i < n; ++i) a = fma(a, b, 1e-9); TICK(t1);
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void burner(double* out, int n) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3; const double b = 1.0000001;
    for (int i = 0; i < n; ++i) { a0 = fma(a0, b, 1e-9); a1 = fma(a1, b, 1e-9); a2 = fma(a2, b, 1e-9); a3 = fma(a3, b, 1e-9); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
int main() {
    double *o1, *o2; long long* cyc; hipMalloc(&o1, 8 * 1024); hipMalloc(&o2, 8 * 1024 * 1024); hipMalloc(&cyc, 8);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    long long h; const int n = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, sa, o1, cyc, n); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("alone           : %.2f ticks per dependent fma_f64\n", (double)h / n);
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(burner, dim3(240 * 4), dim3(256), 0, sb, o2, 4000000);     // ~ hundreds of ms on 240 CUs
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, sa, o1, cyc, n);
        hipStreamSynchronize(sa); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("with burner     : %.2f ticks per dependent fma_f64\n", (double)h / n);
        hipDeviceSynchronize();
    }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, sa, o1, cyc, n); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("right after burn: %.2f ticks per dependent fma_f64\n", (double)h / n);
    return 0;
}
