// fp64 vector latency / issue rate on one CU of gfx950 as a function of the number of active waves in the workgroup
#include <hip/hip_runtime.h>
#include <cstdio>
#define TICK(x) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int n, int nact) {
    const int t = threadIdx.x, wave = t >> 6;
    long long t0, t1; double acc = 0;
    __syncthreads();
    if (wave < nact) {
        { double a = 1.0 + t * 1e-9, b = 1.0000001; TICK(t0); for (int i = 0; i < n; ++i) a = fma(a, b, 1e-9); TICK(t1); if (t == 0) cyc[0] = t1 - t0; acc += a; }
        { double a0 = 1, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8; const double b = 1.0000001; TICK(t0);
          for (int i = 0; i < n; ++i) { a0 = fma(a0, b, 1e-9); a1 = fma(a1, b, 1e-9); a2 = fma(a2, b, 1e-9); a3 = fma(a3, b, 1e-9); a4 = fma(a4, b, 1e-9); a5 = fma(a5, b, 1e-9); a6 = fma(a6, b, 1e-9); a7 = fma(a7, b, 1e-9); }
          TICK(t1); if (t == 0) cyc[1] = t1 - t0; acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }
        { float a = 1.0f + t * 1e-6f, b = 1.0001f; TICK(t0); for (int i = 0; i < n; ++i) a = fmaf(a, b, 1e-6f); TICK(t1); if (t == 0) cyc[2] = t1 - t0; acc += a; }
        { double x = 2.0 + t; TICK(t0); for (int i = 0; i < n; ++i) { double y = __builtin_amdgcn_rsq(x); x = fma(y, 1e-3, x); } TICK(t1); if (t == 0) cyc[3] = t1 - t0; acc += x; }
        { double a = 1.0 + t * 1e-9; TICK(t0); for (int i = 0; i < n; ++i) a = a * 1.0000001; TICK(t1); if (t == 0) cyc[4] = t1 - t0; acc += a; }
        { double a = 1.0 + t * 1e-9; TICK(t0); for (int i = 0; i < n; ++i) a = a + 1.0000001; TICK(t1); if (t == 0) cyc[5] = t1 - t0; acc += a; }
    }
    out[t] = acc;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 512); hipMalloc(&cyc, 8 * 16);
    const int n = 2000;
    for (int nact : {1, 2, 4, 8}) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, n, nact); hipDeviceSynchronize(); }
        long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("active waves %d: dep fma_f64 %.1f | 8 indep fma_f64 (per fma) %.1f | dep fma_f32 %.1f | rsq_f64+fma %.1f | dep mul_f64 %.1f | dep add_f64 %.1f ticks\n", nact,
               (double)h[0] / n, (double)h[1] / n / 8, (double)h[2] / n, (double)h[3] / n, (double)h[4] / n, (double)h[5] / n);
    }
    // the same with 64 workgroups on other CUs busy (clock state)
    return 0;
}
