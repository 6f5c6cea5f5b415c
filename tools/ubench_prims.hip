// micro-benchmark of the primitives the single-workgroup step kernel is made of (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define TICK(x) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int n) {
    extern __shared__ double lds[];
    const int t = threadIdx.x;
    for (int i = t; i < 14000; i += blockDim.x) lds[i] = (double)((i * 7 + 3) % 1000);
    __syncthreads();
    long long t0, t1;
    double acc = 0; int idx = t & 63;
    // (a) barrier loop
    TICK(t0); for (int i = 0; i < n; ++i) __syncthreads(); TICK(t1);
    if (t == 0) cyc[0] = t1 - t0;
    // (b) dependent LDS read chain (pointer chasing), wave 0 only
    if (t < 64) { TICK(t0); for (int i = 0; i < n; ++i) { idx = (int)lds[idx] ; } TICK(t1); if (t == 0) cyc[1] = t1 - t0; acc += idx; }
    __syncthreads();
    // (c) dependent fp64 FMA chain
    { double a = 1.0 + t * 1e-9, b = 1.0000001; TICK(t0); for (int i = 0; i < n; ++i) a = fma(a, b, 1e-9); TICK(t1); if (t == 0) cyc[2] = t1 - t0; acc += a; }
    // (d) dependent readlane + fma chain
    { double a = 1.0 + t * 1e-9; TICK(t0); for (int i = 0; i < n; ++i) { int lo = __builtin_amdgcn_readlane(__double2loint(a), 3), hi = __builtin_amdgcn_readlane(__double2hiint(a), 3); a = fma(__hiloint2double(hi, lo), 1.0000001, a * 1e-9); } TICK(t1); if (t == 0) cyc[3] = t1 - t0; acc += a; }
    // (e) LDS write then read same wave (store->load forwarding latency)
    if (t < 64) { double a = t; TICK(t0); for (int i = 0; i < n; ++i) { lds[t] = a; a = lds[(t + 1) & 63] + 1.0; } TICK(t1); if (t == 0) cyc[4] = t1 - t0; acc += a; }
    __syncthreads();
    // (f) fp64 rsq + newton (sqrt_rsqrt) dependent chain
    { double x = 2.0 + t; TICK(t0); for (int i = 0; i < n; ++i) { double y = __builtin_amdgcn_rsq(x); double g = x * y, h = 0.5 * y; double e = fma(-h, g, 0.5); g = fma(g, e, g); h = fma(h, e, h); e = fma(-h, g, 0.5); g = fma(g, e, g); x = g + 1.0; } TICK(t1); if (t == 0) cyc[5] = t1 - t0; acc += x; }
    // (g) independent FMAs x8 (issue rate)
    { double a0 = 1, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8; const double b = 1.0000001; TICK(t0); for (int i = 0; i < n; ++i) { a0 = fma(a0, b, 1e-9); a1 = fma(a1, b, 1e-9); a2 = fma(a2, b, 1e-9); a3 = fma(a3, b, 1e-9); a4 = fma(a4, b, 1e-9); a5 = fma(a5, b, 1e-9); a6 = fma(a6, b, 1e-9); a7 = fma(a7, b, 1e-9); } TICK(t1); if (t == 0) cyc[6] = t1 - t0; acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }
    // (h) global load dependent chain (L2 hit)
    if (t < 64) { int j = t; TICK(t0); for (int i = 0; i < n; ++i) { j = (int)out[8192 + (j & 1023)]; } TICK(t1); if (t == 0) cyc[7] = t1 - t0; acc += j; }
    out[t] = acc;
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 16384); hipMalloc(&cyc, 8 * 16);
    hipMemset(out, 0, 8 * 16384);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 114688);
    const int n = 1000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(512), 114688, 0, out, cyc, n); hipDeviceSynchronize(); }
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char* nm[8] = {"__syncthreads (512 thr)", "dependent ds_read chain", "dependent fma_f64 chain", "readlane x2 + fma chain", "lds write->read (wave)", "rsq+newton sqrt chain (per sqrt)", "8 independent fma (per fma)", "dependent global load (L2)"};
    const double div[8] = {1, 1, 1, 1, 1, 1, 8, 1};
    for (int i = 0; i < 8; ++i) printf("%-36s %8.1f ticks\n", nm[i], (double)h[i] / n / div[i]);
    return 0;
}
