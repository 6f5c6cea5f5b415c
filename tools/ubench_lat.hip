// latency microbenchmarks for the dense-Cholesky critical path (gfx950): dependent fp64 fma, rsq / rcp, LDS round trip, barrier, mfma f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long long now() { long long t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__device__ __forceinline__ long long nowd(double& v) { long long t; asm volatile("s_nop 0\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(v) :: "memory"); return t; }
__global__ __launch_bounds__(512) void k(double* out, long long* tm, double seed) {
    __shared__ double sh[4096];
    const int t = threadIdx.x, w = t >> 6;
    double x = seed + t * 1e-9, y = 1.0000001;
    sh[t] = x; __syncthreads();
    long long t0, t1; int q = 0;
    // 1. dependent fma chain, 64 ops (only wave 0 active on its SIMD?  all waves run it: 2 waves per SIMD)
    t0 = nowd(x);
#pragma unroll
    for (int i = 0; i < 64; ++i) x = __builtin_fma(x, y, 1e-9);
    t1 = nowd(x); if (t == 0) tm[q] = t1 - t0; ++q;
    // 2. the same with only wave 0 working
    __syncthreads();
    if (w == 0) { t0 = nowd(x);
#pragma unroll
        for (int i = 0; i < 64; ++i) x = __builtin_fma(x, y, 1e-9);
        t1 = nowd(x); if (t == 0) tm[q] = t1 - t0; } ++q;
    __syncthreads();
    // 3. independent fmas (issue rate): 8 chains x 16
    if (w == 0) { double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7; t0 = nowd(a0);
#pragma unroll
        for (int i = 0; i < 16; ++i) { a0 = __builtin_fma(a0, y, 1e-9); a1 = __builtin_fma(a1, y, 1e-9); a2 = __builtin_fma(a2, y, 1e-9); a3 = __builtin_fma(a3, y, 1e-9); a4 = __builtin_fma(a4, y, 1e-9); a5 = __builtin_fma(a5, y, 1e-9); a6 = __builtin_fma(a6, y, 1e-9); a7 = __builtin_fma(a7, y, 1e-9); }
        x = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7)); t1 = nowd(x); if (t == 0) tm[q] = t1 - t0; } ++q;
    __syncthreads();
    // 4. dependent rsq chain x 16
    if (w == 0) { double r = x * x + 2.0; t0 = nowd(r);
#pragma unroll
        for (int i = 0; i < 16; ++i) r = __builtin_amdgcn_rsq(r) + 1.5;
        t1 = nowd(r); if (t == 0) tm[q] = t1 - t0; x += r; } ++q;
    __syncthreads();
    // 5. dependent rcp chain x 16
    if (w == 0) { double r = x * x + 2.0; t0 = nowd(r);
#pragma unroll
        for (int i = 0; i < 16; ++i) r = __builtin_amdgcn_rcp(r) + 1.5;
        t1 = nowd(r); if (t == 0) tm[q] = t1 - t0; x += r; } ++q;
    __syncthreads();
    // 6. LDS write -> read (same wave, dependent) x 16
    if (w == 0) { double r = x; t0 = nowd(r);
#pragma unroll
        for (int i = 0; i < 16; ++i) { sh[t] = r; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); r = sh[(t + 1) & 63] + 1.0; }
        t1 = nowd(r); if (t == 0) tm[q] = t1 - t0; x += r; } ++q;
    __syncthreads();
    // 7. barrier x 16, 8 waves
    t0 = now();
#pragma unroll
    for (int i = 0; i < 16; ++i) { asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory"); }
    t1 = now(); if (t == 0) tm[q] = t1 - t0; ++q;
    // 8. write, barrier, read from other wave x 16
    { double r = x; t0 = now();
#pragma unroll
      for (int i = 0; i < 16; ++i) { sh[t] = r; asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory"); r = sh[(t + 64) & 511] + 1.0; asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory"); }
      t1 = now(); if (t == 0) tm[q] = t1 - t0; x += r; } ++q;
    __syncthreads();
    // 9. dependent mfma f64 16x16x4 x 16 (wave 0)
    if (w == 0) { d4 c = {x, x, x, x}; { double z_ = c[0]; t0 = nowd(z_); c[0] = z_; }
#pragma unroll
        for (int i = 0; i < 16; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c, 0, 0, 0);
        { double z_ = c[0]; t1 = nowd(z_); c[0] = z_; } if (t == 0) tm[q] = t1 - t0; x += c[0] + c[1] + c[2] + c[3]; } ++q;
    __syncthreads();
    // 10. independent mfma x 16 (4 accumulators)
    if (w == 0) { d4 c0 = {x, x, x, x}, c1 = c0, c2 = c0, c3 = c0; { double z_ = c0[0]; t0 = nowd(z_); c0[0] = z_; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c1, 0, 0, 0); c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c3, 0, 0, 0); }
        x += c0[0] + c1[1] + c2[2] + c3[3]; t1 = nowd(x); if (t == 0) tm[q] = t1 - t0; } ++q;
    __syncthreads();
    // 11. v_readlane broadcast of a double x 16 (dependent through fma)
    if (w == 0) { double r = x; t0 = nowd(r);
#pragma unroll
        for (int i = 0; i < 16; ++i) { const int lo = __builtin_amdgcn_readlane((int)__double2loint(r), i), hi = __builtin_amdgcn_readlane(__double2hiint(r), i); r = __builtin_fma(__hiloint2double(hi, lo), y, r); }
        t1 = nowd(r); if (t == 0) tm[q] = t1 - t0; x += r; } ++q;
    __syncthreads();
    // 12. mfma -> LDS write -> barrier -> LDS read round trip x 8 (what a matrix-core update of the next diagonal block costs the chain)
    { d4 c = {x, x, x, x}; double r = y; t0 = now();
#pragma unroll
      for (int i = 0; i < 8; ++i) { c = __builtin_amdgcn_mfma_f64_16x16x4f64(r, y, c, 0, 0, 0); sh[t] = c[0]; sh[512 + t] = c[1]; sh[1024 + t] = c[2]; sh[1536 + t] = c[3]; asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory"); r = sh[(t * 7 + 64) & 2047]; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
      t1 = now(); if (t == 0) tm[q] = t1 - t0; x += r + c[0]; } ++q;
    out[t] = x;
}
int main() {
    double* o; long long* tm; hipMalloc(&o, 4096); hipMalloc(&tm, 256); hipMemset(tm, 0, 256);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, o, tm, 0.5);
    long long h[32]; hipMemcpy(h, tm, 256, hipMemcpyDeviceToHost);
    const char* nm[] = {"dep fma x64, 8 waves", "dep fma x64, 1 wave", "indep fma 128, 1 wave", "dep rsq(+add) x16", "dep rcp(+add) x16", "lds write->read x16", "barrier x16 (8 waves)", "write,barrier,read,barrier x16", "dep mfma x16", "indep mfma x16", "readlane dbl + fma x16", "mfma->lds->barrier->read x8"};
    const int cnt[] = {64, 64, 128, 16, 16, 16, 16, 16, 16, 16, 16, 8};
    for (int q = 0; q < 12; ++q) printf("%-34s total %6lld ticks -> %7.1f ticks each\n", nm[q], h[q], (double)h[q] / cnt[q]);
    return 0;
}
