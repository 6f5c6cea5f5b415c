#include <hip/hip_runtime.h>
#include <cstdio>
#define TICK(x) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(x)::"memory")
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int n, int stride) {
    extern __shared__ double lds[];
    const int t = threadIdx.x;
    for (int i = t; i < 8192; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();
    long long t0, t1;
    // (a) distinct addresses per lane, n atomics back to back
    TICK(t0); for (int i = 0; i < n; ++i) unsafeAtomicAdd(&lds[(t * stride + i * 517) & 8191], 1.0); __syncthreads(); TICK(t1);
    if (t == 0) cyc[0] = t1 - t0;
    // (b) plain read-modify-write, distinct addresses
    TICK(t0); for (int i = 0; i < n; ++i) lds[(t + i * 512) & 8191] += 1.0; __syncthreads(); TICK(t1);
    if (t == 0) cyc[1] = t1 - t0;
    // (c) all lanes of a wave hit 8 addresses (conflicts)
    TICK(t0); for (int i = 0; i < n; ++i) unsafeAtomicAdd(&lds[((t & 7) + i * 8) & 8191], 1.0); __syncthreads(); TICK(t1);
    if (t == 0) cyc[2] = t1 - t0;
    out[t] = lds[t];
}
int main() {
    double* out; long long* cyc; hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 64);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int n = 64;
    for (int stride = 1; stride <= 33; stride += 32) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(512), 65536, 0, out, cyc, n, stride); hipDeviceSynchronize(); }
        long long h[3]; hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
        printf("stride %2d: ds_add_f64 distinct %.1f ticks/atomic/thread | plain RMW %.1f | 8-address conflict %.1f   (512 threads, %d each)\n", stride, (double)h[0] / n, (double)h[1] / n, (double)h[2] / n, n);
    }
    return 0;
}
