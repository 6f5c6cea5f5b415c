// Micro-benchmark: cost of v_mfma_f64_16x16x4_f64 per wave as a function of the number of independent accumulators and of the
// waves per SIMD (MI355X).  hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_chain.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(long long* out, double* sink, int n) {
    d4 acc[NACC];
    for (int u = 0; u < NACC; ++u) acc[u] = d4{0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    long long t0, t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = 0; for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int NACC> void run(int threads, long long* d_out, double* d_sink) {
    const int n = 2000;
    hipLaunchKernelGGL(k<NACC>, dim3(1), dim3(threads), 0, 0, d_out, d_sink, n);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
    printf("threads %4d (waves/SIMD %.2f)  accumulators %d : %.1f cycles per MFMA per wave\n", threads, threads / 256.0, NACC, (double)c / (n * NACC));
}
int main() {
    long long* d_out; double* d_sink; hipMalloc(&d_out, 64); hipMalloc(&d_sink, 8 * 1024);
    for (int threads : {64, 256, 512, 1024}) { run<1>(threads, d_out, d_sink); run<2>(threads, d_out, d_sink); run<4>(threads, d_out, d_sink); }
    return 0;
}
