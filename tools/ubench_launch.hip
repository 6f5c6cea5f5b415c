// what a launch of a grid like k_iter's costs when its workgroups do nothing: back-to-back launches of an empty kernel, by grid size, block size and dynamic LDS
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_launch tools/ubench_launch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ __launch_bounds__(512) void k_empty(int* p) { extern __shared__ double s[]; if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) p[0] = (int)s[0]; }
int main() {
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int grids[] = {8, 64, 256, 404, 832}; const size_t ldss[] = {0, 100 * 1024};
    for (size_t lds : ldss) for (int g : grids) {
        for (int mode = 0; mode < 2; ++mode) {        // 0: direct launches, 1: one hipGraph of 10 launches
            const int N = 10, REP = 200;
            hipGraphExec_t ex = nullptr;
            if (mode) {
                hipGraph_t gr; hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
                for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(g), dim3(512), lds, st, (int*)nullptr);
                hipStreamEndCapture(st, &gr); hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0); hipGraphDestroy(gr);
            }
            auto run = [&]() { if (mode) hipGraphLaunch(ex, st); else for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(g), dim3(512), lds, st, (int*)nullptr); };
            for (int w = 0; w < 20; ++w) run();
            hipStreamSynchronize(st);
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < REP; ++r) run();
            hipStreamSynchronize(st);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REP * N);
            printf("lds %6zu kB grid %4d x 512  %s: %.2f us per launch\n", lds / 1024, g, mode ? "graph of 10" : "direct     ", us);
            if (ex) hipGraphExecDestroy(ex);
        }
    }
    return 0;
}
