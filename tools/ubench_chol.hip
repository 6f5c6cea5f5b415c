// stand-alone timing + check of vd::chol_lookahead (csrc/vil_step.hpp) on random SPD matrices: one workgroup, s_memtime around the call.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -mllvm -disable-machine-licm -o /tmp/ubench_chol tools/ubench_chol.hip   
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <cstring>
#include "../include/vilsolve.h"
#include "../mvil-fusion_amd/csrc/vil_internal.h"
#include "../mvil-fusion_amd/csrc/vil_tuning.hpp"
#include "../mvil-fusion_amd/csrc/vil_coop.hpp"
#include "../mvil-fusion_amd/csrc/vil_dev.hpp"
#include "../mvil-fusion_amd/csrc/vil_finish.hpp"
#include "../mvil-fusion_amd/csrc/vil_sweep.hpp"
#include "../mvil-fusion_amd/csrc/vil_eval.hpp"
#include "../mvil-fusion_amd/csrc/vil_step.hpp"
using namespace vd;
template <int SLOTS, bool RW = false>
__global__ __launch_bounds__(VIL_STEP_THREADS) void k_chol(const double* Ain, double* Lout, int D, long long* tm, int reps, double* xout) {
    extern __shared__ double A[];
    __shared__ StepShared s;
    const int t = threadIdx.x, R = D + 1, T = (R + 15) >> 4;
    if (t == 0) { int g = 0; for (int I = 0; I < T; ++I) for (int J = 0; J <= I; ++J) { s.tI[g] = I; s.tJ[g] = J; ++g; } }
    long long best = 1ll << 60, bestb = 1ll << 60; bool ok = true;
    for (int rep = 0; rep < reps; ++rep) {
        __syncthreads();
        for (int e = t; e < R * R; e += blockDim.x) { const int i = e / R, j = e % R; if (j <= i) A[tl_idx(i, j)] = Ain[(size_t)i * R + j]; }
        __syncthreads();
        long long t0, t1;
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        if constexpr (RW) ok = chol_dense<SLOTS>(A, D, s); else ok = chol_lookahead<SLOTS, false>(A, D, s);
        __syncthreads();
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        if (t1 - t0 < best) best = t1 - t0;
        if (rep == reps - 1) { for (int e = t; e < R * R; e += blockDim.x) { const int i = e / R, j = e % R; Lout[e] = j < i ? A[tl_idx(i, j)] : (j == i && i < D ? 1.0 / s.dinv[i] : 0.0); } __syncthreads(); }
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        if constexpr (RW) back_subst_cols(A, D, s); else back_subst(A, D, s);
        __syncthreads();
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        if (t1 - t0 < bestb) bestb = t1 - t0;
    }
    if (t == 0) { tm[0] = best; tm[1] = ok; tm[2] = bestb; }
    for (int e = t; e < D; e += blockDim.x) xout[e] = s.y[e];
}
int main(int argc, char** argv) {
    const int Ds[] = {67, 67, 127, 127, 79, 79, 40, 40, 64, 64, 19, 31, 48, 55, 97, 97, 131, 131, 68, 68};
    int prevD = -1;
    for (int D : Ds) {
        const bool rw = D == prevD || D == 19 || D == 31 || D == 48 || D == 55; prevD = D;      // a size listed twice: the second run takes chol_dense (row-per-lane panels for D <= 67)
        const int R = D + 1, T = (R + 15) / 16;
        std::vector<double> M((size_t)R * R), A((size_t)R * R, 0.0), L((size_t)R * R);
        srand(7); for (auto& v : M) v = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < D; ++i) for (int j = 0; j <= i; ++j) { double a = 0; for (int k = 0; k < R; ++k) a += M[(size_t)i * R + k] * M[(size_t)j * R + k]; A[(size_t)i * R + j] = a + (i == j ? D : 0); }
        for (int j = 0; j < D; ++j) A[(size_t)D * R + j] = M[(size_t)D * R + j];
        double *dA, *dL, *dx; long long* dt; hipMalloc(&dx, 8 * 512); hipMalloc(&dA, 8 * A.size()); hipMalloc(&dL, 8 * A.size()); hipMalloc(&dt, 512);
        hipMemcpy(dA, A.data(), 8 * A.size(), hipMemcpyHostToDevice);
        const size_t lds = 8 * (size_t)TILE_SZ * (T * (T + 1) / 2);
        if (rw && T * (T + 1) / 2 > 21) { hipFuncSetAttribute((const void*)k_chol<CH_SLOTS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL((k_chol<CH_SLOTS, true>), dim3(1), dim3(VIL_STEP_THREADS), lds, 0, dA, dL, D, dt, 20, dx); }
        else if (rw) { hipFuncSetAttribute((const void*)k_chol<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL((k_chol<3, true>), dim3(1), dim3(VIL_STEP_THREADS), lds, 0, dA, dL, D, dt, 20, dx); }
        else if (T * (T + 1) / 2 <= 21) { hipFuncSetAttribute((const void*)k_chol<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k_chol<3>, dim3(1), dim3(VIL_STEP_THREADS), lds, 0, dA, dL, D, dt, 20, dx); }
        else { hipFuncSetAttribute((const void*)k_chol<CH_SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k_chol<CH_SLOTS>, dim3(1), dim3(VIL_STEP_THREADS), lds, 0, dA, dL, D, dt, 20, dx); }
        long long h[40]; hipMemcpy(h, dt, 34 * 8, hipMemcpyDeviceToHost); hipMemcpy(L.data(), dL, 8 * L.size(), hipMemcpyDeviceToHost);
        double err = 0, nrm = 0;      // || L L^T - A || over the lower triangle, and the forward substitution row
        for (int i = 0; i < D; ++i) for (int j = 0; j <= i; ++j) { double a = 0; for (int k = 0; k <= j; ++k) a += L[(size_t)i * R + k] * L[(size_t)j * R + k]; err = fmax(err, fabs(a - A[(size_t)i * R + j])); nrm = fmax(nrm, fabs(A[(size_t)i * R + j])); }
        for (int j = 0; j < D; ++j) { double a = 0; for (int k = 0; k <= j; ++k) a += L[(size_t)D * R + k] * L[(size_t)j * R + k]; err = fmax(err, fabs(a - A[(size_t)D * R + j])); }
        // back substitution: x solves L^T x = y (y = row D of L)
        std::vector<double> x(D), xr(D);
        hipMemcpy(x.data(), dx, 8 * D, hipMemcpyDeviceToHost);
        for (int i = D - 1; i >= 0; --i) { double a = L[(size_t)D * R + i]; for (int k = i + 1; k < D; ++k) a -= L[(size_t)k * R + i] * xr[k]; xr[i] = a / L[(size_t)i * R + i]; }
        double ex = 0, nx = 0; for (int i = 0; i < D; ++i) { ex = fmax(ex, fabs(x[i] - xr[i])); nx = fmax(nx, fabs(xr[i])); }
        const int nblk = (D + 3) / 4;
        printf("%s D %3d: ok %lld, %6lld ticks = %5.2f us, %5.0f ticks per block step (%d), max |L L^T - A| / max |A| = %.2e | back substitution %6lld ticks = %5.2f us, max |x - x_ref| / max |x| = %.2e\n", rw ? "rowwave  " : "lookahead", D, h[1], h[0], h[0] / 2390.0, (double)h[0] / nblk, nblk, err / nrm, h[2], h[2] / 2390.0, ex / nx);
        hipFree(dA); hipFree(dL); hipFree(dt);
    }
    return 0;
}
