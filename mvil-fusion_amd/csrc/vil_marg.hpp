// Marginalisation on device (estimator.cpp:1484-1683, marginalization_factor.cpp:176-316).
//
// MI355X-first restatement of MarginalizationInfo::marginalize():
//   * A = sum J^T J, b = sum J^T r over the involved factors is the SAME sweep + gather that the solve
//     uses, run on the derived sub-problem {prior, IMU(0,1), visual factors anchored in frame 0,
//     the ICP/LPS constraint touching frame 0} with every block free.  The landmarks anchored in frame 0
//     are eliminated inside the sweep (their block of A_mm is diagonal: exact pivots 1/h_ll), which
//     replaces the reference's m x m SelfAdjointEigenSolver on an arrow-shaped matrix by a 15 x 15 one.
//   * this kernel (one workgroup) then eliminates pose 0 + speed-bias 0 (or pose K-2 for
//     MARGIN_SECOND_NEW) with the reference's eigenvalue-thresholded pseudo inverse (eps = 1e-8,
//     marginalization_factor.h:70), and factors the n x n result A = V S V^T by a parallel-ordered
//     Jacobi eigen-solver into linearized_jacobians = sqrt(S) V^T, linearized_residuals = S^-1/2 V^T b
//     (marginalization_factor.cpp:301-309).
// Eigenvector basis / block order are implementation-defined in the reference;
// parity is on A, b and J0^T J0, J0^T r0.
#pragma once
#include "../../include/vilsolve.h"
#include "vil_dev.hpp"
#include "vil_math.hpp"

struct MargDev {
    int D, nd, n;               // reduced dim of the window, dropped dims, kept dims
    const int* drop_cols;       // nd   columns of S' that are marginalised here
    const int* keep_cols;       // n    columns kept, in the order of the new prior
    const double* S;            // D x D  (landmarks already eliminated)
    const double* g;            // D
    double* Add; double* Vd; double* wd;     // nd x nd work, eigenvectors, eigenvalues
    double* T;                  // n x nd
    double* A; double* b;       // n x n, n   (outputs: reduced information matrix / vector)
    double* V; double* w;       // n x n eigenvectors (columns), n eigenvalues
    double* J0; double* r0;     // n x n column-major, n
    int* pairs;                 // tournament schedule scratch: 2 x npad
    double eps;
};

namespace vd {

// Parallel-ordered cyclic Jacobi for a symmetric n x n matrix in global memory: A <- diag(w), V <- eigenvectors.
// Round-robin tournament: n/2 disjoint rotations per stage, n-1 stages per sweep.
__device__ inline void jacobi_eig(double* A, double* V, double* w, int n, int* pairs, double* sm /*>= 2*npad + 40*/) {
    const int t = threadIdx.x, NT = blockDim.x;
    const int np = (n + 1) & ~1, half = np >> 1;
    double* cs = sm;                 // half (c, s) pairs
    double* red = sm + 2 * half;
    for (int e = t; e < n * n; e += NT) V[e] = (e / n == e % n) ? 1.0 : 0.0;
    for (int i = t; i < np; i += NT) pairs[i] = i;
    __syncthreads();
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0, dg = 0;
        for (int e = t; e < n * n; e += NT) { const int i = e / n, j = e - i * n; const double v = A[e]; if (i != j) off += v * v; else dg += v * v; }
        off = wave_sum(off); dg = wave_sum(dg);
        __syncthreads();
        if ((t & 63) == 0) { red[t >> 6] = off; red[16 + (t >> 6)] = dg; }
        __syncthreads();
        off = 0; dg = 0;
        for (int q = 0; q < (NT >> 6); ++q) { off += red[q]; dg += red[16 + q]; }
        if (off <= 1e-30 * (dg + off) || off == 0.0) break;
        for (int stage = 0; stage < np - 1; ++stage) {
            // pairs: position k plays position np-1-k
            if (t < half) {
                int p = pairs[t], q = pairs[np - 1 - t];
                if (p > q) { const int tmp = p; p = q; q = tmp; }
                double c = 1.0, s = 0.0;
                if (q < n) {
                    const double apq = A[(size_t)p * n + q];
                    if (apq != 0.0) {
                        const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + tt * tt); s = tt * c;
                    }
                }
                cs[2 * t] = c; cs[2 * t + 1] = s;
            }
            __syncthreads();
            // rows: A <- R^T A
            for (int e = t; e < half * n; e += NT) {
                const int k = e / n, j = e - k * n;
                int p = pairs[k], q = pairs[np - 1 - k];
                if (p > q) { const int tmp = p; p = q; q = tmp; }
                if (q >= n) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                if (s == 0.0) continue;
                const double ap = A[(size_t)p * n + j], aq = A[(size_t)q * n + j];
                A[(size_t)p * n + j] = c * ap - s * aq;
                A[(size_t)q * n + j] = s * ap + c * aq;
            }
            __syncthreads();
            // columns: A <- A R ; V <- V R
            for (int e = t; e < half * n; e += NT) {
                const int k = e / n, i = e - k * n;
                int p = pairs[k], q = pairs[np - 1 - k];
                if (p > q) { const int tmp = p; p = q; q = tmp; }
                if (q >= n) continue;
                const double c = cs[2 * k], s = cs[2 * k + 1];
                if (s == 0.0) continue;
                const double ap = A[(size_t)i * n + p], aq = A[(size_t)i * n + q];
                A[(size_t)i * n + p] = c * ap - s * aq;
                A[(size_t)i * n + q] = s * ap + c * aq;
                const double vp = V[(size_t)i * n + p], vq = V[(size_t)i * n + q];
                V[(size_t)i * n + p] = c * vp - s * vq;
                V[(size_t)i * n + q] = s * vp + c * vq;
            }
            __syncthreads();
            // rotate the tournament: position 0 fixed, the others shift by one
            int nxt = 0;
            if (t < np && t > 0) nxt = pairs[t == 1 ? np - 1 : t - 1];
            __syncthreads();
            if (t < np && t > 0) pairs[t] = nxt;
            __syncthreads();
        }
    }
    for (int i = t; i < n; i += NT) w[i] = A[(size_t)i * n + i];
    __syncthreads();
}

}  // namespace vd

__global__ __launch_bounds__(512) void k_marg(MargDev M, int lds_a, int lds_v) {
    using namespace vd;
    __shared__ double sm[512 + 64];
    __shared__ int spairs[160];          // tournament schedule (n <= 158)
    extern __shared__ double mlds[];     // [A n x n | V n x n] when they fit (latency of the ~10 n Jacobi stages is what matters)
    const int t = threadIdx.x, NT = blockDim.x;
    const int D = M.D, nd = M.nd, n = M.n;
    // ---- dropped block, symmetrised (marginalization_factor.cpp:273), eigen pseudo inverse ------------
    double* Addw = mlds;                 // nd <= 15: the small eigen problem runs entirely in LDS (>= 4 KB are always requested)
    double* Vdw = mlds + 256;
    for (int e = t; e < nd * nd; e += NT) {
        const int i = e / nd, j = e - i * nd;
        Addw[e] = 0.5 * (M.S[(size_t)M.drop_cols[i] * D + M.drop_cols[j]] + M.S[(size_t)M.drop_cols[j] * D + M.drop_cols[i]]);
    }
    __syncthreads();
    jacobi_eig(Addw, Vdw, M.wd, nd, spairs, sm);
    for (int e = t; e < nd * nd; e += NT) M.Vd[e] = Vdw[e];
    __syncthreads();
    // T = A_kd pinv(A_dd) = (A_kd Vd) diag(1/w) Vd^T
    for (int e = t; e < n * nd; e += NT) {
        const int i = e / nd, k = e - i * nd;
        double s = 0;
        for (int q = 0; q < nd; ++q) s += M.S[(size_t)M.keep_cols[i] * D + M.drop_cols[q]] * M.Vd[(size_t)q * nd + k];
        M.A[e] = M.wd[k] > M.eps ? s / M.wd[k] : 0.0;       // scratch in A: (A_kd Vd) diag(1/w)
    }
    __syncthreads();
    for (int e = t; e < n * nd; e += NT) {
        const int i = e / nd, j = e - i * nd;
        double s = 0;
        for (int k = 0; k < nd; ++k) s += M.A[(size_t)i * nd + k] * M.Vd[(size_t)j * nd + k];
        M.T[e] = s;
    }
    __syncthreads();
    // A = A_kk - T A_dk ; b = b_k - T b_d      (marginalization_factor.cpp:289-290)
    for (int e = t; e < n * n + n; e += NT) {
        if (e < n * n) {
            const int i = e / n, j = e - i * n;
            double s = M.S[(size_t)M.keep_cols[i] * D + M.keep_cols[j]];
            for (int k = 0; k < nd; ++k) s -= M.T[(size_t)i * nd + k] * M.S[(size_t)M.drop_cols[k] * D + M.keep_cols[j]];
            M.V[e] = s;                                    // staged in V, symmetrised copy goes to A below
        } else {
            const int i = e - n * n;
            double s = M.g[M.keep_cols[i]];
            for (int k = 0; k < nd; ++k) s -= M.T[(size_t)i * nd + k] * M.g[M.drop_cols[k]];
            M.b[i] = s;
        }
    }
    __syncthreads();
    // Eigen's SelfAdjointEigenSolver reads the lower triangle only
    for (int e = t; e < n * n; e += NT) { const int i = e / n, j = e - i * n; M.J0[e] = i >= j ? M.V[e] : M.V[(size_t)j * n + i]; }
    __syncthreads();
    for (int e = t; e < n * n; e += NT) { M.A[e] = M.V[e]; }
    __syncthreads();
    double* Aw = lds_a ? mlds : M.T;                            // working copy of the symmetric matrix
    double* Vw = lds_v ? mlds + (size_t)n * n : M.V;
    for (int e = t; e < n * n; e += NT) Aw[e] = M.J0[e];
    __syncthreads();
    jacobi_eig(Aw, Vw, M.w, n, spairs, sm);
    if (lds_v) { for (int e = t; e < n * n; e += NT) M.V[e] = Vw[e]; __syncthreads(); }
    // linearized_jacobians = sqrt(S) V^T (column-major n x n), linearized_residuals = S^-1/2 V^T b
    for (int e = t; e < n * n + n; e += NT) {
        if (e < n * n) {
            const int j = e / n, k = e - j * n;               // column-major element (k, j): J0[j*n + k]
            const double wk = M.w[k];
            M.J0[e] = wk > M.eps ? sqrt(wk) * M.V[(size_t)j * n + k] : 0.0;
        } else {
            const int k = e - n * n;
            const double wk = M.w[k];
            double s = 0;
            for (int j = 0; j < n; ++j) s += M.V[(size_t)j * n + k] * M.b[j];
            M.r0[k] = wk > M.eps ? s / sqrt(wk) : 0.0;
        }
    }
}
