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
//     marginalization_factor.h:70), and takes a square root of the n x n result A -- pivoted Cholesky in LDS
//     (gram_sqrt below; see the note in k_marg on the choice of square root) -- into linearized_jacobians = sqrt(S) V^T,
//     linearized_residuals = S^-1/2 V^T b (marginalization_factor.cpp:301-309).
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
    double* V; double* w;       // n x n scratch, n eigenvalues
    double* J0; double* r0;     // n x n column-major, n
    double eps;
    int* stat;                  // [0] stages of the small eigen problem, [1] stages of the n x n one (diagnostic)
};

namespace vd {

#define MARG_THREADS 1024
#define MARG_MAX_SWEEPS 30
#define MARG_NMAX 136

// round-robin tournament in closed form: position 0 holds player 0, positions 1..np-1 rotate by one per stage;
// pair k of a stage = positions (k, np-1-k).  `st` = stage mod (np-1).
__device__ __forceinline__ void rr_pair(int k, int st, int np, int& p, int& q) {
    const int ring = np - 1;
    int a = k - 1 + st; if (a >= ring) a -= ring;
    a = k == 0 ? 0 : 1 + a;
    int b = np - 2 - k + st; if (b >= ring) b -= ring;
    b += 1;
    p = a < b ? a : b; q = a < b ? b : a;
}

// fp64 1/sqrt(x) and 1/x from the hardware seeds + Newton steps
// Square root of a symmetric positive semi-definite matrix, in LDS, optionally orthogonalised into its eigen basis.
//   in : M  n x n row-major in LDS (full, symmetric); bb (optional) right-hand side of length n in LDS
//   out: row k of M = g_k with  sum_k g_k g_k^T = M_in ; lam[k] = |g_k|^2 ; bb -> y with sum_k g_k y_k = bb_in
// Phase 1: diagonally pivoted Cholesky M = G G^T in place (the factor column produced at a step is stored in the ROW of
//   its pivot index; pivots that lost all their digits -- d <= max(tiny, 16 n eps d_original, floor_rel max_diag) --
//   give zero rows: the directions SelfAdjointEigenSolver would report as noise-level eigenvalues).  The forward substitution G y = bb rides
//   along as one more column.
// Phase 2 (orthogonalise = true): one-sided (Hestenes) Jacobi on the rows of G, round-robin ordered, one wave per pair
//   (contiguous rows: conflict-free LDS access, the three dot products are wave butterflies), one barrier per stage.
//   Afterwards the rows are mutually orthogonal: eigenvalue lam[k], eigenvector g_k / sqrt(lam[k]).
//   Only the 15 x 15 pseudo inverse needs this; bb must be null then.
__device__ inline int gram_sqrt(double* M, double* lam, int n, double tiny, double floor_rel, bool orthogonalise, double* bb, double* yout, double* sm /* >= 3 * MARG_NMAX + 64 doubles */) {
    const int t = threadIdx.x, NT = blockDim.x;
    const int wave = t >> 6, lane = t & 63, NW = NT >> 6;
    double* thr = sm;                                     // per index: smallest acceptable pivot
    int* done = reinterpret_cast<int*>(sm + MARG_NMAX);   // 0 open, 1 used as pivot, 2 exhausted
    double* wv = sm + MARG_NMAX + MARG_NMAX / 2;          // per-wave argmax value
    int* wi = reinterpret_cast<int*>(wv + 16);            // per-wave argmax index ; wi[32] = sweep flag
    double* ysh = wv + 40;                                // y of the current step
    // largest diagonal entry (every wave reduces the whole diagonal: n <= 136 = three values per lane)
    double dmax = 0.0;
    for (int i = lane; i < n; i += 64) dmax = fmax(dmax, M[i * n + i]);
    dmax = wave_fmax_all(dmax);
    if (t < n) {
        // acceptable pivot: above the rounding level of its own entry AND of the matrix it came from.  The input is a
        // Schur complement whose entries carry ~1e-11 relative noise from the cancellation upstream: a gauge direction
        // shows up as a pivot of pure noise, and dividing a column by its root would amplify that noise.
        const double d0 = M[t * n + t];
        const double rel = fmax(16.0 * n * 2.220446049250313e-16 * d0, floor_rel * dmax);
        thr[t] = rel > tiny ? rel : tiny; done[t] = 0; if (yout) yout[t] = 0.0;
    }
    __syncthreads();
    // ---- phase 1 -----------------------------------------------------------------------------------------------------
    for (int step = 0; step < n; ++step) {
        double v = -1.0; int vi = -1;
        if (t < n && done[t] == 0) { const double d = M[t * n + t]; if (d > thr[t]) { v = d; vi = t; } }
        if (wave * 64 < n) {
            const double vm = wave_fmax_all(v);                     // largest pivot of the wave, then the largest index that holds it
            const int im = wave_imax_all(v == vm ? vi : -1);
            if (lane == 0) { wv[wave] = vm; wi[wave] = im; }
        }
        __syncthreads();
        double pv = -1.0; int pi_ = -1;
        for (int q = 0; q * 64 < n; ++q) { const double v2 = wv[q]; const int i2 = wi[q]; if (v2 > pv || (v2 == pv && i2 > pi_)) { pv = v2; pi_ = i2; } }
        if (pi_ < 0) break;                               // uniform: nothing acceptable is left
        const double isd = rsqrt_nr(pv);
        double* rowp = M + pi_ * n;
        if (t < n) rowp[t] = (done[t] == 0) ? rowp[t] * isd : 0.0;      // includes t == pi_: pv / sqrt(pv)
        if (t == 0 && bb) { const double y = bb[pi_] * isd; ysh[0] = y; yout[pi_] = y; }
        __syncthreads();
        if (t == 0) done[pi_] = 1;
        if (bb && t < n && t != pi_) bb[t] -= rowp[t] * ysh[0];        // rowp is zero on finished indices
        for (int i = wave; i < n; i += NW) {
            if (i == pi_) continue;
            const double li = rowp[i];
            if (li == 0.0) continue;
            double* row = M + i * n;
            for (int k = lane; k < n; k += 64) row[k] -= li * rowp[k];
        }
        __syncthreads();
    }
    __syncthreads();
    for (int i = wave; i < n; i += NW) if (done[i] != 1) for (int k = lane; k < n; k += 64) M[i * n + k] = 0.0;
    __syncthreads();
    // ---- phase 2 -----------------------------------------------------------------------------------------------------
    const int np = (n + 1) & ~1, half = np >> 1, ring = np - 1;
    int stages = 0;
    for (int sweep = 0; orthogonalise && sweep < MARG_MAX_SWEEPS; ++sweep) {
        if (t == 0) wi[32] = 0;
        __syncthreads();
        bool rotated = false;
        for (int stage = 0; stage < ring; ++stage, ++stages) {
            for (int k = wave; k < half; k += NW) {
                int p, q;
                rr_pair(k, stage, np, p, q);
                if (q >= n) continue;
                double* gp = M + p * n; double* gq = M + q * n;
                const int i0 = lane, i1 = lane + 64, i2 = lane + 128;
                const double p0 = i0 < n ? gp[i0] : 0.0, p1 = i1 < n ? gp[i1] : 0.0, p2 = i2 < n ? gp[i2] : 0.0;
                const double q0 = i0 < n ? gq[i0] : 0.0, q1 = i1 < n ? gq[i1] : 0.0, q2 = i2 < n ? gq[i2] : 0.0;
                double al = p0 * p0 + p1 * p1 + p2 * p2, be = q0 * q0 + q1 * q1 + q2 * q2, ga = p0 * q0 + p1 * q1 + p2 * q2;
                al = wave_total(al); be = wave_total(be); ga = wave_total(ga);      // DPP folds, wave-uniform results
                if (ga == 0.0 || ga * ga <= 1e-28 * al * be) continue;
                rotated = true;
                // t = sign(d) ga / (|d| + sqrt(d^2 + ga^2)), d = (be - al)/2 ; c = 1/sqrt(1+t^2) ; s = t c
                const double d = 0.5 * (be - al);
                const double h2 = d * d + ga * ga;
                const double h = fabs(d) + h2 * rsqrt_nr(h2);
                const double tt = (d >= 0 ? ga : -ga) * rcp_nr(h);
                const double c = rsqrt_nr(1.0 + tt * tt), sn = tt * c;
                if (i0 < n) { gp[i0] = c * p0 - sn * q0; gq[i0] = sn * p0 + c * q0; }
                if (i1 < n) { gp[i1] = c * p1 - sn * q1; gq[i1] = sn * p1 + c * q1; }
                if (i2 < n) { gp[i2] = c * p2 - sn * q2; gq[i2] = sn * p2 + c * q2; }
            }
            __syncthreads();
        }
        if (rotated && lane == 0) wi[32] = 1;
        __syncthreads();
        const int any = wi[32];
        __syncthreads();
        if (!any) break;
    }
    for (int i = wave; i < n; i += NW) {
        double s = 0;
        for (int k = lane; k < n; k += 64) { const double g = M[i * n + k]; s += g * g; }
        s = wave_total(s);
        if (lane == 0) lam[i] = s;
    }
    __syncthreads();
    return stages;
}

}  // namespace vd

__global__ __launch_bounds__(MARG_THREADS) void k_marg(MargDev M) {
    using namespace vd;
    __shared__ double sm[3 * MARG_NMAX + 64];
    __shared__ double bsh[MARG_NMAX];
    extern __shared__ double mlds[];     // n x n: the symmetric matrix, then its orthogonalised factor rows
    const int t = threadIdx.x, NT = blockDim.x;
    const int D = M.D, nd = M.nd, n = M.n;
    // ---- dropped block, symmetrised (marginalization_factor.cpp:273), eigen pseudo inverse ------------
    for (int e = t; e < nd * nd; e += NT) {
        const int i = e / nd, j = e - i * nd;
        mlds[e] = 0.5 * (M.S[(size_t)M.drop_cols[i] * D + M.drop_cols[j]] + M.S[(size_t)M.drop_cols[j] * D + M.drop_cols[i]]);
    }
    __syncthreads();
    const int nsd = gram_sqrt(mlds, M.wd, nd, 1e-30, 0.0, true, nullptr, nullptr, sm);
    // eigenvectors (columns of Vd) = normalised factor rows
    for (int e = t; e < nd * nd; e += NT) { const int i = e / nd, k = e - i * nd; const double lk = M.wd[k]; M.Vd[e] = lk > 0.0 ? mlds[k * nd + i] * rsqrt_nr(lk) : 0.0; }
    if (t == 0) M.stat[0] = nsd;
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
    for (int e = t; e < n * n; e += NT) mlds[e] = M.J0[e];
    if (t < n) bsh[t] = M.b[t];
    __syncthreads();
    // linearized_jacobians / linearized_residuals (marginalization_factor.cpp:301-309) are ANY pair with J0^T J0 = A and
    // J0^T r0 = b: the reference takes sqrt(S) V^T from an eigen-decomposition, whose basis is implementation-defined
    // (SURVEY App. C #12).  Here J0 = G^T from the pivoted Cholesky A = G G^T and r0 = G^-1 b: the prior residual
    // r0 + J0 dx is the reference's up to a left orthogonal factor, i.e. the same cost, gradient and Gauss-Newton matrix;
    // pivots at rounding level are dropped where the reference drops eigenvalues below eps.
    const int ns = gram_sqrt(mlds, M.w, n, 1e-30, 1e-10, false, bsh, M.r0, sm);
    if (t == 0) M.stat[1] = ns;
    for (int e = t; e < n * n; e += NT) {
        const int j = e / n, k = e - j * n;                   // column-major element (k, j): J0[j*n + k]
        M.J0[e] = mlds[k * n + j];
    }
}
