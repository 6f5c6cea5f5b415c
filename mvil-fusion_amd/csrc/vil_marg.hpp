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
//     Jacobi eigen-solver (matrix in LDS, two barriers per stage of n/2 rotations; the eigenvectors are rebuilt from
//     the rotation log by a barrier-free, wave-synchronous replay) into linearized_jacobians = sqrt(S) V^T, linearized_residuals = S^-1/2 V^T b
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
    double2* rlog;              // rotation log: (c, s) per pair per stage, MARG_MAX_SWEEPS sweeps
    double* J0; double* r0;     // n x n column-major, n
    double eps;
    int* stat;                  // [0] stages of the small eigen problem, [1] stages of the n x n one (diagnostic)
};

namespace vd {

#define MARG_THREADS 1024
#define MARG_MAX_SWEEPS 30

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

// fp64 1/sqrt(x) and 1/x from the hardware seeds + Newton steps (full precision for the rotation to stay orthogonal)
__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}
__device__ __forceinline__ double rcp_nr(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}

// Parallel-ordered (round-robin) two-sided Jacobi on a symmetric matrix held in LDS -- eigenvalues + a LOG of the
// rotations; the eigenvectors are rebuilt afterwards by jacobi_vectors(), which needs no workgroup barrier at all.
//   A: np x np working copy (np = n rounded up to even; the padding row / column is zero and never rotated).
//   One stage = np/2 disjoint rotations (p_k, q_k).  The 2 x 2 block of A in rows (p_k, q_k), columns (p_m, q_m)
//   becomes R_k^T B R_m and is owned by ONE thread, so a stage is: angles (np/2 lanes) | barrier | one pass over the
//   (np/2)^2 blocks | barrier.  Returns the number of stages logged (uniform over the workgroup).
__device__ inline int jacobi_eig(double* A, double* w, int n, double2* rlog, double* sm /*>= 3 * 68 + 40 doubles*/) {
    const int t = threadIdx.x, NT = blockDim.x;
    const int np = (n + 1) & ~1, half = np >> 1, ring = np - 1;
    double* cs = sm;                                  // (c, s) per pair
    int* pq = reinterpret_cast<int*>(sm + 2 * half);   // (p, q) per pair
    double* red = sm + 3 * half;
    const float inv_half = 1.0f / (float)half;
    int gs = 0;
    for (int sweep = 0; sweep < MARG_MAX_SWEEPS; ++sweep) {
        double off = 0, dg = 0;
        for (int e = t; e < np * np; e += NT) { const double v = A[e]; if ((e % (np + 1)) == 0) dg += v * v; else off += v * v; }
        off = wave_sum(off); dg = wave_sum(dg);
        __syncthreads();
        if ((t & 63) == 0) { red[t >> 6] = off; red[16 + (t >> 6)] = dg; }
        __syncthreads();
        off = 0; dg = 0;
        for (int q = 0; q < (NT >> 6); ++q) { off += red[q]; dg += red[16 + q]; }
        if (off <= 1e-30 * dg || off <= 0.0) break;
        for (int stage = 0; stage < ring; ++stage, ++gs) {
            if (t < half) {
                int p_, q_;
                rr_pair(t, stage, np, p_, q_);
                double c = 1.0, sn = 0.0;
                if (q_ < n) {
                    const double apq = A[p_ * np + q_];
                    if (apq != 0.0) {
                        // t = sign(d) apq / (|d| + sqrt(d^2 + apq^2)), d = (aqq - app)/2 ; c = 1/sqrt(1+t^2) ; s = t c
                        const double d = 0.5 * (A[q_ * np + q_] - A[p_ * np + p_]);
                        const double h2 = d * d + apq * apq;
                        const double h = fabs(d) + h2 * rsqrt_nr(h2);
                        const double tt = (d >= 0 ? apq : -apq) * rcp_nr(h);
                        c = rsqrt_nr(1.0 + tt * tt); sn = tt * c;
                    }
                }
                cs[2 * t] = c; cs[2 * t + 1] = sn; pq[2 * t] = p_; pq[2 * t + 1] = q_;
                rlog[(size_t)gs * half + t] = make_double2(c, sn);
            }
            __syncthreads();
            for (int e = t; e < half * half; e += NT) {
                int k = (int)((float)e * inv_half);
                if (k * half > e) --k; else if ((k + 1) * half <= e) ++k;
                const int m = e - k * half;
                const double ck = cs[2 * k], sk = cs[2 * k + 1], cm = cs[2 * m], sm_ = cs[2 * m + 1];
                if (sk == 0.0 && sm_ == 0.0) continue;
                const int pk = pq[2 * k], qk = pq[2 * k + 1], pm = pq[2 * m], qm = pq[2 * m + 1];
                double* r0 = A + pk * np; double* r1 = A + qk * np;
                const double bpp = r0[pm], bpq = r0[qm], bqp = r1[pm], bqq = r1[qm];
                const double tpp = ck * bpp - sk * bqp, tpq = ck * bpq - sk * bqq;
                const double tqp = sk * bpp + ck * bqp, tqq = sk * bpq + ck * bqq;
                const bool dgb = k == m;
                r0[pm] = cm * tpp - sm_ * tpq; r0[qm] = dgb ? 0.0 : sm_ * tpp + cm * tpq;
                r1[pm] = dgb ? 0.0 : cm * tqp - sm_ * tqq; r1[qm] = sm_ * tqp + cm * tqq;
            }
            __syncthreads();
        }
    }
    for (int i = t; i < n; i += NT) w[i] = A[i * np + i];
    __syncthreads();
    return gs;
}

// V = product of the logged rotations, V(:, k) = eigenvector of w[k].  Row i of V only ever mixes with itself, so each
// WAVE owns whole rows (LDS, ld = np) and replays the stages in program order: lanes = the disjoint pairs of a stage,
// no workgroup barrier inside.  The log is read 8 stages ahead to hide the L2 latency.
__device__ inline void jacobi_vectors(double* V, int n, const double2* rlog, int nstages) {
    const int t = threadIdx.x, NT = blockDim.x;
    const int np = (n + 1) & ~1, half = np >> 1, ring = np - 1;
    const int wave = t >> 6, lane = t & 63, NW = NT >> 6;
    for (int e = t; e < np * np; e += NT) V[e] = (e % (np + 1)) == 0 ? 1.0 : 0.0;
    __syncthreads();
    int st = 0;
    const int k0 = lane, k1 = 64 + lane;                 // np/2 <= 68 pairs per stage: at most two per lane
    for (int g0 = 0; g0 < nstages; g0 += 8) {
        const int nu = nstages - g0 < 8 ? nstages - g0 : 8;
        double2 ra[8], rb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            ra[u] = (u < nu && k0 < half) ? rlog[(size_t)(g0 + u) * half + k0] : make_double2(1.0, 0.0);
            rb[u] = (u < nu && k1 < half) ? rlog[(size_t)(g0 + u) * half + k1] : make_double2(1.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // the pairs of one stage are disjoint, stages must be applied in order
            if (ra[u].y != 0.0) {
                int p, q;
                rr_pair(k0, st, np, p, q);
                for (int i = wave; i < n; i += NW) {
                    double* row = V + i * np;
                    const double vp = row[p], vq = row[q];
                    row[p] = ra[u].x * vp - ra[u].y * vq; row[q] = ra[u].y * vp + ra[u].x * vq;
                }
            }
            if (rb[u].y != 0.0) {
                int p, q;
                rr_pair(k1, st, np, p, q);
                for (int i = wave; i < n; i += NW) {
                    double* row = V + i * np;
                    const double vp = row[p], vq = row[q];
                    row[p] = rb[u].x * vp - rb[u].y * vq; row[q] = rb[u].y * vp + rb[u].x * vq;
                }
            }
            if (u < nu) { if (++st == ring) st = 0; }
        }
    }
    __syncthreads();
}

}  // namespace vd

__global__ __launch_bounds__(MARG_THREADS) void k_marg(MargDev M) {
    using namespace vd;
    __shared__ double sm[3 * 68 + 40];
    extern __shared__ double mlds[];     // np x np: the symmetric working copy during the rotations, then the eigenvectors
    const int t = threadIdx.x, NT = blockDim.x;
    const int D = M.D, nd = M.nd, n = M.n;
    const int npd = (nd + 1) & ~1, np = (n + 1) & ~1;
    // ---- dropped block, symmetrised (marginalization_factor.cpp:273), eigen pseudo inverse ------------
    for (int e = t; e < npd * npd; e += NT) {
        const int i = e / npd, j = e - i * npd;
        mlds[e] = (i < nd && j < nd) ? 0.5 * (M.S[(size_t)M.drop_cols[i] * D + M.drop_cols[j]] + M.S[(size_t)M.drop_cols[j] * D + M.drop_cols[i]]) : 0.0;
    }
    __syncthreads();
    double2* rlog_d = M.rlog + (size_t)MARG_MAX_SWEEPS * (np - 1) * (np >> 1);     // own region: never aliases the big problem's log
    const int nsd = jacobi_eig(mlds, M.wd, nd, rlog_d, sm);
    __threadfence_block();
    jacobi_vectors(mlds, nd, rlog_d, nsd);
    if (t == 0) M.stat[0] = nsd;
    for (int e = t; e < nd * nd; e += NT) { const int i = e / nd, j = e - i * nd; M.Vd[e] = mlds[i * npd + j]; }
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
    for (int e = t; e < np * np; e += NT) { const int i = e / np, j = e - i * np; mlds[e] = (i < n && j < n) ? M.J0[(size_t)i * n + j] : 0.0; }
    __syncthreads();
    const int ns = jacobi_eig(mlds, M.w, n, M.rlog, sm);
    __threadfence_block();
    jacobi_vectors(mlds, n, M.rlog, ns);
    if (t == 0) M.stat[1] = ns;
    for (int e = t; e < n * n; e += NT) { const int i = e / n, j = e - i * n; M.V[e] = mlds[(size_t)i * np + j]; }
    __syncthreads();
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
