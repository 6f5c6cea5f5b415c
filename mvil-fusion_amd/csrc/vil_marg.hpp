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
    unsigned long long* ts;     // 16 wall-clock stamps (100 MHz) of the phases of the three launches (vil_debug_marg_stamps): k_marg 0 .. 6, k_marg_fast 7 .. 10, k_marg phase 1 11 .. 12
};
#define MSTAMP(k) do { if (threadIdx.x == 0 && M.ts) M.ts[k] = wall_clock64(); } while (0)

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

// phase 0: dropped block, Schur complement -> A, b (and the symmetrised copy in J0).  phase 1: pivoted square root, unless the
// fast path (k_marg_fast, between the two) already produced J0 / r0 (stat[2] == 1).
__global__ __launch_bounds__(MARG_THREADS) void k_marg(MargDev M, int phase) {
    using namespace vd;
    __shared__ double sm[3 * MARG_NMAX + 64];
    __shared__ double bsh[MARG_NMAX];
    extern __shared__ double mlds[];     // n x n: the symmetric matrix, then its orthogonalised factor rows
    const int t = threadIdx.x, NT = blockDim.x;
    const int D = M.D, nd = M.nd, n = M.n;
    if (phase == 1) {
        MSTAMP(11);
        if (M.stat[2] == 1) return;                      // the un-pivoted factorisation went through
        for (int e = t; e < n * n; e += NT) mlds[e] = M.J0[e];
        if (t < n) bsh[t] = M.b[t];
        __syncthreads();
        const int ns = gram_sqrt(mlds, M.w, n, 1e-30, 1e-10, false, bsh, M.r0, sm);
        if (t == 0) M.stat[1] = ns;
        for (int e = t; e < n * n; e += NT) {
            const int j = e / n, k = e - j * n;                   // column-major element (k, j): J0[j*n + k]
            M.J0[e] = mlds[k * n + j];
        }
        MSTAMP(12);
        return;
    }
    MSTAMP(0);
    // ---- dropped block, symmetrised (marginalization_factor.cpp:273), eigen pseudo inverse ------------
    for (int e = t; e < nd * nd; e += NT) {
        const int i = e / nd, j = e - i * nd;
        mlds[e] = 0.5 * (M.S[(size_t)M.drop_cols[i] * D + M.drop_cols[j]] + M.S[(size_t)M.drop_cols[j] * D + M.drop_cols[i]]);
    }
    __syncthreads();
    // Fast path for the dropped block: when every eigenvalue of A_dd is safely above eps the thresholded pseudo inverse of
    // marginalization_factor.cpp:277 IS the inverse, and a 15 x 15 Cholesky inverse costs a few microseconds where the
    // eigen-decomposition (one-sided Jacobi, a workgroup barrier per rotation stage) costs ~100.  Certificate:
    // lambda_min = 1 / lambda_max(A_dd^-1) >= 1 / ||A_dd^-1||_F.  Anything else takes the eigen route below.
    MSTAMP(1);
    __shared__ int fast_ok;
    double* W = sm; double* Li = sm + 225; double* Wr = sm + 450;      // nd <= 15: factor, its inverse, reciprocal pivots (sm holds 3 * 136 + 64 doubles)
    if (t == 0) fast_ok = nd <= 15 ? 1 : 0;
    for (int e = t; e < nd * nd; e += NT) W[e] = mlds[e];
    __syncthreads();
    // (ONE wave factors the block, wave-synchronously -- LDS operations of a wave execute in order, a wait for them is all a step needs: 15 pivots in ~3 us.  With the
    //  whole workgroup, three barriers of 1024 threads per pivot, the same loop took 14.7 us of every marginalisation: profiles/r06_marg_phases.txt)
    if (t < 64 && fast_ok) {
        bool okw = true;
        for (int p = 0; p < nd; ++p) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const double d = W[p * nd + p];
            if (!(d > 0.0) || !isfinite(d)) { okw = false; break; }      // (uniform: every lane read the same entry)
            const double r = rsqrt_nr(d);
            if (t == 0) Wr[p] = r;
            if (t >= p && t < nd) W[t * nd + p] *= r;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            for (int e = t; e < nd * nd; e += 64) { const int i = e / nd, j = e - i * nd; if (j > p && i >= j) W[e] -= W[i * nd + p] * W[j * nd + p]; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (!okw && t == 0) fast_ok = 0;
    }
    __syncthreads();
    if (fast_ok) {
        if (t < nd) {                                    // column t of L^-1 by forward substitution
            const int j = t;
            for (int i = 0; i < nd; ++i) {
                double acc = i == j ? 1.0 : 0.0;
                for (int k = j; k < i; ++k) acc -= W[i * nd + k] * Li[k * nd + j];
                Li[i * nd + j] = i < j ? 0.0 : acc * Wr[i];
            }
        }
        __syncthreads();
        double f2 = 0.0;
        for (int e = t; e < nd * nd; e += NT) {           // A_dd^-1 = L^-T L^-1 -> Vd (scratch)
            const int i = e / nd, j = e - i * nd;
            double acc = 0.0;
            for (int k = (i > j ? i : j); k < nd; ++k) acc += Li[k * nd + i] * Li[k * nd + j];
            M.Vd[e] = acc; W[e] = acc; f2 += acc * acc;          // (W: the factor is not read any more)
        }
        f2 = wave_total(f2);
        __shared__ double f2w[16];
        if ((t & 63) == 0) f2w[t >> 6] = f2;
        __syncthreads();
        if (t == 0) { double tot = 0.0; for (int q = 0; q < (NT >> 6); ++q) tot += f2w[q]; if (!(tot > 0.0) || !(rsqrt_nr(tot) > 4.0 * M.eps)) fast_ok = 0; }
        __syncthreads();
    }
    // staged = the products below run out of LDS: the kept x dropped blocks of S, A_dd^-1 and T are gathered once, with every load of a
    // thread independent of the others -- the direct loops chase index -> S entry -> product nd times in a row per output element
    // (n = 70: some 75 dependent L2 round trips per thread, most of this kernel's time)
    MSTAMP(2);
    const bool staged = fast_ok && (size_t)n * n >= 3 * (size_t)n * nd;
    double* Skd = mlds; double* Sdk = mlds + (size_t)n * nd; double* Tl = mlds + 2 * (size_t)n * nd;
    if (staged) {
        if (t == 0) M.stat[0] = 0;
        const double* Ai = W;                            // A_dd^-1 (written over the factor above)
        for (int e = t; e < n * nd; e += NT) {
            const int i = e / nd, q = e - i * nd;
            Skd[e] = M.S[(size_t)M.keep_cols[i] * D + M.drop_cols[q]];
            Sdk[(size_t)q * n + i] = M.S[(size_t)M.drop_cols[q] * D + M.keep_cols[i]];
        }
        __syncthreads();
        MSTAMP(3);
        for (int e = t; e < n * nd; e += NT) {           // T = A_kd A_dd^-1
            const int i = e / nd, j = e - i * nd;
            double acc = 0.0;
            for (int q = 0; q < nd; ++q) acc += Skd[i * nd + q] * Ai[q * nd + j];
            Tl[e] = acc; M.T[e] = acc;
        }
        __syncthreads();
    } else if (fast_ok) {
        if (t == 0) M.stat[0] = 0;
        for (int e = t; e < n * nd; e += NT) {           // T = A_kd A_dd^-1
            const int i = e / nd, j = e - i * nd;
            double acc = 0.0;
            for (int q = 0; q < nd; ++q) acc += M.S[(size_t)M.keep_cols[i] * D + M.drop_cols[q]] * M.Vd[(size_t)q * nd + j];
            M.T[e] = acc;
        }
        __syncthreads();
    } else {
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
    }
    MSTAMP(4);
    // A = A_kk - T A_dk ; b = b_k - T b_d      (marginalization_factor.cpp:289-290)
    for (int e = t; e < n * n + n; e += NT) {
        if (e < n * n) {
            const int i = e / n, j = e - i * n;
            double s = M.S[(size_t)M.keep_cols[i] * D + M.keep_cols[j]];
            if (staged) { for (int k = 0; k < nd; ++k) s -= Tl[i * nd + k] * Sdk[(size_t)k * n + j]; }
            else for (int k = 0; k < nd; ++k) s -= M.T[(size_t)i * nd + k] * M.S[(size_t)M.drop_cols[k] * D + M.keep_cols[j]];
            M.V[e] = s;                                    // staged in V, symmetrised copy goes to A below
        } else {
            const int i = e - n * n;
            double s = M.g[M.keep_cols[i]];
            for (int k = 0; k < nd; ++k) s -= M.T[(size_t)i * nd + k] * M.g[M.drop_cols[k]];
            M.b[i] = s;
        }
    }
    __syncthreads();
    MSTAMP(5);
    // Eigen's SelfAdjointEigenSolver reads the lower triangle only
    for (int e = t; e < n * n; e += NT) { const int i = e / n, j = e - i * n; M.J0[e] = i >= j ? M.V[e] : M.V[(size_t)j * n + i]; }
    __syncthreads();
    for (int e = t; e < n * n; e += NT) { M.A[e] = M.V[e]; }
    __syncthreads();
    if (t == 0) M.stat[2] = 0;
    MSTAMP(6);
    // linearized_jacobians / linearized_residuals (marginalization_factor.cpp:301-309) are ANY pair with J0^T J0 = A and
    // J0^T r0 = b: the reference takes sqrt(S) V^T from an eigen-decomposition, whose basis is implementation-defined
    // (SURVEY App. C #12).  Here J0 = G^T from a Cholesky A = G G^T and r0 = G^-1 b: the prior residual r0 + J0 dx is the
    // reference's up to a left orthogonal factor, i.e. the same cost, gradient and Gauss-Newton matrix.  k_marg_fast tries the
    // un-pivoted blocked factorisation of the step kernel first (matrix cores, ~25 us at n = 70); if a pivot falls to its
    // rounding level -- where the reference drops eigenvalues below eps -- phase 1 redoes it with diagonal pivoting (gram_sqrt).
}

// Fast path of the square root: A (symmetrised, in J0) = L L^T with chol_blocked (vil_step.hpp), r0 = L^-1 b as the extra row.
// Accepted only if every pivot clears the threshold gram_sqrt would apply to it; then J0 = L^T (column-major), r0, stat[2] = 1.
__global__ __launch_bounds__(VIL_STEP_THREADS) void k_marg_fast(MargDev M) {
    using namespace vd;
    __shared__ StepShared s;
    extern __shared__ double tl[];
    const int t = threadIdx.x, NT = blockDim.x, n = M.n;
    MSTAMP(7);
    for (int q = t; q < 256; q += NT) {
        int Ir = (int)((sqrtf(8.f * (float)q + 1.f) - 1.f) * 0.5f);
        if (((Ir + 1) * (Ir + 2)) / 2 <= q) ++Ir;
        if ((Ir * (Ir + 1)) / 2 > q) --Ir;
        s.tI[q] = (unsigned char)Ir; s.tJ[q] = (unsigned char)(q - (Ir * (Ir + 1)) / 2);
    }
    if (t == 0) s.ok = 1;
    __syncthreads();
    const int R = n + 1, T = (R + 15) >> 4, NE = ((T * (T + 1)) >> 1) << 8;
    double dmax = 0.0;
    for (int i = t; i < n; i += NT) dmax = fmax(dmax, M.J0[(size_t)i * n + i]);
    dmax = bmax(dmax, s);
    for (int e = t; e < NE; e += NT) {
        const int tile = e >> 8, w = e & 255;
        const int i = (s.tI[tile] << 4) + (w >> 4), j = (s.tJ[tile] << 4) + (w & 15);
        double v = 0.0;
        if (i < n && j <= i) v = M.J0[(size_t)i * n + j];
        else if (i == n && j < n) v = M.b[j];
        tl[tl_phys(e)] = v;
    }
    for (int i = t; i < n; i += NT) {                    // smallest acceptable pivot of column i (the rule of gram_sqrt)
        const double rel = fmax(16.0 * n * 2.220446049250313e-16 * M.J0[(size_t)i * n + i], 1e-10 * dmax);
        s.gr[i] = rel > 1e-30 ? rel : 1e-30;
    }
    __syncthreads();
    MSTAMP(8);
    bool ok = chol_lookahead(tl, n, s);
    __syncthreads();
    MSTAMP(9);
    if (ok) {
        for (int i = t; i < n; i += NT) { const double l = 1.0 / s.dinv[i]; if (!(l * l > s.gr[i])) s.ok = 0; }
        __syncthreads();
        ok = s.ok != 0;
    }
    if (!ok) return;                                     // stat[2] stays 0: phase 1 of k_marg takes over
    for (int e = t; e < n * n; e += NT) {
        const int j = e / n, k = e - j * n;              // column-major element (k, j) of J0 = L^T: L[j][k] for k <= j
        M.J0[e] = k <= j ? tl[tl_idx(j, k)] : 0.0;
    }
    for (int i = t; i < n; i += NT) M.r0[i] = tl[tl_idx(n, i)];
    if (t == 0) { M.stat[2] = 1; M.stat[1] = 0; }
    MSTAMP(10);
}
