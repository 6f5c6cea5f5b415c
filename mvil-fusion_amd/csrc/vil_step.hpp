// The trust-region step kernel: ONE workgroup runs, per iteration and without any host round trip,
// what ceres' TrustRegionMinimizer + DoglegStrategy + the dense Cholesky of DENSE_SCHUR do around
// the sweep (estimator.cpp:1400-1414; algorithm restated in SURVEY.md Appendix B):
//   judge the candidate just swept (function tolerance, relative decrease, radius update) ->
//   on a new linearisation: Jacobi/dogleg scaling, gradient, Cauchy point, reduced Cholesky solve,
//   landmark back-substitution -> traditional dogleg blend -> model decrease -> candidate state.
// All trust-region state lives in `Ctl` in device memory; `done` makes later launches no-ops.
//
// Dense solve: the scaled reduced matrix M = Sc S' Sc + mu dc^2 (D x D, D = 15K+7) plus the
// right-hand side as an extra row is held PACKED-LOWER in LDS (100 KB at K = 10) and factored by a
// right-looking blocked Cholesky, NB = 8: wave 0 factors and inverts the 8x8 diagonal block in
// registers (wave shuffles), one thread per row applies the inverse to the panel, and the trailing
// update -- the only dense contraction on this path -- runs on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64, one 16x16 tile per wave per step).  The forward substitution is the
// factorisation of the extra row.  Windows whose packed matrix exceeds LDS (K >= 14) use the same
// code on a packed global (L2-resident) buffer.
#pragma once
#include "vil_dev.hpp"
#include "vil_factors.hpp"

namespace vd {

#define STEP_NB 8

struct StepShared {
    Ctl c;
    double red[32];
    double X[64 * 64];      // inverse of every 8x8 diagonal block of L (<= 64 blocks, D <= 512)
    double y[512];
    double sc[512], dcs[512], gr[512], gn[512];   // Sc, dogleg diagonal, gradient_, gauss_newton_step_ (camera part)
    int need, was_first, ok;
    long long tacc[3];
};

// block-wide sum(a), sum(b) and sum-or-max(c) with one pair of barriers
template <bool CMAX = false>
__device__ __forceinline__ void bsum3(double& a, double& b, double& c, StepShared& s) {
    a = wave_sum(a); b = wave_sum(b);
    if (CMAX) { for (int o = 32; o > 0; o >>= 1) c = fmax(c, __shfl_xor(c, o, 64)); } else c = wave_sum(c);
    __syncthreads();
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { s.red[w] = a; s.red[8 + w] = b; s.red[16 + w] = c; }
    __syncthreads();
    a = 0; b = 0; c = 0;
    for (int q = 0; q < nw; ++q) { a += s.red[q]; b += s.red[8 + q]; c = CMAX ? fmax(c, s.red[16 + q]) : c + s.red[16 + q]; }
}

__device__ __forceinline__ double bsum(double v, StepShared& s) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) t += s.red[w];
    return t;
}
__device__ __forceinline__ double bmax(double v, StepShared& s) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) t = fmax(t, s.red[w]);
    return t;
}

// e_l . v_c  for landmark l (compact e storage: anchor 6 | ex 6 | td 1 | per-factor observer 6)
__device__ __forceinline__ double lm_dot(const DevP& P, const SysBuf& sb, int l, const double* vc) {
    const int fs = P.lm_start[l], fe = P.lm_start[l + 1];
    if (fe == fs) return 0.0;
    const double* e = sb.eA + (size_t)l * 13;
    const int a = P.vis_i[fs];
    double s = 0;
    const double* va = vc + col_pose(P, a); const double* vx = vc + col_ex(P);
#pragma unroll
    for (int k = 0; k < 6; ++k) s += e[k] * va[k] + e[6 + k] * vx[k];
    s += e[12] * vc[col_td(P)];
    for (int f = fs; f < fe; ++f) {
        const double* eo = sb.eO + (size_t)f * 6; const double* vj = vc + col_pose(P, P.vis_j[f]);
#pragma unroll
        for (int k = 0; k < 6; ++k) s += eo[k] * vj[k];
    }
    return s;
}

// v^T H v over all free parameters, H = J^T J of the corrected Jacobian, from the reduced pieces:
//   v_c^T H_cc v_c = v_c^T S' v_c + sum_l invp (e_l.v_c)^2     (S' = H_cc - sum_l invp e e^T)
// vc must be readable by every thread (LDS or global).
__device__ __forceinline__ double quad_form(const DevP& P, const SysBuf& sb, const double* vc, const double* vl, StepShared& s) {
    const int D = P.D, L = P.L, t = threadIdx.x, NT = blockDim.x;
    double part = 0;
    for (int e = t; e < D * D; e += NT) { const int i = e / D, j = e - i * D; part += vc[i] * sb.S[e] * vc[j]; }
    for (int l = t; l < L; l += NT) {
        const double ip = sb.invp[l];
        if (ip == 0.0) continue;
        const double ev = lm_dot(P, sb, l, vc);
        part += ip * ev * ev + 2.0 * vl[l] * ev + sb.hll[l] * vl[l] * vl[l];
    }
    return bsum(part, s);
}

__device__ __forceinline__ void pose_plus(const double* in, const double* d, double* o) {
    for (int k = 0; k < 3; ++k) o[k] = in[k] + d[k];
    Q4 q = qmul(qload(in + 3), Q4{1.0, 0.5 * d[3], 0.5 * d[4], 0.5 * d[5]});
    const double n = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    o[3] = q.x * n; o[4] = q.y * n; o[5] = q.z * n; o[6] = q.w * n;
}

typedef double d4 __attribute__((ext_vector_type(4)));

// broadcast lane `L` (compile-time constant) of a double through SGPRs: v_readlane_b32 x2, no LDS round trip
template <int L>
__device__ __forceinline__ double bcast(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), L);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), L);
    return __hiloint2double(hi, lo);
}
template <int J, int C>
struct DiagUpd {   // trailing update inside the 8x8 diagonal block for column J, target columns C..7
    static __device__ __forceinline__ void run(double (&a)[STEP_NB], int r) {
        const double lcj = bcast<C>(a[J]);
        if (r >= C) a[C] -= a[J] * lcj;
        if constexpr (C + 1 < STEP_NB) DiagUpd<J, C + 1>::run(a, r);
    }
};
// sqrt(x) and 1/sqrt(x) together: hardware rsq seed + two coupled Newton steps (no divide on the pivot chain)
__device__ __forceinline__ void sqrt_rsqrt(double x, double& sq, double& rs) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    sq = g; rs = h + h;
}
template <int J>
struct DiagCol {
    static __device__ __forceinline__ void run(double (&a)[STEP_NB], double (&dinv)[STEP_NB], int r, bool& ok) {
        const double pj = bcast<J>(a[J]);
        if (!(pj > 0.0) || !isfinite(pj)) ok = false;
        double dj, inv;
        sqrt_rsqrt(pj, dj, inv);
        dinv[J] = inv;
        if (r == J) a[J] = dj; else if (r > J) a[J] *= inv;
        if constexpr (J + 1 < STEP_NB) { DiagUpd<J, J + 1>::run(a, r); DiagCol<J + 1>::run(a, dinv, r, ok); }
    }
};
template <int RR, int K>
struct InvDot {
    static __device__ __forceinline__ void run(const double (&a)[STEP_NB], const double (&x)[STEP_NB], double& sum) {
        if constexpr (K < RR) { sum += bcast<RR>(a[K]) * x[K]; InvDot<RR, K + 1>::run(a, x, sum); }
    }
};
template <int RR>
struct InvRow {
    static __device__ __forceinline__ void run(const double (&a)[STEP_NB], const double (&dinv)[STEP_NB], double (&x)[STEP_NB], int cc) {
        double sum = 0.0;
        InvDot<RR, 0>::run(a, x, sum);
        x[RR] = (RR == cc) ? dinv[RR] : (RR > cc ? -sum * dinv[RR] : 0.0);
        if constexpr (RR + 1 < STEP_NB) InvRow<RR + 1>::run(a, dinv, x, cc);
    }
};

__device__ __forceinline__ int tri_off(int i) { return (i * (i + 1)) >> 1; }

// Blocked Cholesky of the packed-lower (D+1) x (D+1) array A whose last row is the right-hand side:
// on return rows < D hold L, row D holds y = L^-1 rhs, s.X the inverses of the diagonal blocks.
// Returns false (uniformly) if a pivot is not positive.
template <class PTR>
__device__ __forceinline__ bool chol_blocked(PTR A, int D, StepShared& s) {
    const int t = threadIdx.x, NT = blockDim.x, wave = t >> 6, lane = t & 63, NW = NT >> 6;
    const int R = D + 1;   // rows including the rhs row
    long long tacc[3] = {0, 0, 0}, tprev = 0;
    #define CSTAMP(k) do { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); if (k >= 0) tacc[k < 0 ? 0 : k] += tt_ - tprev; tprev = tt_; } while (0)
    for (int kb = 0, blk = 0; kb < D; kb += STEP_NB, ++blk) {
        const int nb = min(STEP_NB, D - kb);
        CSTAMP(-1);
        // ---- 1. diagonal block: factor + invert in registers of wave 0 --------------------------
        if (wave == 0) {
            const int r = lane & 7;
            double a[STEP_NB];
#pragma unroll
            for (int c = 0; c < STEP_NB; ++c) a[c] = (r < nb && c <= r && c < nb) ? A[tri_off(kb + r) + kb + c] : (c == r ? 1.0 : 0.0);
            bool ok = true;
            double dinv[STEP_NB];
            DiagCol<0>::run(a, dinv, r, ok);
            // inverse: lane c computes column c of X = L^-1
            double x[STEP_NB];
#pragma unroll
            for (int q = 0; q < STEP_NB; ++q) x[q] = 0.0;
            const int cc = lane & 7;
            InvRow<0>::run(a, dinv, x, cc);
            if (lane < STEP_NB) {
#pragma unroll
                for (int c = 0; c < STEP_NB; ++c) {
                    if (lane < nb && c <= lane && c < nb) A[tri_off(kb + lane) + kb + c] = a[c];
                }
#pragma unroll
                for (int rr = 0; rr < STEP_NB; ++rr) s.X[blk * 64 + rr * 8 + lane] = x[rr];   // X[rr][cc]
            }
            if (lane == 0) s.ok = ok ? 1 : 0;
        }
        __syncthreads();
        CSTAMP(0);
        if (!s.ok) return false;
        // ---- 2. panel: rows below the block (incl. rhs row), L_i = A_i X^T ------------------------
        const int r0 = kb + nb;
        for (int i = r0 + t; i < R; i += NT) {
            double a[STEP_NB], o[STEP_NB];
            const int base = tri_off(i) + kb;
#pragma unroll
            for (int c = 0; c < STEP_NB; ++c) a[c] = c < nb ? A[base + c] : 0.0;
#pragma unroll
            for (int c = 0; c < STEP_NB; ++c) { double v = 0;
#pragma unroll
                for (int k = 0; k <= c; ++k) v += a[k] * s.X[blk * 64 + c * 8 + k]; o[c] = v; }
#pragma unroll
            for (int c = 0; c < STEP_NB; ++c) if (c < nb) A[base + c] = o[c];
        }
        __syncthreads();
        CSTAMP(1);
        // ---- 3. trailing update on the fp64 matrix cores: C[r][c] -= sum_k L[r][kb+k] L[c][kb+k] -----
        const int rem = R - r0;                // rows r0 .. D (rhs row included)
        if (rem > 0) {
            const int nt = (rem + 15) >> 4;
            const int ntile = nt * (nt + 1) / 2;
            // software-pipelined over up to 8 tiles per wave: all operand loads, then the MFMAs, then the read-modify-writes
            for (int t0 = 0; t0 < ntile; t0 += 8 * NW) {
                d4 acc[8]; double av[8][2], bv[8][2]; int rowb[8], colb[8];
                const int cnt = min(8, (ntile - t0 - wave + NW - 1) / NW);   // wave-uniform number of live slots
#pragma unroll
                for (int u = 0; u < 8; ++u) if (u < cnt) {
                    const int tile = t0 + wave + u * NW;
                    int I = (int)((sqrtf(8.f * (float)tile + 1.f) - 1.f) * 0.5f);
                    if (((I + 1) * (I + 2)) / 2 <= tile) ++I;
                    if ((I * (I + 1)) / 2 > tile) --I;
                    const int J = tile - (I * (I + 1)) / 2;
                    const bool valid = tile < ntile;
                    rowb[u] = valid ? r0 + 16 * I : R; colb[u] = valid ? r0 + 16 * J : D;
                    const int rr = rowb[u] + (lane & 15), cr = colb[u] + (lane & 15);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int k = 4 * h + (lane >> 4);
                        av[u][h] = (rr < R && k < nb) ? A[tri_off(rr) + kb + k] : 0.0;
                        bv[u][h] = (cr < D && k < nb) ? A[tri_off(cr) + kb + k] : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (u < cnt) { d4 z = {0.0, 0.0, 0.0, 0.0}; acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][0], bv[u][0], z, 0, 0, 0); }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (u < cnt) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u][1], bv[u][1], acc[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 8; ++u) if (u < cnt) {
                    const int col = colb[u] + (lane & 15);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int row = rowb[u] + (lane >> 4) + 4 * g;
                        if (row < R && col < D && col <= row) A[tri_off(row) + col] -= acc[u][g];
                    }
                }
            }
        }
        __syncthreads();
        CSTAMP(2);
    }
    if (threadIdx.x == 0) { s.tacc[0] = tacc[0]; s.tacc[1] = tacc[1]; s.tacc[2] = tacc[2]; }
    return true;
}

// back substitution L^T x = y (y = row D of A), blocked; result in s.y[0..D)
template <class PTR>
__device__ __forceinline__ void back_subst(PTR A, int D, StepShared& s) {
    const int t = threadIdx.x, NT = blockDim.x;
    for (int i = t; i < D; i += NT) s.y[i] = A[tri_off(D) + i];
    __syncthreads();
    const int nblk = (D + STEP_NB - 1) / STEP_NB;
    for (int blk = nblk - 1; blk >= 0; --blk) {
        const int kb = blk * STEP_NB, nb = min(STEP_NB, D - kb);
        // x_blk = X^T y_blk
        double xv = 0.0;
        if (t < nb) { for (int k = t; k < nb; ++k) xv += s.X[blk * 64 + k * 8 + t] * s.y[kb + k]; }
        __syncthreads();
        if (t < nb) s.y[kb + t] = xv;
        __syncthreads();
        // y_c -= sum_r L[kb+r][c] x_r  for c < kb
        for (int c = t; c < kb; c += NT) {
            double v = s.y[c];
            for (int r = 0; r < nb; ++r) v -= A[tri_off(kb + r) + c] * s.y[kb + r];
            s.y[c] = v;
        }
        __syncthreads();
    }
}

}  // namespace vd

// PHASE 0: whole step (single GPU).  PHASE 1 / 2: the step split around the all-reduce of the landmark-dependent
// scalars (multi-GPU: every rank owns a slice of the landmarks).
template <bool LDSM, int PHASE>
__global__ __launch_bounds__(VIL_STEP_THREADS) void k_step(DevP P, SolveOpts O) {
    using namespace vd;
    __shared__ StepShared s;
    extern __shared__ double Alds[];
    const int t = threadIdx.x, NT = blockDim.x;
    const int D = P.D, L = P.L;
    if (t == 0) { s.c = *P.ctl; s.need = 0; s.was_first = 0; s.ok = 1; }
    __syncthreads();
    #define STAMP(k) do { __syncthreads(); if (t == 0) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); P.dbg[k] = tt_; } } while (0)
    if (s.c.done) return;
    const bool multi = P.split != 0;
    const bool cam = P.world <= 1 || P.rank == 0;      // camera-side terms of global sums are counted once
    STAMP(0);
    if (PHASE == 1) {   // the all-reduced candidate system arrives in the staging buffer
        double* dst = P.sys[1 - s.c.cur].ar;
        const int cnt = D * D + 3 * D + 3;
        for (int e = t; e < cnt; e += NT) dst[e] = P.arstage[e];
        __syncthreads();
    }
    if (PHASE == 2) {
        if (s.c.skip_b) { if (t == 0) { s.c.skip_b = 0; *P.ctl = s.c; } return; }
        if (t == 0 && s.c.phase_need) {
            Ctl& c = s.c;
            const double g2 = P.scal[0], q = P.scal[1], gm = P.scal[2];
            if (gm <= O.gradient_tolerance) { c.done = 1; c.term = 2; }
            c.alpha = g2 / q; c.mu_used = c.mu; c.gn2 = P.scal[3]; c.g2 = g2; c.gg = P.scal[4];
            c.mu = fmax(O.min_mu, 2.0 * c.mu / 10.0);
            c.phase_need = 0;
        }
        __syncthreads();
        if (s.c.done) { if (t == 0) *P.ctl = s.c; return; }
    }
    // ---------------- judge the candidate that the sweep just linearised -------------------------
    if (PHASE != 2 && t == 0) {
        Ctl& c = s.c;
        const int cand = 1 - c.cur;
        const double cand_cost = *P.sys[cand].cost;
        c.cand_cost = cand_cost;
        if (multi && !c.first && !c.resweep) {     // parameter tolerance of the step just evaluated (norms all-reduced with S)
            const double* tail = P.sys[cand].ar + (size_t)P.D * P.D + 3 * P.D + 1;
            if (sqrt(tail[1]) <= O.parameter_tolerance * (sqrt(tail[0]) + O.parameter_tolerance)) { c.done = 1; c.term = 3; }
        }
        if (c.done) {}
        else if (c.first || c.resweep) {
            if (c.first) { c.initial_cost = cand_cost; s.was_first = 1; }
            if (!isfinite(cand_cost)) { c.done = 1; c.term = 6; c.status = -3; }
            c.cur = cand; c.cost_cur = cand_cost; c.first = 0; c.resweep = 0; s.need = 1;
        } else {
            if (fabs(c.cost_cur - cand_cost) <= O.function_tolerance * c.cost_cur) { c.done = 1; c.term = 1; }
            else {
                const double rel = (c.cost_cur - cand_cost) / c.model_change;
                if (isfinite(cand_cost) && rel > O.min_relative_decrease) {
                    c.cur = cand; c.cost_cur = cand_cost; c.nsucc++;
                    if (rel < 0.25) c.radius *= 0.5;
                    if (rel > 0.75) c.radius = fmax(c.radius, 3.0 * c.dogleg_norm);
                    c.radius = fmin(O.max_radius, c.radius);
                    c.reuse = 0; s.need = 1;
                } else { c.radius *= 0.5; c.reuse = 1; s.need = 0; }
            }
            if (c.iter >= 1 && c.iter <= 64) c.cost_trace[c.iter - 1] = c.cost_cur;
        }
        if (!c.done) {
            if (c.iter >= O.max_iterations) { c.done = 1; c.term = 4; }
            else if (c.radius <= 1e-32) { c.done = 1; c.term = 6; c.status = -4; }
        }
    }
    __syncthreads();
    const int cur = s.c.cur;
    SysBuf sb = P.sys[cur];
    const double* x = P.x[cur];
    double* xc = P.x[1 - cur];
    if (s.c.done) { if (t == 0) *P.ctl = s.c; return; }
    STAMP(1);
    double gn2 = 0, g2 = 0, gg = 0;
    if (PHASE != 2 && s.need) {
        // ---- camera vectors: Jacobi scaling (first linearisation), dogleg diagonal, gradient_, u = Sc gradient_/d
        double gm = 0;
        for (int i = t; i < D; i += NT) {
            const double dg = sb.diag[i], b = sb.bc[i];
            double Sc;
            if (s.was_first) { Sc = O.jacobi_scaling ? 1.0 / (1.0 + sqrt(dg)) : 1.0; P.Sc[i] = Sc; } else Sc = P.Sc[i];
            const double d = sqrt(fmin(fmax(Sc * Sc * dg, 1e-6), 1e32));
            const double g = Sc * b / d;
            s.sc[i] = Sc; s.dcs[i] = d; s.gr[i] = g; s.y[i] = Sc * g / d;
            P.dc[i] = d; P.gradc[i] = g;
            if (cam) { g2 += g * g; gm = fmax(gm, fabs(b)); }
        }
        __syncthreads();
        STAMP(9);
        // ---- one pass over the landmarks: dl, gradient_, and their share of u^T H u --------------------------
        double q = 0;
        for (int l = t; l < L; l += NT) {
            const double ip = sb.invp[l], Sl = P.Sl[l], h = sb.hll[l], b = sb.bl[l];
            const double d = sqrt(fmin(fmax(Sl * Sl * h, 1e-6), 1e32));
            const double g = ip != 0.0 ? Sl * b / d : 0.0;
            P.dl[l] = d; P.gradl[l] = g;
            if (ip != 0.0) {
                const double ul = Sl * g / d;
                const double ev = lm_dot(P, sb, l, s.y);
                q += ip * ev * ev + 2.0 * ul * ev + h * ul * ul;
                g2 += g * g; gm = fmax(gm, fabs(b));
            }
        }
        STAMP(10);
        // ---- camera share of u^T H u, fused with packing M = Sc S' Sc + mu dc^2 (+ rhs row) into LDS ------------
        const double mu = s.c.mu;
        const int NL = tri_off(D);            // packed lower entries of rows 0..D-1 ; row D (rhs) follows
        double* Ag = P.M;
        for (int base = t; base < NL; base += 8 * NT) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * NT;
                if (idx < NL) {
                    int i = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
                    while (tri_off(i + 1) <= idx) ++i;
                    while (tri_off(i) > idx) --i;
                    v[u] = sb.S[(size_t)i * D + (idx - tri_off(i))];
                } else v[u] = 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * NT;
                if (idx < NL) {
                    int i = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
                    while (tri_off(i + 1) <= idx) ++i;
                    while (tri_off(i) > idx) --i;
                    const int j = idx - tri_off(i);
                    if (cam) q += (i == j ? 1.0 : 2.0) * s.y[i] * v[u] * s.y[j];
                    double m = s.sc[i] * v[u] * s.sc[j];
                    if (i == j) m += mu * s.dcs[i] * s.dcs[i];
                    if constexpr (LDSM) Alds[idx] = m; else Ag[idx] = m;
                }
            }
        }
        STAMP(11);
        for (int j = t; j < D; j += NT) { const double m = s.sc[j] * sb.gred[j]; if constexpr (LDSM) Alds[NL + j] = m; else Ag[NL + j] = m; }
        bsum3<true>(g2, q, gm, s);
        STAMP(2);
        if (PHASE == 0 && gm <= O.gradient_tolerance) { if (t == 0) { s.c.done = 1; s.c.term = 2; *P.ctl = s.c; } return; }
        bool ok;
        if constexpr (LDSM) ok = chol_blocked(Alds, D, s); else ok = chol_blocked(Ag, D, s);
        STAMP(3);
        if (t == 0) { P.dbg[20] = s.tacc[0]; P.dbg[21] = s.tacc[1]; P.dbg[22] = s.tacc[2]; }
        if (!ok) {
            // dogleg_strategy.cc: mu *= 10 and retry; the Schur pivots depend on mu, so re-sweep at x_cur
            if (t == 0) {
                Ctl& c = s.c;
                c.mu *= 10.0;
                if (!(c.mu < O.max_mu)) { c.iter++; c.invalid_run++; c.reuse = 0; if (c.invalid_run >= 5) { c.done = 1; c.term = 6; c.status = -4; } }
                c.resweep = 1; c.skip_b = (PHASE == 1) ? 1 : 0;
            }
            for (int i = t; i < P.NS; i += NT) xc[i] = x[i];
            __syncthreads();
            if (t == 0) *P.ctl = s.c;
            return;
        }
        if constexpr (LDSM) back_subst(Alds, D, s); else back_subst(Ag, D, s);
        STAMP(4);
        // ---- gauss-newton step in dogleg space; landmark back-substitution fused with the dogleg sums ------------
        for (int i = t; i < D; i += NT) {
            const double xi = s.y[i];
            const double gnv = -xi * s.dcs[i];
            s.gn[i] = gnv; P.gnc[i] = gnv;
            if (cam) { gn2 += gnv * gnv; gg += gnv * s.gr[i]; }
            s.y[i] = s.sc[i] * xi;     // Sc x_c for the landmark back-substitution
        }
        __syncthreads();
        for (int l = t; l < L; l += NT) {
            const double ip = sb.invp[l];
            double gnv = 0.0;
            if (ip != 0.0) {
                const double xl = (sb.bl[l] - lm_dot(P, sb, l, s.y)) * ip / P.Sl[l];
                gnv = -xl * P.dl[l];
                gn2 += gnv * gnv; gg += gnv * P.gradl[l];
            }
            P.gnl[l] = gnv;
        }
        double dummy = 0;
        bsum3(gn2, gg, dummy, s);
        if (PHASE == 1) {
            if (t == 0) { P.scal[0] = g2; P.scal[1] = q; P.scal[2] = gm; P.scal[3] = gn2; P.scal[4] = gg; s.c.phase_need = 1; *P.ctl = s.c; }
            return;
        }
        if (t == 0) {
            Ctl& c = s.c;
            c.alpha = g2 / q; c.mu_used = c.mu; c.gn2 = gn2; c.g2 = g2; c.gg = gg;
            c.mu = fmax(O.min_mu, 2.0 * c.mu / 10.0);
        }
        __syncthreads();
    } else {
        if (PHASE == 1) { if (t == 0) { for (int k = 0; k < 5; ++k) P.scal[k] = 0.0; s.c.phase_need = 0; *P.ctl = s.c; } return; }
        for (int i = t; i < D; i += NT) { s.sc[i] = P.Sc[i]; s.dcs[i] = P.dc[i]; s.gr[i] = P.gradc[i]; s.gn[i] = P.gnc[i]; }
        __syncthreads();
    }
    STAMP(5);
    // ---------------- traditional dogleg in dogleg space (scalars saved with the linearisation) ----------------
    gn2 = s.c.gn2; g2 = s.c.g2; gg = s.c.gg;
    const double radius = s.c.radius, alpha = s.c.alpha, mu_u = s.c.mu_used;
    const double gn_norm = sqrt(gn2), g_norm = sqrt(g2);
    double cg, cn, dnorm;
    if (gn_norm <= radius) { cg = 0; cn = 1; dnorm = gn_norm; }
    else if (g_norm * alpha >= radius) { cg = -(radius / g_norm); cn = 0; dnorm = radius; }
    else {
        const double b_dot_a = -alpha * gg;
        const double a2 = (alpha * g_norm) * (alpha * g_norm);
        const double bma2 = a2 - 2 * b_dot_a + gn2;
        const double cc = b_dot_a - a2;
        const double dd = sqrt(cc * cc + bma2 * (radius * radius - a2));
        const double beta = (cc <= 0) ? (dd - cc) / bma2 : (radius * radius - a2) / (dd + cc);
        cg = -alpha * (1 - beta); cn = beta; dnorm = radius;
    }
    // model decrease from the linear-algebra identities of the solved system (u = Sc gradient_/d, w = GN step):
    //   u^T H u = g2/alpha ; H w = -g - mu (d/S)^2 w  =>  u^T H w = -g2 - mu gg ,  w^T H w = -gg - mu gn2 ;  g^T u = g2 , g^T w = gg
    const double qd = cg * cg * (g2 / alpha) + 2.0 * cg * cn * (-g2 - mu_u * gg) + cn * cn * (-gg - mu_u * gn2);
    const double gd = cg * g2 + cn * gg;
    const double model_change = -(0.5 * qd + gd);
    // ---------------- candidate state x_cur (+) step, parameter tolerance ---------------------------
    for (int i = t; i < D; i += NT) s.y[i] = s.sc[i] * (cg * s.gr[i] + cn * s.gn[i]) / s.dcs[i];
    __syncthreads();
    const double* stepc = s.y;
    double xn = 0, sn = 0;
    const int K = P.K;
    for (int k = t; k < 2 * K + 2; k += NT) {
        if (k < K) {
            const double* in = x + xo_pose(P, k); double* o = xc + xo_pose(P, k);
            if (P.pose_const && P.pose_const[k]) { for (int q = 0; q < 7; ++q) o[q] = in[q]; }
            else { pose_plus(in, stepc + col_pose(P, k), o); if (cam) for (int q = 0; q < 7; ++q) { xn += in[q] * in[q]; sn += (in[q] - o[q]) * (in[q] - o[q]); } }
        } else if (k < 2 * K) {
            const int kk = k - K;
            const double* in = x + xo_sb(P, kk); double* o = xc + xo_sb(P, kk);
            const bool cst = P.sb_const && P.sb_const[kk];
            for (int q = 0; q < 9; ++q) { const double d = cst ? 0.0 : stepc[col_sb(P, kk) + q]; o[q] = in[q] + d; if (!cst && cam) { xn += in[q] * in[q]; sn += d * d; } }
        } else if (k == 2 * K) {
            const double* in = x + xo_ex(P); double* o = xc + xo_ex(P);
            if (P.ex_const) { for (int q = 0; q < 7; ++q) o[q] = in[q]; }
            else { pose_plus(in, stepc + col_ex(P), o); if (cam) for (int q = 0; q < 7; ++q) { xn += in[q] * in[q]; sn += (in[q] - o[q]) * (in[q] - o[q]); } }
        } else {
            const double in = x[xo_td(P)];
            const double d = P.td_free ? stepc[col_td(P)] : 0.0;
            xc[xo_td(P)] = in + d;
            if (P.td_free && cam) { xn += in * in; sn += d * d; }
        }
    }
    for (int l = t; l < L; l += NT) {
        const double in = x[xo_lam(P) + l];
        double d = 0.0;
        if (sb.invp[l] != 0.0) d = P.Sl[l] * (cg * P.gradl[l] + cn * P.gnl[l]) / P.dl[l];
        xc[xo_lam(P) + l] = in + d;
        if (multi ? (sb.invp[l] != 0.0) : !(P.lm_const && P.lm_const[l])) { xn += in * in; sn += d * d; }
    }
    double dummy2 = 0;
    bsum3(xn, sn, dummy2, s);
    STAMP(6);
    if (t == 0) {
        Ctl& c = s.c;
        c.iter++;
        if (c.iter <= 64) { c.radius_trace[c.iter - 1] = c.radius; c.cost_trace[c.iter - 1] = c.cost_cur; }
        c.dogleg_norm = dnorm;
        c.model_change = model_change;
        if (!(model_change > 0.0)) {
            // invalid step (trust_region_minimizer.cc HandleInvalidStep): mu *= 10, same linearisation, new pivots
            c.invalid_run++;
            if (c.invalid_run >= 5) { c.done = 1; c.term = 6; c.status = -4; }
            c.mu *= 10.0; c.reuse = 0; c.resweep = 1;
        } else {
            c.invalid_run = 0;
            if (!multi) { if (sqrt(sn) <= O.parameter_tolerance * (sqrt(xn) + O.parameter_tolerance)) { c.done = 1; c.term = 3; } }
            else { double* tail = P.arstage + (size_t)D * D + 3 * D + 1; tail[0] = xn; tail[1] = sn; }   // judged after the next all-reduce
        }
    }
    __syncthreads();
    if (s.c.resweep && !s.c.done) { for (int i = t; i < P.NS; i += NT) xc[i] = x[i]; }
    STAMP(7);
    if (t == 0) *P.ctl = s.c;
}
