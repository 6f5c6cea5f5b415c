// The trust-region step kernel: ONE workgroup runs, per iteration and without any host round trip,
// what ceres' TrustRegionMinimizer + DoglegStrategy + the dense Cholesky of DENSE_SCHUR do around
// the sweep (estimator.cpp:1400-1414; algorithm restated in SURVEY.md Appendix B):
//   judge the candidate just swept (function tolerance, relative decrease, radius update) ->
//   on a new linearisation: Jacobi/dogleg scaling, gradient, Cauchy point, reduced Cholesky solve,
//   landmark back-substitution -> traditional dogleg blend -> model decrease -> candidate state.
// All trust-region state lives in `Ctl` in device memory; `done` makes later launches no-ops.
#pragma once
#include "vil_dev.hpp"
#include "vil_factors.hpp"

namespace vd {

struct StepShared {
    Ctl c;
    double red[32];
    double y[512];
    double col[512];
    double dg[512];
    int need, was_first, ok;
};

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
__device__ inline double quad_form(const DevP& P, const SysBuf& sb, const double* vc, const double* vl, StepShared& s) {
    const int D = P.D, L = P.L, t = threadIdx.x, NT = blockDim.x;
    double part = 0;
    for (int i = t; i < D; i += NT) {
        double row = 0;
        for (int k = 0; k < D; ++k) row += sb.S[(size_t)k * D + i] * vc[k];   // symmetric: column read is coalesced
        part += vc[i] * row;
    }
    for (int l = t; l < L; l += NT) {
        const double ip = sb.invp[l];
        if (ip == 0.0) continue;
        const double ev = lm_dot(P, sb, l, vc);
        part += ip * ev * ev + 2.0 * vl[l] * ev + sb.hll[l] * vl[l] * vl[l];
    }
    return bsum(part, s);
}

__device__ inline void pose_plus(const double* in, const double* d, double* o) {
    for (int k = 0; k < 3; ++k) o[k] = in[k] + d[k];
    Q4 q = qmul(qload(in + 3), Q4{1.0, 0.5 * d[3], 0.5 * d[4], 0.5 * d[5]});
    const double n = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    o[3] = q.x * n; o[4] = q.y * n; o[5] = q.z * n; o[6] = q.w * n;
}

}  // namespace vd

__global__ __launch_bounds__(VIL_STEP_THREADS) void k_step(DevP P, SolveOpts O) {
    using namespace vd;
    __shared__ StepShared s;
    const int t = threadIdx.x, NT = blockDim.x;
    const int D = P.D, L = P.L;
    if (t == 0) { s.c = *P.ctl; s.need = 0; s.was_first = 0; s.ok = 1; }
    __syncthreads();
    if (s.c.done) return;
    // ---------------- judge the candidate that the sweep just linearised -------------------------
    if (t == 0) {
        Ctl& c = s.c;
        const int cand = 1 - c.cur;
        const double cand_cost = *P.sys[cand].cost;
        c.cand_cost = cand_cost;
        if (c.first || c.resweep) {
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

    if (s.need) {
        // mirror the upper triangle written by the sweep
        for (int e = t; e < D * D; e += NT) { const int i = e / D, j = e % D; if (i < j) sb.S[(size_t)j * D + i] = sb.S[e]; }
        // gradient tolerance on the new linearisation
        double gm = 0;
        for (int i = t; i < D; i += NT) gm = fmax(gm, fabs(sb.bc[i]));
        for (int l = t; l < L; l += NT) if (sb.invp[l] != 0.0) gm = fmax(gm, fabs(sb.bl[l]));
        gm = bmax(gm, s);
        if (gm <= O.gradient_tolerance) { if (t == 0) { s.c.done = 1; s.c.term = 2; *P.ctl = s.c; } return; }
        // Jacobi scaling from the first Jacobian, dogleg diagonal, gradient in dogleg space
        if (s.was_first) for (int i = t; i < D; i += NT) P.Sc[i] = O.jacobi_scaling ? 1.0 / (1.0 + sqrt(sb.diag[i])) : 1.0;
        __syncthreads();
        double g2 = 0;
        for (int i = t; i < D; i += NT) {
            const double Sc = P.Sc[i];
            const double d = sqrt(fmin(fmax(Sc * Sc * sb.diag[i], 1e-6), 1e32));
            const double g = Sc * sb.bc[i] / d;
            P.dc[i] = d; P.gradc[i] = g; P.tmpc[i] = Sc * g / d; g2 += g * g;
        }
        for (int l = t; l < L; l += NT) {
            const double Sl = P.Sl[l];
            const double d = sqrt(fmin(fmax(Sl * Sl * sb.hll[l], 1e-6), 1e32));
            const double g = sb.invp[l] != 0.0 ? Sl * sb.bl[l] / d : 0.0;
            P.dl[l] = d; P.gradl[l] = g; P.tmpl[l] = Sl * g / d; g2 += g * g;
        }
        g2 = bsum(g2, s);
        const double q = quad_form(P, sb, P.tmpc, P.tmpl, s);
        // reduced system  M = Sc S' Sc + mu dc^2 ,  rhs = Sc gred   (M lower triangle used)
        const double mu = s.c.mu;
        for (int e = t; e < D * D; e += NT) {
            const int i = e / D, j = e % D;
            double v = P.Sc[i] * sb.S[e] * P.Sc[j];
            if (i == j) v += mu * P.dc[i] * P.dc[i];
            P.M[e] = v;
        }
        for (int i = t; i < D; i += NT) s.y[i] = P.Sc[i] * sb.gred[i];
        __syncthreads();
        // ---- dense Cholesky (right-looking), forward substitution fused as an extra row -----------
        bool fail = false;
        for (int j = 0; j < D; ++j) {
            const double piv = P.M[(size_t)j * D + j];
            if (!(piv > 0.0) || !isfinite(piv)) { fail = true; break; }
            const double inv = 1.0 / sqrt(piv);
            __syncthreads();
            for (int i = j + 1 + t; i < D; i += NT) { const double v = P.M[(size_t)i * D + j] * inv; P.M[(size_t)i * D + j] = v; s.col[i] = v; }
            if (t == 0) { s.dg[j] = 1.0 / inv; s.y[j] *= inv; }
            __syncthreads();
            const int r = D - 1 - j;
            const double yj = s.y[j];
            for (int e = t; e < r * r; e += NT) {
                const int i = j + 1 + e / r, k = j + 1 + e % r;
                if (k <= i) P.M[(size_t)i * D + k] -= s.col[i] * s.col[k];
            }
            for (int k = j + 1 + t; k < D; k += NT) s.y[k] -= yj * s.col[k];
            __syncthreads();
        }
        if (fail) {
            // dogleg_strategy.cc: mu *= 10 and retry; the Schur pivots depend on mu, so re-sweep at x_cur
            if (t == 0) {
                Ctl& c = s.c;
                c.mu *= 10.0;
                if (!(c.mu < O.max_mu)) { c.iter++; c.invalid_run++; c.reuse = 0; if (c.invalid_run >= 5) { c.done = 1; c.term = 6; c.status = -4; } }
                c.resweep = 1;
            }
            for (int i = t; i < P.NS; i += NT) xc[i] = x[i];
            SysBuf z = P.sys[1 - cur];
            for (int e = t; e < D * D; e += NT) z.S[e] = 0.0;
            for (int i = t; i < D; i += NT) { z.gred[i] = 0; z.bc[i] = 0; z.diag[i] = 0; }
            if (t == 0) z.cost[0] = 0.0;
            __syncthreads();
            if (t == 0) *P.ctl = s.c;
            return;
        }
        // back substitution L^T x = y
        for (int j = D - 1; j >= 0; --j) {
            __syncthreads();
            const double xj = s.y[j] / s.dg[j];
            for (int k = t; k < j; k += NT) s.y[k] -= P.M[(size_t)j * D + k] * xj;
            __syncthreads();
            if (t == 0) s.y[j] = xj;
        }
        __syncthreads();
        // gauss-newton step in dogleg space (camera part), landmark back-substitution
        for (int i = t; i < D; i += NT) { const double xi = s.y[i]; P.gnc[i] = -xi * P.dc[i]; P.tmpc[i] = P.Sc[i] * xi; }
        __syncthreads();
        for (int l = t; l < L; l += NT) {
            const double ip = sb.invp[l];
            double xl = 0.0;
            if (ip != 0.0) xl = (sb.bl[l] - lm_dot(P, sb, l, P.tmpc)) * ip / P.Sl[l];
            P.gnl[l] = -xl * P.dl[l];
        }
        if (t == 0) { s.c.alpha = g2 / q; s.c.mu = fmax(O.min_mu, 2.0 * s.c.mu / 10.0); }
        __syncthreads();
    }
    // ---------------- traditional dogleg in dogleg space -----------------------------------------
    double gn2 = 0, g2 = 0, gg = 0;
    for (int i = t; i < D; i += NT) { const double a = P.gnc[i], b = P.gradc[i]; gn2 += a * a; g2 += b * b; gg += a * b; }
    for (int l = t; l < L; l += NT) { const double a = P.gnl[l], b = P.gradl[l]; gn2 += a * a; g2 += b * b; gg += a * b; }
    gn2 = bsum(gn2, s); g2 = bsum(g2, s); gg = bsum(gg, s);
    const double radius = s.c.radius, alpha = s.c.alpha;
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
    double gd = 0;
    for (int i = t; i < D; i += NT) { const double st = P.Sc[i] * (cg * P.gradc[i] + cn * P.gnc[i]) / P.dc[i]; P.stepc[i] = st; gd += sb.bc[i] * st; }
    for (int l = t; l < L; l += NT) {
        double st = 0.0;
        if (sb.invp[l] != 0.0) { st = P.Sl[l] * (cg * P.gradl[l] + cn * P.gnl[l]) / P.dl[l]; gd += sb.bl[l] * st; }
        P.stepl[l] = st;
    }
    gd = bsum(gd, s);
    const double qd = quad_form(P, sb, P.stepc, P.stepl, s);
    const double model_change = -(0.5 * qd + gd);
    // ---------------- candidate state x_cur (+) step, parameter tolerance ---------------------------
    double xn = 0, sn = 0;
    const int K = P.K;
    for (int k = t; k < 2 * K + 2; k += NT) {
        if (k < K) {
            const double* in = x + xo_pose(P, k); double* o = xc + xo_pose(P, k);
            if (P.pose_const && P.pose_const[k]) { for (int q = 0; q < 7; ++q) o[q] = in[q]; }
            else { pose_plus(in, P.stepc + col_pose(P, k), o); for (int q = 0; q < 7; ++q) { xn += in[q] * in[q]; sn += (in[q] - o[q]) * (in[q] - o[q]); } }
        } else if (k < 2 * K) {
            const int kk = k - K;
            const double* in = x + xo_sb(P, kk); double* o = xc + xo_sb(P, kk);
            const bool cst = P.sb_const && P.sb_const[kk];
            for (int q = 0; q < 9; ++q) { const double d = cst ? 0.0 : P.stepc[col_sb(P, kk) + q]; o[q] = in[q] + d; if (!cst) { xn += in[q] * in[q]; sn += d * d; } }
        } else if (k == 2 * K) {
            const double* in = x + xo_ex(P); double* o = xc + xo_ex(P);
            if (P.ex_const) { for (int q = 0; q < 7; ++q) o[q] = in[q]; }
            else { pose_plus(in, P.stepc + col_ex(P), o); for (int q = 0; q < 7; ++q) { xn += in[q] * in[q]; sn += (in[q] - o[q]) * (in[q] - o[q]); } }
        } else {
            const double in = x[xo_td(P)];
            const double d = P.td_free ? P.stepc[col_td(P)] : 0.0;
            xc[xo_td(P)] = in + d;
            if (P.td_free) { xn += in * in; sn += d * d; }
        }
    }
    for (int l = t; l < L; l += NT) {
        const double in = x[xo_lam(P) + l];
        const bool fr = !(P.lm_const && P.lm_const[l]);
        const double d = fr ? P.stepl[l] : 0.0;
        xc[xo_lam(P) + l] = in + d;
        if (fr) { xn += in * in; sn += d * d; }
    }
    xn = bsum(xn, s); sn = bsum(sn, s);
    // zero the candidate system for the next sweep
    {
        SysBuf z = P.sys[1 - cur];
        for (int e = t; e < D * D; e += NT) z.S[e] = 0.0;
        for (int i = t; i < D; i += NT) { z.gred[i] = 0; z.bc[i] = 0; z.diag[i] = 0; }
        if (t == 0) z.cost[0] = 0.0;
    }
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
            if (sqrt(sn) <= O.parameter_tolerance * (sqrt(xn) + O.parameter_tolerance)) { c.done = 1; c.term = 3; }
        }
    }
    __syncthreads();
    if (s.c.resweep && !s.c.done) { for (int i = t; i < P.NS; i += NT) xc[i] = x[i]; }
    if (t == 0) *P.ctl = s.c;
}
