// The factor sweep: ONE launch evaluates every residual block of the window at the candidate state
// and reduces it into the Schur-complement normal equations of the candidate linearisation
// (what ceres' Evaluate + SchurEliminator do per trust-region iteration behind estimator.cpp:1414).
//
// Work-group roles by blockIdx (all 256 threads):
//   [imu]     one WG per IMU factor: lane 0 forms the raw 15x30 block, the WG whitens with the
//             pre-factored sqrt-information and contracts to a 30x30 H block
//   [visual]  one WG per chunk of <= 8 landmarks: thread-per-factor evaluation staged in LDS,
//             thread-per-landmark Schur pivots, wave-per-landmark block outer products
//   [plane]/[edge] one WG per <=256 pose-uniform LiDAR points: thread-per-point evaluation,
//             wave64 butterfly reduction of the 6x6 + 6 + cost, one atomic set per wave
//   [misc]    prior (n x n gemv on the pre-contracted J0^T J0), ICP and LPS AutoDiff factors
#pragma once
#include "vil_dev.hpp"
#include "vil_factors.hpp"

namespace vd {

__device__ __forceinline__ void S_add(const DevP& P, double* S, int i, int j, double v) {
    if (i > j) { int t = i; i = j; j = t; }
    atomic_add_f64(S + (size_t)i * P.D + j, v);
}

__device__ __forceinline__ double block_sum(double v, double* red /*>= 4 doubles*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}

// ---------------------------------------------------------------------------------------------
__device__ inline void sweep_imu(const DevP& P, const SolveOpts& O, int f, const double* x, SysBuf& sb, double* sm) {
    double* Jraw = sm;            // 450
    double* rr = sm + 450;        // 15
    double* UJ = sm + 480;        // 450
    double* Ur = sm + 930;        // 15
    const double* c = P.imu_c + (size_t)f * 287;
    if (c[16] > 10.0) return;     // estimator.cpp:1182
    const int i = P.imu_i[f], j = P.imu_j[f];
    const int t = threadIdx.x;
    if (t == 0) imu_raw(c, V3{P.G[0], P.G[1], P.G[2]}, x + xo_pose(P, i), x + xo_sb(P, i), x + xo_pose(P, j), x + xo_sb(P, j), rr, Jraw);
    __syncthreads();
    const double* U = P.imu_U + (size_t)f * 225;
    const bool ci = P.pose_const && P.pose_const[i], cj = P.pose_const && P.pose_const[j];
    const bool si = P.sb_const && P.sb_const[i], sj = P.sb_const && P.sb_const[j];
    for (int e = t; e < 465; e += blockDim.x) {
        if (e < 450) {
            const int row = e / 30, col = e % 30;
            const bool cst = col < 6 ? ci : (col < 15 ? si : (col < 21 ? cj : sj));
            double s = 0;
            if (!cst) for (int k = row; k < 15; ++k) s += U[row * 15 + k] * Jraw[k * 30 + col];
            UJ[e] = s;
        } else {
            const int row = e - 450;
            double s = 0;
            for (int k = row; k < 15; ++k) s += U[row * 15 + k] * rr[k];
            Ur[row] = s;
        }
    }
    __syncthreads();
    auto gcol = [&](int a) { return a < 6 ? col_pose(P, i) + a : (a < 15 ? col_sb(P, i) + a - 6 : (a < 21 ? col_pose(P, j) + a - 15 : col_sb(P, j) + a - 21)); };
    for (int e = t; e < 900 + 30; e += blockDim.x) {
        if (e < 900) {
            const int a = e / 30, b = e % 30;
            if (a > b) continue;
            double s = 0;
            for (int k = 0; k < 15; ++k) s += UJ[k * 30 + a] * UJ[k * 30 + b];
            if (s != 0.0) {
                S_add(P, sb.S, gcol(a), gcol(b), s);
                if (a == b) atomic_add_f64(sb.diag + gcol(a), s);
            }
        } else {
            const int a = e - 900;
            double s = 0;
            for (int k = 0; k < 15; ++k) s += UJ[k * 30 + a] * Ur[k];
            if (s != 0.0) { atomic_add_f64(sb.bc + gcol(a), s); atomic_add_f64(sb.gred + gcol(a), s); }
        }
    }
    if (t == 0) { double s = 0; for (int k = 0; k < 15; ++k) s += Ur[k] * Ur[k]; atomic_add_f64(sb.cost, 0.5 * s); }
}

// ---------------------------------------------------------------------------------------------
// LDS per factor: [Ji 12 | Jj 12 | Jex 12 | Jt 2 | Jl 2 | r 2 | eO 6] = 48 doubles
#define VF_STRIDE 49   // odd stride: conflict-free column access
__device__ inline void sweep_visual(const DevP& P, const SolveOpts& O, const Ctl& ctl, int chunk, const double* x, SysBuf& sb, double* sm) {
    const int l0 = P.vchunk[2 * chunk], l1 = P.vchunk[2 * chunk + 1];
    const int f0 = P.lm_start[l0], f1 = P.lm_start[l1];
    const int nf = f1 - f0, nl = l1 - l0;
    const int t = threadIdx.x;
    double* Jf = sm;                                   // VIL_VCHUNK_F x VF_STRIDE
    double* lmr = sm + VIL_VCHUNK_F * VF_STRIDE;       // VIL_VCHUNK_LM x 16: invp, eA[13]
    double* red = lmr + VIL_VCHUNK_LM * 16;
    const bool exc = P.ex_const != 0, tdc = !P.td_free;
    double cost = 0.0;
    if (t < nf) {
        const int f = f0 + t;
        double c[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) c[k] = P.vis_c[(size_t)k * P.vis_stride + f];
        const int i = P.vis_i[f], j = P.vis_j[f], l = P.vis_l[f];
        const double* pi = x + xo_pose(P, i); const double* pj = x + xo_pose(P, j); const double* ex = x + xo_ex(P);
        VisJ o;
        visual_eval(c, quatR(pi + 3), V3{pi[0], pi[1], pi[2]}, quatR(pj + 3), V3{pj[0], pj[1], pj[2]}, quatR(ex + 3), V3{ex[0], ex[1], ex[2]},
                    x[xo_lam(P) + l], x[xo_td(P)], P.sqrt_info, P.k_tr, P.use_td, o);
        double rho, rho1;
        loss_eval(O.visual_loss, O.visual_loss_scale, o.r[0] * o.r[0] + o.r[1] * o.r[1], rho, rho1);
        cost = 0.5 * rho;
        const double sr = sqrt(rho1);
        const bool ci = P.pose_const && P.pose_const[i], cj = P.pose_const && P.pose_const[j], cl = P.lm_const && P.lm_const[l];
        double* w = Jf + t * VF_STRIDE;
        for (int k = 0; k < 12; ++k) { w[k] = ci ? 0.0 : sr * o.Ji[k]; w[12 + k] = cj ? 0.0 : sr * o.Jj[k]; w[24 + k] = exc ? 0.0 : sr * o.Jex[k]; }
        w[36] = tdc ? 0.0 : sr * o.Jt[0]; w[37] = tdc ? 0.0 : sr * o.Jt[1];
        w[38] = cl ? 0.0 : sr * o.Jl[0]; w[39] = cl ? 0.0 : sr * o.Jl[1];
        w[40] = sr * o.r[0]; w[41] = sr * o.r[1];
        // observer-pose pieces that need no landmark-level sum
        for (int k = 0; k < 6; ++k) {
            const double j0 = w[12 + k], j1 = w[18 + k];
            const double eo = j0 * w[38] + j1 * w[39];
            w[42 + k] = eo;
            sb.eO[(size_t)f * 6 + k] = eo;
            if (!cj) {
                const double g = j0 * w[40] + j1 * w[41];
                atomic_add_f64(sb.bc + col_pose(P, j) + k, g);
                atomic_add_f64(sb.gred + col_pose(P, j) + k, g);
                atomic_add_f64(sb.diag + col_pose(P, j) + k, j0 * j0 + j1 * j1);
            }
        }
    }
    cost = block_sum(cost, red);
    if (t == 0) atomic_add_f64(sb.cost, cost);
    __syncthreads();
    // ---- per landmark: pivots, e on the shared groups, gradients ------------------------------
    if (t < nl) {
        const int l = l0 + t;
        const int fs = P.lm_start[l] - f0, fe = P.lm_start[l + 1] - f0;
        const int a = P.vis_i[P.lm_start[l]];
        double h = 0, b = 0, e[13], g[13], dg[13];
        for (int k = 0; k < 13; ++k) { e[k] = 0; g[k] = 0; dg[k] = 0; }
        for (int q = fs; q < fe; ++q) {
            const double* w = Jf + q * VF_STRIDE;
            const double l0_ = w[38], l1_ = w[39], r0 = w[40], r1 = w[41];
            h += l0_ * l0_ + l1_ * l1_; b += l0_ * r0 + l1_ * r1;
            for (int k = 0; k < 6; ++k) {
                e[k] += w[k] * l0_ + w[6 + k] * l1_; g[k] += w[k] * r0 + w[6 + k] * r1; dg[k] += w[k] * w[k] + w[6 + k] * w[6 + k];
                e[6 + k] += w[24 + k] * l0_ + w[30 + k] * l1_; g[6 + k] += w[24 + k] * r0 + w[30 + k] * r1; dg[6 + k] += w[24 + k] * w[24 + k] + w[30 + k] * w[30 + k];
            }
            e[12] += w[36] * l0_ + w[37] * l1_; g[12] += w[36] * r0 + w[37] * r1; dg[12] += w[36] * w[36] + w[37] * w[37];
        }
        const bool cl = P.lm_const && P.lm_const[l];
        double Sl = 1.0;
        if (ctl.first) { Sl = (O.jacobi_scaling && !ctl.lin_mode) ? 1.0 / (1.0 + sqrt(h)) : 1.0; P.Sl[l] = Sl; }
        else Sl = P.Sl[l];
        double dl2 = Sl * Sl * h; dl2 = fmin(fmax(dl2, 1e-6), 1e32);
        const double p = ctl.lin_mode ? h : h + ctl.mu * dl2 / (Sl * Sl);
        const double invp = (cl || !(p > 0.0)) ? 0.0 : 1.0 / p;
        sb.hll[l] = h; sb.bl[l] = b; sb.invp[l] = invp;
        double* lr = lmr + t * 16;
        lr[0] = invp;
        const double ib = invp * b;
        for (int k = 0; k < 13; ++k) {
            lr[1 + k] = e[k]; sb.eA[(size_t)l * 13 + k] = e[k];
            const int col = k < 6 ? col_pose(P, a) + k : (k < 12 ? col_ex(P) + k - 6 : col_td(P));
            if (g[k] != 0.0 || dg[k] != 0.0) {
                atomic_add_f64(sb.bc + col, g[k]);
                atomic_add_f64(sb.gred + col, g[k] - ib * e[k]);
                atomic_add_f64(sb.diag + col, dg[k]);
            }
        }
        for (int q = fs; q < fe; ++q) {
            const double* w = Jf + q * VF_STRIDE;
            const int j = P.vis_j[f0 + q];
            for (int k = 0; k < 6; ++k) if (w[42 + k] != 0.0) atomic_add_f64(sb.gred + col_pose(P, j) + k, -ib * w[42 + k]);
        }
    }
    __syncthreads();
    // ---- wave per landmark: S += sum_f Jc^T Jc - invp e e^T, by (group, group) blocks -----------
    const int wave = t >> 6, lane = t & 63;
    for (int tl = wave; tl < nl; tl += (int)(blockDim.x >> 6)) {
        const int l = l0 + tl;
        const int fs = P.lm_start[l] - f0, fe = P.lm_start[l + 1] - f0;
        const int m = fe - fs, ng = 3 + m, np = ng * (ng + 1) / 2;
        const int a = P.vis_i[P.lm_start[l]];
        const double* lr = lmr + tl * 16;
        const double invp = lr[0];
        for (int p = lane; p < np; p += 64) {
            int g1 = 0, rem = p;
            while (rem >= ng - g1) { rem -= ng - g1; ++g1; }
            const int g2 = g1 + rem;
            // group descriptors: offset in the factor record, #cols, global column, e pointer
            auto goff = [&](int g) { return g == 0 ? 0 : (g == 1 ? 24 : (g == 2 ? 36 : 12)); };
            auto gn = [&](int g) { return g == 2 ? 1 : 6; };
            auto gcol = [&](int g) { return g == 0 ? col_pose(P, a) : (g == 1 ? col_ex(P) : (g == 2 ? col_td(P) : col_pose(P, P.vis_j[f0 + fs + g - 3]))); };
            const int n1 = gn(g1), n2 = gn(g2), o1 = goff(g1), o2 = goff(g2), c1 = gcol(g1), c2 = gcol(g2);
            const double* e1 = g1 < 3 ? lr + 1 + (g1 == 0 ? 0 : (g1 == 1 ? 6 : 12)) : Jf + (fs + g1 - 3) * VF_STRIDE + 42;
            const double* e2 = g2 < 3 ? lr + 1 + (g2 == 0 ? 0 : (g2 == 1 ? 6 : 12)) : Jf + (fs + g2 - 3) * VF_STRIDE + 42;
            // factors common to both groups
            int qa, qb;
            if (g1 >= 3 && g2 >= 3) { if (g1 == g2) { qa = fs + g1 - 3; qb = qa + 1; } else { qa = 0; qb = 0; } }
            else if (g2 >= 3) { qa = fs + g2 - 3; qb = qa + 1; }
            else if (g1 >= 3) { qa = fs + g1 - 3; qb = qa + 1; }
            else { qa = fs; qb = fe; }
            const int s1 = n1 == 1 ? 1 : 6, s2 = n2 == 1 ? 1 : 6;   // row stride inside a 2 x n block
            for (int r = 0; r < n1; ++r) for (int c = (g1 == g2 ? r : 0); c < n2; ++c) {
                double s = 0;
                for (int q = qa; q < qb; ++q) {
                    const double* w = Jf + q * VF_STRIDE;
                    s += w[o1 + r] * w[o2 + c] + w[o1 + s1 + r] * w[o2 + s2 + c];
                }
                s -= invp * e1[r] * e2[c];
                if (s != 0.0) S_add(P, sb.S, c1 + r, c2 + c, s);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int NR>
__device__ inline void sweep_lidar(const DevP& P, const SolveOpts& O, int chunk, const double* x, SysBuf& sb) {
    const int* ch = (NR == 1 ? P.pchunk : P.echunk) + 3 * chunk;
    const int start = ch[0], cnt = ch[1], k = ch[2];
    const int t = threadIdx.x;
    const double* pose = x + xo_pose(P, k);
    const bool cst = P.pose_const && P.pose_const[k];
    double acc[28];
#pragma unroll
    for (int q = 0; q < 28; ++q) acc[q] = 0.0;
    if (t < cnt) {
        const int f = start + t;
        const M3 R = quatR(pose + 3), Rbl = loadM3(P.Rbl);
        const V3 Pk{pose[0], pose[1], pose[2]}, tbl{P.tbl[0], P.tbl[1], P.tbl[2]};
        double r[NR], J[NR * 6];
        if (NR == 1) {
            const double* c = P.pl_c; const int s = P.pl_stride;
            plane_eval(V3{c[f], c[s + f], c[2 * s + f]}, V3{c[3 * s + f], c[4 * s + f], c[5 * s + f]}, c[6 * s + f], Rbl, tbl, R, Pk, r[0], J);
        } else {
            const double* c = P.ed_c; const int s = P.ed_stride;
            edge_eval(V3{c[f], c[s + f], c[2 * s + f]}, V3{c[3 * s + f], c[4 * s + f], c[5 * s + f]}, V3{c[6 * s + f], c[7 * s + f], c[8 * s + f]}, Rbl, tbl, R, Pk, r, J);
        }
        double sq = 0;
#pragma unroll
        for (int q = 0; q < NR; ++q) sq += r[q] * r[q];
        double rho, rho1;
        loss_eval(O.lidar_loss, O.lidar_loss_scale, sq, rho, rho1);
        acc[27] = 0.5 * rho;
        // rho1 multiplies J^T J and J^T r (sqrt(rho1) on each factor of the product)
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = a; b < 6; ++b) {
                double s = 0;
#pragma unroll
                for (int q = 0; q < NR; ++q) s += J[q * 6 + a] * J[q * 6 + b];
                acc[idx++] = rho1 * s;
            }
            double g = 0;
#pragma unroll
            for (int q = 0; q < NR; ++q) g += J[q * 6 + a] * r[q];
            acc[21 + a] = rho1 * g;
        }
    }
#pragma unroll
    for (int q = 0; q < 28; ++q) acc[q] = wave_sum(acc[q]);
    if ((t & 63) == 0 && (t & ~63) < cnt) {
        atomic_add_f64(sb.cost, acc[27]);
        if (!cst) {
            const int c0 = col_pose(P, k);
            int idx = 0;
            for (int a = 0; a < 6; ++a) {
                for (int b = a; b < 6; ++b) { atomic_add_f64(sb.S + (size_t)(c0 + a) * P.D + c0 + b, acc[idx]); if (a == b) atomic_add_f64(sb.diag + c0 + a, acc[idx]); ++idx; }
                atomic_add_f64(sb.bc + c0 + a, acc[21 + a]);
                atomic_add_f64(sb.gred + c0 + a, acc[21 + a]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
__device__ inline const double* prior_block_ptr(const DevP& P, const double* x, int b) {
    const int kind = P.pblk_kind[b], idx = P.pblk_index[b];
    return kind == 0 ? x + xo_pose(P, idx) : (kind == 1 ? x + xo_sb(P, idx) : (kind == 2 ? x + xo_ex(P) : x + xo_td(P)));
}

__device__ inline void sweep_misc(const DevP& P, const SolveOpts& O, const double* x, SysBuf& sb, double* sm) {
    const int t = threadIdx.x;
    // ---- prior: r = r0 + J0 dx ;  J0^T J0 = pH, J0^T r0 = pg0, r0^T r0 = pc0 are pre-contracted ------------
    if (P.pn > 0) {
        const int n = P.pn;
        double* dx = sm;           // n
        double* red = sm + 512;
        if (t < P.pnblk) {
            const int kind = P.pblk_kind[t];
            const int gs = kind == 0 || kind == 2 ? 7 : (kind == 1 ? 9 : 1);
            double d[9];
            prior_block_dx(gs, prior_block_ptr(P, x, t), P.px0 + P.pblk_xoff[t], d);
            const int ls = gs == 7 ? 6 : gs;
            for (int k = 0; k < ls; ++k) dx[P.pblk_col[t] + k] = d[k];
        }
        __syncthreads();
        double part = 0;
        for (int i = t; i < n; i += blockDim.x) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += P.pH[(size_t)k * n + i] * dx[k];
            const double g = P.pg0[i] + s;
            part += dx[i] * (P.pg0[i] + g);
            const int col = P.pmap[i];
            if (col >= 0) { atomic_add_f64(sb.bc + col, g); atomic_add_f64(sb.gred + col, g); }
        }
        part = block_sum(part, red);
        if (t == 0) atomic_add_f64(sb.cost, 0.5 * (P.pc0[0] + part));
        for (int e = t; e < n * n; e += blockDim.x) {
            const int i = e / n, k = e % n;
            if (i > k) continue;
            const int ci = P.pmap[i], ck = P.pmap[k];
            if (ci < 0 || ck < 0) continue;
            const double v = P.pH[e];
            S_add(P, sb.S, ci, ck, v);
            if (i == k) atomic_add_f64(sb.diag + ci, v);
        }
        __syncthreads();
    }
    // ---- ICP (4 pose blocks) and LPS (2 pose blocks): thread per (factor, block) ----------------------------
    double* Jb = sm;              // up to 12 factors x 4 blocks x 21
    double* rb = sm + 12 * 84;    // 12 x 3
    const int n_rel = P.n_icp + P.n_lps;
    if (n_rel == 0) return;
    __syncthreads();
    if (t < 4 * n_rel) {
        const int f = t >> 2, b = t & 3;
        double r3[3], J21[21];
        if (f < P.n_icp) {
            const int* id = P.icp_ids + 4 * f;
            icp_eval(P.icp_c + (size_t)f * 10, x + xo_pose(P, id[0]), x + xo_pose(P, id[1]), x + xo_pose(P, id[2]), x + xo_pose(P, id[3]), b, r3, J21);
            if (!O.autodiff_quirk) tangent_fix(x + xo_pose(P, id[b]), J21);
            for (int k = 0; k < 21; ++k) Jb[(f * 4 + b) * 21 + k] = J21[k];
            if (b == 0) for (int k = 0; k < 3; ++k) rb[f * 3 + k] = r3[k];
        } else if (b < 2) {
            const int g = f - P.n_icp;
            const int* id = P.lps_ids + 2 * g;
            lps_eval(P.lps_c + (size_t)g * 7, x + xo_pose(P, id[0]), x + xo_pose(P, id[1]), b, r3, J21);
            if (!O.autodiff_quirk) tangent_fix(x + xo_pose(P, id[b]), J21);
            for (int k = 0; k < 21; ++k) Jb[(f * 4 + b) * 21 + k] = J21[k];
            if (b == 0) for (int k = 0; k < 3; ++k) rb[f * 3 + k] = r3[k];
        }
    }
    __syncthreads();
    // per factor: nb blocks x 6 local columns; entries (a,b) of the (6 nb)^2 block + gradient
    for (int f = 0; f < n_rel; ++f) {
        const bool icp = f < P.n_icp;
        const int nb = icp ? 4 : 2, nc = 6 * nb;
        const int* id = icp ? P.icp_ids + 4 * f : P.lps_ids + 2 * (f - P.n_icp);
        const double* r = rb + f * 3;
        double rho, rho1;
        loss_eval(O.rel_loss, O.rel_loss_scale, r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho, rho1);
        if (t == 0) atomic_add_f64(sb.cost, 0.5 * rho);
        for (int e = t; e < nc * nc + nc; e += blockDim.x) {
            if (e < nc * nc) {
                const int a = e / nc, b = e % nc;
                const int ba = a / 6, bb = b / 6;
                const int ca = col_pose(P, id[ba]) + a % 6, cb = col_pose(P, id[bb]) + b % 6;
                if (ca > cb) continue;
                if (ca == cb && a > b) continue;
                if ((P.pose_const && (P.pose_const[id[ba]] || P.pose_const[id[bb]]))) continue;
                const double* Ja = Jb + (f * 4 + ba) * 21 + a % 6;
                const double* Jc = Jb + (f * 4 + bb) * 21 + b % 6;
                double s = rho1 * (Ja[0] * Jc[0] + Ja[7] * Jc[7] + Ja[14] * Jc[14]);
                if (ca == cb && ba != bb) s *= 2.0;   // duplicated pose id inside one factor: both cross terms land on one entry
                atomic_add_f64(sb.S + (size_t)ca * P.D + cb, s);
                if (ca == cb) atomic_add_f64(sb.diag + ca, s);
            } else {
                const int a = e - nc * nc, ba = a / 6;
                if (P.pose_const && P.pose_const[id[ba]]) continue;
                const double* Ja = Jb + (f * 4 + ba) * 21 + a % 6;
                const double g = rho1 * (Ja[0] * r[0] + Ja[7] * r[1] + Ja[14] * r[2]);
                const int ca = col_pose(P, id[ba]) + a % 6;
                atomic_add_f64(sb.bc + ca, g); atomic_add_f64(sb.gred + ca, g);
            }
        }
    }
}

}  // namespace vd

// grid = n_imu + n_vchunk + n_pchunk + n_echunk + 1 workgroups of 256 threads
__global__ __launch_bounds__(VIL_THREADS) void k_sweep(DevP P, SolveOpts O) {
    extern __shared__ double sm[];
    const Ctl ctl = *P.ctl;
    if (ctl.done) return;
    const int cand = 1 - ctl.cur;
    const double* x = P.x[cand];
    SysBuf sb = P.sys[cand];
    int b = blockIdx.x;
    if (b < P.n_imu) { vd::sweep_imu(P, O, b, x, sb, sm); return; }
    b -= P.n_imu;
    if (b < P.n_vchunk) { vd::sweep_visual(P, O, ctl, b, x, sb, sm); return; }
    b -= P.n_vchunk;
    if (b < P.n_pchunk) { vd::sweep_lidar<1>(P, O, b, x, sb); return; }
    b -= P.n_pchunk;
    if (b < P.n_echunk) { vd::sweep_lidar<3>(P, O, b, x, sb); return; }
    vd::sweep_misc(P, O, x, sb, sm);
}
