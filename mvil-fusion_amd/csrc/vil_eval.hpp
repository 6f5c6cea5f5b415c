// Evaluate()-compatible kernels: raw (no loss) residuals and ROW-MAJOR GLOBAL-size Jacobian blocks,
// exactly what ceres::CostFunction::Evaluate(parameters, residuals, jacobians) writes
// (imu_factor.h:19, projection_td_factor.cpp:34, marginalization_factor.cpp:352, lidar_backend.h:45/107)
// for a whole factor class at once -- the materialised-Jacobian sweep.  Also the one-time set-up
// kernel (IMU sqrt-information factorisation, prior contraction J0^T J0).
#pragma once
#include "vil_dev.hpp"
#include "vil_factors.hpp"

// r: 2/factor, J: 46/factor = [Ji 2x7 | Jj 2x7 | Jex 2x7 | Jl 2 | Jt 2]
__global__ void k_eval_visual(DevP P, const double* x, double* r, double* J) {
    using namespace vd;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= P.n_vis) return;
    double c[14];
    const int fs = P.vfinv[f];                           // the tables are stored in the sweep's sorted order; the output keeps the caller's factor order
#pragma unroll
    for (int k = 0; k < 14; ++k) c[k] = P.vis_c[(size_t)k * P.vis_stride + fs];
    const int i = P.vis_i[fs], j = P.vis_j[fs], l = P.vis_l[fs];
    const double* pi = x + xo_pose(P, i); const double* pj = x + xo_pose(P, j); const double* ex = x + xo_ex(P);
    VisJ o;
    visual_eval(c, quatR(pi + 3), V3{pi[0], pi[1], pi[2]}, quatR(pj + 3), V3{pj[0], pj[1], pj[2]}, quatR(ex + 3), V3{ex[0], ex[1], ex[2]},
                x[xo_lam(P) + l], x[xo_td(P)], P.sqrt_info, P.k_tr, P.use_td, o);
    r[2 * f] = o.r[0]; r[2 * f + 1] = o.r[1];
    if (!J) return;
    double* w = J + (size_t)f * 46;
    for (int row = 0; row < 2; ++row) {
        for (int k = 0; k < 6; ++k) { w[row * 7 + k] = o.Ji[row * 6 + k]; w[14 + row * 7 + k] = o.Jj[row * 6 + k]; w[28 + row * 7 + k] = o.Jex[row * 6 + k]; }
        w[row * 7 + 6] = 0.0; w[14 + row * 7 + 6] = 0.0; w[28 + row * 7 + 6] = 0.0;
    }
    w[42] = o.Jl[0]; w[43] = o.Jl[1]; w[44] = o.Jt[0]; w[45] = o.Jt[1];
}

template <int NR>
__global__ void k_eval_lidar(DevP P, const double* x, double* r, double* J) {
    using namespace vd;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = NR == 1 ? P.n_plane : P.n_edge;
    if (f >= n) return;
    // pose id: recover from the chunk table (points are pose-sorted); binary search over chunks
    const int* ch = NR == 1 ? P.pchunk : P.echunk;
    int lo = 0, hi = (NR == 1 ? P.n_pchunk : P.n_echunk) - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (ch[3 * mid] <= f) lo = mid; else hi = mid - 1; }
    const int k = ch[3 * lo + 2];
    const double* pose = x + xo_pose(P, k);
    const M3 R = quatR(pose + 3), Rbl = loadM3(P.Rbl);
    const V3 Pk{pose[0], pose[1], pose[2]}, tbl{P.tbl[0], P.tbl[1], P.tbl[2]};
    double rr[NR], JJ[NR * 6];
    if (NR == 1) { const double* c = P.pl_c; const int s = P.pl_stride; plane_eval(V3{c[f], c[s + f], c[2 * s + f]}, V3{c[3 * s + f], c[4 * s + f], c[5 * s + f]}, c[6 * s + f], Rbl, tbl, R, Pk, rr[0], JJ); }
    else { const double* c = P.ed_c; const int s = P.ed_stride; edge_eval(V3{c[f], c[s + f], c[2 * s + f]}, V3{c[3 * s + f], c[4 * s + f], c[5 * s + f]}, V3{c[6 * s + f], c[7 * s + f], c[8 * s + f]}, Rbl, tbl, R, Pk, rr, JJ); }
    for (int q = 0; q < NR; ++q) {
        r[(size_t)f * NR + q] = rr[q];
        if (J) { for (int c2 = 0; c2 < 6; ++c2) J[((size_t)f * NR + q) * 7 + c2] = JJ[q * 6 + c2]; J[((size_t)f * NR + q) * 7 + 6] = 0.0; }
    }
}

// all four lidarFactor.hpp functors at one pose (vil_eval_lidar_functors): thread = point, constants point-major
struct LidarFunctorArgs { double Rbl[9], tbl[3], pose[7]; };
__global__ void k_eval_lidar_functors(int kind, int n, const double* __restrict__ c, LidarFunctorArgs A, double* __restrict__ r, double* __restrict__ J) {
    using namespace vd;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n) return;
    const M3 R = quatR(A.pose + 3), Rbl = loadM3(A.Rbl);
    const V3 Pk{A.pose[0], A.pose[1], A.pose[2]}, tbl{A.tbl[0], A.tbl[1], A.tbl[2]};
    const int nc = kind == 0 ? 9 : (kind == 1 ? 12 : (kind == 2 ? 7 : 6)), nr = (kind == 0 || kind == 3) ? 3 : 1;
    const double* q = c + (size_t)f * nc;
    const V3 cp{q[0], q[1], q[2]};
    double rr[3] = {0, 0, 0}, JJ[18];
    for (int e = 0; e < 18; ++e) JJ[e] = 0.0;
    if (kind == 0) edge_eval(cp, V3{q[3], q[4], q[5]}, V3{q[6], q[7], q[8]}, Rbl, tbl, R, Pk, rr, JJ);
    else if (kind == 1) {                               // LidarPlaneFactor: ljm_norm = ((j - l) x (j - m)).normalized(), residual (lp - j) . ljm_norm
        const V3 pj{q[3], q[4], q[5]}, pl_{q[6], q[7], q[8]}, pm{q[9], q[10], q[11]};
        V3 nv = cross(pj - pl_, pj - pm);
        const double inv = 1.0 / sqrt(dot(nv, nv));
        nv = inv * nv;
        plane_eval(cp, nv, -dot(nv, pj), Rbl, tbl, R, Pk, rr[0], JJ);
    } else if (kind == 2) plane_eval(cp, V3{q[3], q[4], q[5]}, q[6], Rbl, tbl, R, Pk, rr[0], JJ);
    else {                                              // LidarDistanceFactor: r = p_w - closed ; J = [I | -R [p_b]x]
        const V3 pb = mul(Rbl, cp) + tbl, pw = mul(R, pb) + Pk;
        rr[0] = pw.x - q[3]; rr[1] = pw.y - q[4]; rr[2] = pw.z - q[5];
        for (int i = 0; i < 3; ++i) {
            const V3 ei{i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0};
            const V3 jr = cross(pb, mulT(R, ei));         // row i of -R [p_b]x
            JJ[6 * i + i] = 1.0; JJ[6 * i + 3] = jr.x; JJ[6 * i + 4] = jr.y; JJ[6 * i + 5] = jr.z;
        }
    }
    for (int i = 0; i < nr; ++i) {
        r[(size_t)f * nr + i] = rr[i];
        if (J) { for (int c2 = 0; c2 < 6; ++c2) J[((size_t)f * nr + i) * 7 + c2] = JJ[i * 6 + c2]; J[((size_t)f * nr + i) * 7 + 6] = 0.0; }
    }
}

// one WG per IMU factor; J: 480/factor = [15x7 | 15x9 | 15x7 | 15x9]
__global__ void k_eval_imu(DevP P, const double* x, double* r, double* J) {
    using namespace vd;
    __shared__ double Jraw[450];
    __shared__ double rr[15];
    const int f = blockIdx.x, t = threadIdx.x;
    const int i = P.imu_i[f], j = P.imu_j[f];
    if (t == 0) imu_raw(P.imu_c + (size_t)f * 287, V3{P.G[0], P.G[1], P.G[2]}, x + xo_pose(P, i), x + xo_sb(P, i), x + xo_pose(P, j), x + xo_sb(P, j), rr, Jraw);
    __syncthreads();
    const double* U = P.imu_U + (size_t)f * 225;
    for (int e = t; e < 495; e += blockDim.x) {
        if (e < 480) {
            if (!J) continue;
            // decode Evaluate layout -> (row, raw column or -1 for the 7th pose column)
            int row, col;
            if (e < 105) { row = e / 7; col = e % 7; col = col == 6 ? -1 : col; }
            else if (e < 240) { row = (e - 105) / 9; col = 6 + (e - 105) % 9; }
            else if (e < 345) { row = (e - 240) / 7; col = (e - 240) % 7; col = col == 6 ? -1 : 15 + col; }
            else { row = (e - 345) / 9; col = 21 + (e - 345) % 9; }
            double s = 0;
            if (col >= 0) for (int k = row; k < 15; ++k) s += U[row * 15 + k] * Jraw[k * 30 + col];
            J[(size_t)f * 480 + e] = s;
        } else {
            const int row = e - 480;
            double s = 0;
            for (int k = row; k < 15; ++k) s += U[row * 15 + k] * rr[k];
            r[(size_t)f * 15 + row] = s;
        }
    }
}

// thread per (factor, pose block, global coordinate).  r: 3/factor, J: 84 (ICP) / 42 (LPS) raw AutoDiff blocks
__global__ void k_eval_rel(DevP P, const double* x, int icp, double* r, double* J) {
    using namespace vd;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = icp ? 4 : 2, n = icp ? P.n_icp : P.n_lps;
    if (t >= n * nb * 7) return;
    const int f = t / (nb * 7), rem = t - f * nb * 7, b = rem / 7, k = rem - 7 * b;
    double r3[3], d3[3];
    if (icp) { const int* id = P.icp_ids + 4 * f; icp_eval1(P.icp_c + (size_t)f * 10, x + xo_pose(P, id[0]), x + xo_pose(P, id[1]), x + xo_pose(P, id[2]), x + xo_pose(P, id[3]), b, k, r3, d3); }
    else { const int* id = P.lps_ids + 2 * f; lps_eval1(P.lps_c + (size_t)f * 7, x + xo_pose(P, id[0]), x + xo_pose(P, id[1]), b, k, r3, d3); }
    if (b == 0 && k == 0) for (int q = 0; q < 3; ++q) r[3 * f + q] = r3[q];
    if (J) for (int q = 0; q < 3; ++q) J[((size_t)f * nb + b) * 21 + 7 * q + k] = d3[q];
}

// prior: r = r0 + J0 dx ; J blocks = J0 columns left-aligned in n x gsize row-major blocks
__global__ void k_eval_prior(DevP P, const double* x, double* r, double* J, const int* joff /*nblk: offset of block in J*/) {
    using namespace vd;
    extern __shared__ double dx[];
    const int t = threadIdx.x, n = P.pn;
    if (t < P.pnblk) {
        const int kind = P.pblk_kind[t];
        const int gs = kind == 0 || kind == 2 ? 7 : (kind == 1 ? 9 : 1);
        double d[9];
        prior_block_dx(gs, prior_block_ptr(P, x, t), P.px0 + P.pblk_xoff[t], d);
        const int ls = gs == 7 ? 6 : gs;
        for (int k = 0; k < ls; ++k) dx[P.pblk_col[t] + k] = d[k];
    }
    __syncthreads();
    for (int i = t; i < n; i += blockDim.x) {
        double s = P.pr0[i];
        for (int k = 0; k < n; ++k) s += P.pJ0[(size_t)k * n + i] * dx[k];
        r[i] = s;
    }
    if (!J) return;
    for (int b = 0; b < P.pnblk; ++b) {
        const int kind = P.pblk_kind[b];
        const int gs = kind == 0 || kind == 2 ? 7 : (kind == 1 ? 9 : 1), ls = gs == 7 ? 6 : gs;
        const int col = P.pblk_col[b];
        for (int e = t; e < n * gs; e += blockDim.x) {
            const int i = e / gs, c = e % gs;
            J[joff[b] + e] = c < ls ? P.pJ0[(size_t)(col + c) * n + i] : 0.0;
        }
    }
}

// one-time set-up at upload: IMU sqrt-information (imu_factor.h:64, hoisted out of Evaluate) and the
// prior contractions pH = J0^T J0, pg0 = J0^T r0, pc0 = r0^T r0.
// 15 x 15 Cholesky of the matrix in LDS `A` (row-major, lower triangle used), right-looking, 256 threads:
// column scale by one lane per row, trailing update by one thread per element.  Result: lower factor in A.
// Pivots through v_rsq_f64 + two Newton steps (1-2 ulp): the IEEE square root and divide are 1 k-cycle dependent chains each, thirty
// of them in a row here, and this kernel runs at the head of every frame's upload.  rinv (optional): 1 / L_jj per column.
__device__ inline bool chol15_wg(double* A, int* bad, double* rinv = nullptr) {
    const int t = threadIdx.x;
    const int i = t / 15, k = t - 15 * (t / 15);
    for (int j = 0; j < 15; ++j) {
        const double d = A[j * 15 + j];
        if (!(d > 0.0)) { if (t == 0) *bad = 1; }
        const double r = vd::rsqrt_nr(d);
        __syncthreads();
        if (t == j) { A[j * 15 + j] = d * r; if (rinv) rinv[j] = r; }
        else if (t > j && t < 15) A[t * 15 + j] *= r;
        __syncthreads();
        if (t < 225 && i > j && k > j && k <= i) A[i * 15 + k] -= A[i * 15 + j] * A[k * 15 + j];
        __syncthreads();
    }
    return true;
}

// U with U^T U = cov^-1: cov = C C^T, W = C^-1, cov^-1 = W^T W = L L^T, U = L^T   (imu_factor.h:64).  One workgroup of VIL_THREADS.
__device__ inline void imu_sqrtinfo_wg(const double* cov, double* U_out, int* status) {
    using namespace vd;
    const int t = threadIdx.x;
    __shared__ double C[225], W[225], A[225], rC[16];
    __shared__ int bad;
    if (t == 0) bad = 0;
    if (t < 225) { C[t] = cov[t]; W[t] = 0.0; }
    __syncthreads();
    chol15_wg(C, &bad, rC);
    __syncthreads();
    if (t < 15) {                                    // column t of W = C^-1 by forward substitution
        const int j = t;
        W[j * 15 + j] = rC[j];
        for (int i = j + 1; i < 15; ++i) { double s = 0; for (int k = j; k < i; ++k) s -= C[i * 15 + k] * W[k * 15 + j]; W[i * 15 + j] = s * rC[i]; }
    }
    __syncthreads();
    if (t < 225) { const int i = t / 15, j = t - 15 * i; double s = 0; if (j <= i) for (int k = i; k < 15; ++k) s += W[k * 15 + i] * W[k * 15 + j]; A[t] = s; }
    __syncthreads();
    chol15_wg(A, &bad);
    if (t < 225) { const int i = t / 15, j = t - 15 * i; U_out[t] = j >= i ? A[j * 15 + i] : 0.0; }   // U = L^T
    if (t == 0 && (bad || !(A[224] == A[224]))) atomicExch(status, -4);
}

__global__ __launch_bounds__(VIL_THREADS) void k_setup(DevP P, double* imu_U, int* status) {
    using namespace vd;
    const int b = blockIdx.x, t = threadIdx.x;
    if (b < P.n_imu) { imu_sqrtinfo_wg(P.imu_c + (size_t)b * 287 + 62, imu_U + (size_t)b * 225, status); return; }
    const int n = P.pn;
    if (n <= 0) return;
    const int nb = gridDim.x - P.n_imu;
    for (int e = (b - P.n_imu) * blockDim.x + t; e < n * n + n + 1; e += nb * blockDim.x) {
        if (e < n * n) {
            const int i = e / n, k = e % n;
            double s = 0;
            for (int q = 0; q < n; ++q) s += P.pJ0[(size_t)i * n + q] * P.pJ0[(size_t)k * n + q];
            P.pH[e] = s;
        } else if (e < n * n + n) {
            const int i = e - n * n;
            double s = 0;
            for (int q = 0; q < n; ++q) s += P.pJ0[(size_t)i * n + q] * P.pr0[q];
            P.pg0[i] = s;
        } else {
            double s = 0;
            for (int q = 0; q < n; ++q) s += P.pr0[q] * P.pr0[q];
            P.pc0[0] = s;
        }
    }
}

// mirror + export of one linearisation (vil_linearize)
__global__ void k_mirror(DevP P, int set) {
    const int D = P.D;
    double* S = P.sys[set].S;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < D * D; e += gridDim.x * blockDim.x) { const int i = e / D, j = e % D; if (i < j) S[(size_t)j * D + i] = S[e]; }
}
