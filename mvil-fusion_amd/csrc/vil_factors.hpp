// Device-side residual / Jacobian arithmetic of every factor class on the hot path, written from the
// condensed formulas for register-resident evaluation: one thread = one factor
// for the bulk classes (visual, LiDAR edge/plane), cooperative workgroups for the 15x30 IMU factor.
// Tangent Jacobians are 6 columns per pose ([dp | dtheta], right-multiplicative).
#pragma once
#include "vil_math.hpp"

namespace vd {

// ------------------------------------------------------------------------------------------------
// A6/A7 visual reprojection (+td)   projection_td_factor.cpp:34-141 / projection_factor.cpp:21-121
//   c[14] = pts_i(3) pts_j(3) vel_i(2) vel_j(2) td_i td_j row_i row_j
// out: r[2]; J blocks 2x6 row-major: Ji, Jj, Jex; Jl[2] (inverse depth); Jt[2] (td)
// ------------------------------------------------------------------------------------------------
template <class T> struct VisJT { T r[2]; T Ji[12], Jj[12], Jex[12], Jl[2], Jt[2]; };
using VisJ = VisJT<double>;

template <class T>
__device__ __forceinline__ void visual_eval_t(const double* c, const M3T<T>& Ri, V3T<T> Pi, const M3T<T>& Rj, V3T<T> Pj, const M3T<T>& Ric, V3T<T> tic,
                                              T lam, T td, T s, T k_tr, int use_td, VisJT<T>& o) {
    using V = V3T<T>;
    const T zero = (T)0;
    V pi{(T)c[0], (T)c[1], (T)c[2]}, pj{(T)c[3], (T)c[4], (T)c[5]};
    const V veli{(T)c[6], (T)c[7], zero}, velj{(T)c[8], (T)c[9], zero};
    if (use_td) {
        const T ti = td - (T)c[10] + k_tr * (T)c[12], tj = td - (T)c[11] + k_tr * (T)c[13];
        pi = pi - ti * veli;
        pj = pj - tj * velj;
    }
    const T il = (T)1 / lam;
    const V xci = il * pi;
    const V xbi = mul(Ric, xci) + tic;
    const V xw = mul(Ri, xbi) + Pi;
    const V xbj = mulT(Rj, xw - Pj);
    const V xcj = mulT(Ric, xbj - tic);
    const T iz = (T)1 / xcj.z;
    o.r[0] = s * (xcj.x * iz - pj.x);
    o.r[1] = s * (xcj.y * iz - pj.y);
    // E = s [[1/z 0 -x/z^2],[0 1/z -y/z^2]]
    const V e0{s * iz, zero, -s * xcj.x * iz * iz}, e1{zero, s * iz, -s * xcj.y * iz * iz};
    // C = E Ric^T ; A = C Rj^T ; B = A Ri ; T = B Ric       (all 2x3, row vectors)
    const V c0 = rowmulT(e0, Ric), c1 = rowmulT(e1, Ric);
    const V a0 = rowmulT(c0, Rj), a1 = rowmulT(c1, Rj);
    const V b0 = rowmul(a0, Ri), b1 = rowmul(a1, Ri);
    const V t0 = rowmul(b0, Ric), t1 = rowmul(b1, Ric);
    // pose i: [A | -B [xbi]x] ;  row^T [v]x = (row x v)^T
    const V ri0 = cross(xbi, b0), ri1 = cross(xbi, b1);
    o.Ji[0] = a0.x; o.Ji[1] = a0.y; o.Ji[2] = a0.z; o.Ji[3] = ri0.x; o.Ji[4] = ri0.y; o.Ji[5] = ri0.z;
    o.Ji[6] = a1.x; o.Ji[7] = a1.y; o.Ji[8] = a1.z; o.Ji[9] = ri1.x; o.Ji[10] = ri1.y; o.Ji[11] = ri1.z;
    // pose j: [-A | C [xbj]x]
    const V rj0 = cross(c0, xbj), rj1 = cross(c1, xbj);
    o.Jj[0] = -a0.x; o.Jj[1] = -a0.y; o.Jj[2] = -a0.z; o.Jj[3] = rj0.x; o.Jj[4] = rj0.y; o.Jj[5] = rj0.z;
    o.Jj[6] = -a1.x; o.Jj[7] = -a1.y; o.Jj[8] = -a1.z; o.Jj[9] = rj1.x; o.Jj[10] = rj1.y; o.Jj[11] = rj1.z;
    // extrinsic: [B - C | -T [xci]x + E [T xci + w]x],  T xci + w == xcj
    const V rx0 = cross(xci, t0) + cross(e0, xcj), rx1 = cross(xci, t1) + cross(e1, xcj);
    o.Jex[0] = b0.x - c0.x; o.Jex[1] = b0.y - c0.y; o.Jex[2] = b0.z - c0.z; o.Jex[3] = rx0.x; o.Jex[4] = rx0.y; o.Jex[5] = rx0.z;
    o.Jex[6] = b1.x - c1.x; o.Jex[7] = b1.y - c1.y; o.Jex[8] = b1.z - c1.z; o.Jex[9] = rx1.x; o.Jex[10] = rx1.y; o.Jex[11] = rx1.z;
    // inverse depth: E T pts_i_td (-1/lam^2) = -(T xci)/lam
    o.Jl[0] = -dot(t0, xci) * il;
    o.Jl[1] = -dot(t1, xci) * il;
    if (use_td) {
        o.Jt[0] = -dot(t0, veli) * il + s * velj.x;
        o.Jt[1] = -dot(t1, veli) * il + s * velj.y;
    } else { o.Jt[0] = zero; o.Jt[1] = zero; }
}

// fp64 entry (reference arithmetic) and the fp32-evaluation variant: same formulas in float, result widened for the
// fp64 accumulation.  The rotation matrices are formed in fp64 from the quaternions and rounded once.
__device__ __forceinline__ void visual_eval(const double* c, const M3& Ri, V3 Pi, const M3& Rj, V3 Pj, const M3& Ric, V3 tic,
                                            double lam, double td, double s, double k_tr, int use_td, VisJ& o) {
    visual_eval_t<double>(c, Ri, Pi, Rj, Pj, Ric, tic, lam, td, s, k_tr, use_td, o);
}
__device__ __forceinline__ void visual_eval_f32(const double* c, const M3& Ri, V3 Pi, const M3& Rj, V3 Pj, const M3& Ric, V3 tic,
                                                double lam, double td, double s, double k_tr, int use_td, VisJ& o) {
    VisJT<float> of;
    visual_eval_t<float>(c, m3cast<float>(Ri), v3cast<float>(Pi), m3cast<float>(Rj), v3cast<float>(Pj), m3cast<float>(Ric), v3cast<float>(tic),
                         (float)lam, (float)td, (float)s, (float)k_tr, use_td, of);
    o.r[0] = of.r[0]; o.r[1] = of.r[1]; o.Jl[0] = of.Jl[0]; o.Jl[1] = of.Jl[1]; o.Jt[0] = of.Jt[0]; o.Jt[1] = of.Jt[1];
#pragma unroll
    for (int k = 0; k < 12; ++k) { o.Ji[k] = of.Ji[k]; o.Jj[k] = of.Jj[k]; o.Jex[k] = of.Jex[k]; }
}

// ------------------------------------------------------------------------------------------------
// A14 LiDAR point factors in window-pose form: p_b = R_bl p_l + t_bl, p_w = R p_b + P
//   plane (lidarFactor.hpp:106-138): r = n.p_w + d            J = [n^T | (p_b x R^T n)^T]
//   edge  (lidarFactor.hpp:12-55):   r = ((p_w-a)x(p_w-b))/|a-b|   J = [-[dh]x | [dh]x R [p_b]x], dh=(a-b)/|a-b|
// PREC = 1: the same arithmetic in float, widened on output.
// ------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ void plane_eval_t(V3T<T> pl, V3T<T> n, T d, const M3T<T>& Rbl, V3T<T> tbl, const M3T<T>& R, V3T<T> P, double& r, double* J6) {
    const V3T<T> pb = mul(Rbl, pl) + tbl;
    const V3T<T> pw = mul(R, pb) + P;
    r = dot(n, pw) + d;
    const V3T<T> rn = mulT(R, n);
    const V3T<T> jr = cross(pb, rn);
    J6[0] = n.x; J6[1] = n.y; J6[2] = n.z; J6[3] = jr.x; J6[4] = jr.y; J6[5] = jr.z;
}

template <class T>
__device__ __forceinline__ void edge_eval_t(V3T<T> pl, V3T<T> a, V3T<T> b, const M3T<T>& Rbl, V3T<T> tbl, const M3T<T>& R, V3T<T> P, double* r3, double* J18) {
    using V = V3T<T>;
    const T zero = (T)0;
    const V pb = mul(Rbl, pl) + tbl;
    const V pw = mul(R, pb) + P;
    const V de = a - b;
    const T inv = (T)1 / sqrt(dot(de, de));
    const V nu = cross(pw - a, pw - b);
    r3[0] = nu.x * inv; r3[1] = nu.y * inv; r3[2] = nu.z * inv;
    const V dh = inv * de;
    // G = -[dh]x  (rows g0,g1,g2)
    const V g0{zero, dh.z, -dh.y}, g1{-dh.z, zero, dh.x}, g2{dh.y, -dh.x, zero};
    // rotation part: G (-R [pb]x): row_i = -(g_i^T R) [pb]x = -( (R^T g_i) x pb ) = pb x (R^T g_i)
    const V q0 = cross(pb, mulT(R, g0)), q1 = cross(pb, mulT(R, g1)), q2 = cross(pb, mulT(R, g2));
    J18[0] = g0.x; J18[1] = g0.y; J18[2] = g0.z; J18[3] = q0.x; J18[4] = q0.y; J18[5] = q0.z;
    J18[6] = g1.x; J18[7] = g1.y; J18[8] = g1.z; J18[9] = q1.x; J18[10] = q1.y; J18[11] = q1.z;
    J18[12] = g2.x; J18[13] = g2.y; J18[14] = g2.z; J18[15] = q2.x; J18[16] = q2.y; J18[17] = q2.z;
}

__device__ __forceinline__ void plane_eval(V3 pl, V3 n, double d, const M3& Rbl, V3 tbl, const M3& R, V3 P, double& r, double* J6, int prec = 0) {
    if (prec) plane_eval_t<float>(v3cast<float>(pl), v3cast<float>(n), (float)d, m3cast<float>(Rbl), v3cast<float>(tbl), m3cast<float>(R), v3cast<float>(P), r, J6);
    else plane_eval_t<double>(pl, n, d, Rbl, tbl, R, P, r, J6);
}
__device__ __forceinline__ void edge_eval(V3 pl, V3 a, V3 b, const M3& Rbl, V3 tbl, const M3& R, V3 P, double* r3, double* J18, int prec = 0) {
    if (prec) edge_eval_t<float>(v3cast<float>(pl), v3cast<float>(a), v3cast<float>(b), m3cast<float>(Rbl), v3cast<float>(tbl), m3cast<float>(R), v3cast<float>(P), r3, J18);
    else edge_eval_t<double>(pl, a, b, Rbl, tbl, R, P, r3, J18);
}

// ------------------------------------------------------------------------------------------------
// A4 IMU factor, raw (un-whitened) residual and 15x30 Jacobian   imu_factor.h:19-181,
// integration_base.h:175-201.  Column layout of J: [pose_i 0..5 | sb_i 6..14 | pose_j 15..20 | sb_j 21..29]
// Single-thread routine; the sweep calls it from one lane and whitens / contracts cooperatively.
// ------------------------------------------------------------------------------------------------
__device__ inline void put33(double* J, int ld, int r0, int c0, const M3& m, double s) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J[(r0 + i) * ld + c0 + j] = s * m.m[3 * i + j];
}
__device__ inline M3 skewm(V3 v) { M3 r; r.m[0] = 0; r.m[1] = -v.z; r.m[2] = v.y; r.m[3] = v.z; r.m[4] = 0; r.m[5] = -v.x; r.m[6] = -v.y; r.m[7] = v.x; r.m[8] = 0; return r; }
__device__ inline M3 mm(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j]; return r; }
// bottom-right 3x3 of Qleft(q) = w I + [v]x ; of Qright(q) = w I - [v]x   (utility.h:47-64)
__device__ inline M3 ql33(Q4 q) { M3 r = skewm({q.x, q.y, q.z}); r.m[0] += q.w; r.m[4] += q.w; r.m[8] += q.w; return r; }
__device__ inline M3 qr33(Q4 q) { M3 r = skewm({-q.x, -q.y, -q.z}); r.m[0] += q.w; r.m[4] += q.w; r.m[8] += q.w; return r; }

struct ImuCommon {
    M3 RiT; V3 tp, tv; Q4 qt, qe, qji, dq; double dt;
    V3 Bai, Baj, Bgi, Bgj, pt, vt;
};
__device__ inline void imu_common(const double* c, V3 G, const double* pi, const double* sbi, const double* pj, const double* sbj, ImuCommon& o) {
    const V3 Pi{pi[0], pi[1], pi[2]}, Pj{pj[0], pj[1], pj[2]};
    const Q4 Qi = qload(pi + 3), Qj = qload(pj + 3);
    const V3 Vi{sbi[0], sbi[1], sbi[2]}, Vj{sbj[0], sbj[1], sbj[2]};
    o.Bai = V3{sbi[3], sbi[4], sbi[5]}; o.Bgi = V3{sbi[6], sbi[7], sbi[8]};
    o.Baj = V3{sbj[3], sbj[4], sbj[5]}; o.Bgj = V3{sbj[6], sbj[7], sbj[8]};
    const V3 dp{c[0], c[1], c[2]}, dv{c[7], c[8], c[9]};
    o.dq = Q4{c[6], c[3], c[4], c[5]};
    o.dt = c[16];
    const double dt = o.dt;
    const M3 Jp_ba = loadM3(c + 17), Jp_bg = loadM3(c + 26), Jq_bg = loadM3(c + 35), Jv_ba = loadM3(c + 44), Jv_bg = loadM3(c + 53);
    const V3 dba = o.Bai - V3{c[10], c[11], c[12]}, dbg = o.Bgi - V3{c[13], c[14], c[15]};
    const V3 th = mul(Jq_bg, dbg);
    o.qt = qmul(o.dq, Q4{1.0, 0.5 * th.x, 0.5 * th.y, 0.5 * th.z});    // corrected delta_q, not normalised
    o.vt = dv + mul(Jv_ba, dba) + mul(Jv_bg, dbg);
    o.pt = dp + mul(Jp_ba, dba) + mul(Jp_bg, dbg);
    const Q4 Qi_inv = qinv(Qi);
    o.tp = qrot(Qi_inv, (0.5 * dt * dt) * G + Pj - Pi - dt * Vi);
    o.tv = qrot(Qi_inv, dt * G + Vj - Vi);
    o.qe = qmul(qinv(o.qt), qmul(Qi_inv, Qj));
    o.qji = qmul(qinv(Qj), Qi);
    const double qiv[4] = {Qi_inv.x, Qi_inv.y, Qi_inv.z, Qi_inv.w};
    o.RiT = quatR(qiv);
}
__device__ inline void imu_resid(const ImuCommon& o, double* r) {
    r[0] = o.tp.x - o.pt.x; r[1] = o.tp.y - o.pt.y; r[2] = o.tp.z - o.pt.z;
    r[3] = 2 * o.qe.x; r[4] = 2 * o.qe.y; r[5] = 2 * o.qe.z;
    r[6] = o.tv.x - o.vt.x; r[7] = o.tv.y - o.vt.y; r[8] = o.tv.z - o.vt.z;
    r[9] = o.Baj.x - o.Bai.x; r[10] = o.Baj.y - o.Bai.y; r[11] = o.Baj.z - o.Bai.z;
    r[12] = o.Bgj.x - o.Bgi.x; r[13] = o.Bgj.y - o.Bgi.y; r[14] = o.Bgj.z - o.Bgi.z;
}
#define IMU_NBLOCKS 17
// block b of the raw 15x30 Jacobian: rows r0.., cols c0.. get s * m
__device__ inline void imu_block(const ImuCommon& o, const double* c, int b, int& r0, int& c0, M3& m, double& s) {
    M3 I3; for (int i = 0; i < 9; ++i) I3.m[i] = 0; I3.m[0] = I3.m[4] = I3.m[8] = 1;
    s = 1.0;
    switch (b) {
        case 0: r0 = 0; c0 = 0; m = o.RiT; s = -1.0; break;                        // pose_i
        case 1: r0 = 0; c0 = 3; m = skewm(o.tp); break;
        case 2: {   // -(Qleft(Qj^-1 Qi) Qright(qt)) bottom-right: -v pv^T + (w I + [v]x)(pw I - [pv]x)
            r0 = 3; c0 = 3; s = -1.0;
            m = mm(ql33(o.qji), qr33(o.qt));
            const double v[3] = {o.qji.x, o.qji.y, o.qji.z}, pv[3] = {o.qt.x, o.qt.y, o.qt.z};
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m.m[3 * i + j] -= v[i] * pv[j];
            break; }
        case 3: r0 = 6; c0 = 3; m = skewm(o.tv); break;
        case 4: r0 = 0; c0 = 6; m = o.RiT; s = -o.dt; break;                       // speedbias_i
        case 5: r0 = 0; c0 = 9; m = loadM3(c + 17); s = -1.0; break;
        case 6: r0 = 0; c0 = 12; m = loadM3(c + 26); s = -1.0; break;
        case 7: r0 = 3; c0 = 12; m = mm(ql33(qmul(o.qji, o.dq)), loadM3(c + 35)); s = -1.0; break;
        case 8: r0 = 6; c0 = 6; m = o.RiT; s = -1.0; break;
        case 9: r0 = 6; c0 = 9; m = loadM3(c + 44); s = -1.0; break;
        case 10: r0 = 6; c0 = 12; m = loadM3(c + 53); s = -1.0; break;
        case 11: r0 = 9; c0 = 9; m = I3; s = -1.0; break;
        case 12: r0 = 12; c0 = 12; m = I3; s = -1.0; break;
        case 13: r0 = 0; c0 = 15; m = o.RiT; break;                                  // pose_j
        case 14: r0 = 3; c0 = 18; m = ql33(o.qe); break;
        case 15: r0 = 6; c0 = 21; m = o.RiT; break;                                  // speedbias_j
        default: r0 = 9; c0 = 24; m = I3; break;                                     // b == 16: also (12,27) = I, written by the caller
    }
}
__device__ inline void imu_raw(const double* c, V3 G, const double* pi, const double* sbi, const double* pj, const double* sbj,
                               double* r /*15*/, double* J /*15x30 row-major, or nullptr*/) {
    ImuCommon o;
    imu_common(c, G, pi, sbi, pj, sbj, o);
    imu_resid(o, r);
    if (!J) return;
    for (int i = 0; i < 450; ++i) J[i] = 0.0;
    for (int b = 0; b < IMU_NBLOCKS; ++b) {
        int r0, c0; M3 m; double s;
        imu_block(o, c, b, r0, c0, m, s);
        put33(J, 30, r0, c0, m, s);
        if (b == 16) put33(J, 30, 12, 27, m, s);
    }
}

// sqrt_info = LLT(cov^-1).matrixL()^T (imu_factor.h:64): upper-triangular U with U^T U = cov^-1.
// cov is SPD: invert through its Cholesky factor, then factor the inverse.  Single thread, 15x15.
__device__ inline bool imu_sqrt_info(const double* cov, double* U /*225 row-major*/) {
    double C[225], W[225];
    for (int j = 0; j < 15; ++j) {                       // cov = C C^T
        double d = cov[j * 15 + j];
        for (int k = 0; k < j; ++k) d -= C[j * 15 + k] * C[j * 15 + k];
        if (!(d > 0.0)) return false;
        d = sqrt(d); C[j * 15 + j] = d;
        for (int i = j + 1; i < 15; ++i) { double s = cov[i * 15 + j]; for (int k = 0; k < j; ++k) s -= C[i * 15 + k] * C[j * 15 + k]; C[i * 15 + j] = s / d; }
    }
    for (int j = 0; j < 15; ++j) {                       // W = C^-1 (lower)
        for (int i = 0; i < 15; ++i) W[i * 15 + j] = 0.0;
        W[j * 15 + j] = 1.0 / C[j * 15 + j];
        for (int i = j + 1; i < 15; ++i) { double s = 0; for (int k = j; k < i; ++k) s -= C[i * 15 + k] * W[k * 15 + j]; W[i * 15 + j] = s / C[i * 15 + i]; }
    }
    double* A = C;                                        // A = W^T W = cov^-1 (reuse storage)
    for (int i = 0; i < 15; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = i; k < 15; ++k) s += W[k * 15 + i] * W[k * 15 + j]; U[i * 15 + j] = s; }
    for (int i = 0; i < 15; ++i) for (int j = 0; j <= i; ++j) A[i * 15 + j] = U[i * 15 + j];
    for (int i = 0; i < 225; ++i) U[i] = 0.0;
    for (int j = 0; j < 15; ++j) {                       // A = L L^T, U = L^T
        double d = A[j * 15 + j];
        for (int k = 0; k < j; ++k) d -= U[k * 15 + j] * U[k * 15 + j];
        if (!(d > 0.0)) return false;
        d = sqrt(d); U[j * 15 + j] = d;
        for (int i = j + 1; i < 15; ++i) { double s = A[i * 15 + j]; for (int k = 0; k < j; ++k) s -= U[k * 15 + i] * U[k * 15 + j]; U[j * 15 + i] = s / d; }
    }
    return true;
}

// ------------------------------------------------------------------------------------------------
// forward-mode duals for the two AutoDiff factors (A11 ICP lidar_backend.h:107-169, A12 LPS :45-80).
// One thread differentiates w.r.t. ONE global coordinate (seed = 7*block + k), so a dual carries a
// single partial; 7 threads rebuild what a ceres::Jet<double,7> holds for one pose block.
// ------------------------------------------------------------------------------------------------
struct D1 { double a, d; };
__device__ __forceinline__ D1 dc(double s) { return {s, 0.0}; }
__device__ __forceinline__ D1 operator+(D1 x, D1 y) { return {x.a + y.a, x.d + y.d}; }
__device__ __forceinline__ D1 operator-(D1 x, D1 y) { return {x.a - y.a, x.d - y.d}; }
__device__ __forceinline__ D1 operator-(D1 x) { return {-x.a, -x.d}; }
__device__ __forceinline__ D1 operator*(D1 x, D1 y) { return {x.a * y.a, x.a * y.d + x.d * y.a}; }
__device__ __forceinline__ D1 operator/(D1 x, D1 y) { const double inv = 1.0 / y.a; const double q = x.a * inv; return {q, (x.d - q * y.d) * inv}; }
__device__ __forceinline__ D1 dsin(D1 x) { double sn, cs; sincos(x.a, &sn, &cs); return {sn, cs * x.d}; }   // one range reduction for both
__device__ __forceinline__ D1 dacos(D1 x) { return {acos(x.a), -x.d / sqrt(1.0 - x.a * x.a)}; }
struct DQ { D1 w, x, y, z; };
struct DV { D1 x, y, z; };
__device__ __forceinline__ DQ dq_load(const double* p, int k /*seeded coordinate 3..6 or -1*/) {
    return {D1{p[6], k == 6 ? 1.0 : 0.0}, D1{p[3], k == 3 ? 1.0 : 0.0}, D1{p[4], k == 4 ? 1.0 : 0.0}, D1{p[5], k == 5 ? 1.0 : 0.0}};
}
__device__ __forceinline__ DV dv_load(const double* p, int k /*0..2 or -1*/) {
    return {D1{p[0], k == 0 ? 1.0 : 0.0}, D1{p[1], k == 1 ? 1.0 : 0.0}, D1{p[2], k == 2 ? 1.0 : 0.0}};
}
__device__ __forceinline__ DQ dq_mul(const DQ& a, const DQ& b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ DQ dq_inv(const DQ& q) { const D1 n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; return {q.w / n2, -(q.x / n2), -(q.y / n2), -(q.z / n2)}; }
__device__ __forceinline__ DV dv_cross(const DV& a, const DV& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ DV dq_rot(const DQ& q, const DV& v) {   // Eigen _transformVector
    const DV u{q.x, q.y, q.z};
    DV uv = dv_cross(u, v);
    uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
    const DV w = dv_cross(u, uv);
    return {v.x + q.w * uv.x + w.x, v.y + q.w * uv.y + w.y, v.z + q.w * uv.z + w.z};
}
__device__ __forceinline__ DQ dq_slerp(const DQ& a, double t, const DQ& b) {   // Eigen::QuaternionBase::slerp
    const double one = 1.0 - 2.220446049250313e-16;
    const D1 d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
    const D1 ad = d.a < 0.0 ? -d : d;
    D1 s0, s1;
    if (ad.a >= one) { s0 = dc(1.0 - t); s1 = dc(t); }
    else {
        const D1 th = dacos(ad), st = dsin(th);
        s0 = dsin(th * dc(1.0 - t)) / st;
        s1 = dsin(th * dc(t)) / st;
    }
    if (d.a < 0.0) s1 = -s1;
    return {s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}

// ICP: c[10] = ta tb tc td ti tj PIJ(3) s ; poses a,b,c,d ; differentiates w.r.t. coordinate k of block `blk`:
// writes r3 (values) and dr3 = d r / d (block blk, coordinate k)
__device__ inline void icp_eval1(const double* c, const double* pa, const double* pb, const double* pc, const double* pd, int blk, int k, double* r3, double* dr3) {
    const double ta = c[0], tb = c[1], tc = c[2], td = c[3], ti = c[4], tj = c[5];
    const DQ Qa = dq_load(pa, blk == 0 ? k : -1), Qb = dq_load(pb, blk == 1 ? k : -1), Qc = dq_load(pc, blk == 2 ? k : -1), Qd = dq_load(pd, blk == 3 ? k : -1);
    const DV Pa = dv_load(pa, blk == 0 ? k : -1), Pb = dv_load(pb, blk == 1 ? k : -1), Pc = dv_load(pc, blk == 2 ? k : -1), Pd = dv_load(pd, blk == 3 ? k : -1);
    const DQ Qi = dq_slerp(Qa, (ti - ta) / (tb - ta), Qb);
    const DQ Qj = dq_slerp(Qc, (tj - tc) / (td - tc), Qd);
    const D1 wab = dc(tb - ta), tia = dc(ti - ta), wcd = dc(td - tc), tjc = dc(tj - tc);
    const DV Pi{Pa.x + (Pb.x - Pa.x) / wab * tia, Pa.y + (Pb.y - Pa.y) / wab * tia, Pa.z + (Pb.z - Pa.z) / wab * tia};
    const DV Pj{Pc.x + (Pd.x - Pc.x) / wcd * tjc, Pc.y + (Pd.y - Pc.y) / wcd * tjc, Pc.z + (Pd.z - Pc.z) / wcd * tjc};
    const DQ temQ = dq_mul(dq_inv(Qj), Qi);
    const DV dP{Pj.x - Pi.x, Pj.y - Pi.y, Pj.z - Pi.z};
    const DV tem = dq_rot(dq_inv(Qi), dP);
    const DV df{dc(c[6]) - tem.x, dc(c[7]) - tem.y, dc(c[8]) - tem.z};
    const DV RES = dq_rot(temQ, df);
    const D1 o0 = RES.x * dc(c[9]), o2 = RES.z * dc(c[9]);
    r3[0] = o0.a; r3[1] = 0.0; r3[2] = o2.a;
    dr3[0] = o0.d; dr3[1] = 0.0; dr3[2] = o2.d;
}
// LPS: c[7] = tl tr tk q(x y z w) ; poses a,b
__device__ inline void lps_eval1(const double* c, const double* pa, const double* pb, int blk, int k, double* r3, double* dr3) {
    const DQ Qa = dq_load(pa, blk == 0 ? k : -1), Qb = dq_load(pb, blk == 1 ? k : -1);
    const DQ Qi = dq_slerp(Qa, (c[2] - c[0]) / (c[1] - c[0]), Qb);
    const DQ Q1{dc(c[6]), dc(c[3]), dc(c[4]), dc(c[5])};
    const DQ Q12 = dq_mul(dq_inv(Qi), Q1);
    const D1 o0 = dc(2.0) * Q12.x / dc(0.01), o1 = dc(2.0) * Q12.y / dc(0.01), o2 = dc(2.0) * Q12.z / dc(0.01);
    r3[0] = o0.a; r3[1] = o1.a; r3[2] = o2.a;
    dr3[0] = o0.d; dr3[1] = o1.d; dr3[2] = o2.d;
}
// mathematically-correct tangent option (vil_options.autodiff_quirk == 0): J[:,3:6] <- J[:,3:7] d(q (x) [1,dth/2])/d dth
__device__ inline void tangent_fix(const double* pose, double* J21) {
    const double x = pose[3], y = pose[4], z = pose[5], w = pose[6];
    for (int i = 0; i < 3; ++i) {
        const double q0 = J21[7 * i + 3], q1 = J21[7 * i + 4], q2 = J21[7 * i + 5], q3 = J21[7 * i + 6];
        J21[7 * i + 3] = 0.5 * (q0 * w + q1 * z - q2 * y - q3 * x);
        J21[7 * i + 4] = 0.5 * (-q0 * z + q1 * w + q2 * x - q3 * y);
        J21[7 * i + 5] = 0.5 * (q0 * y - q1 * x + q2 * w - q3 * z);
        J21[7 * i + 6] = 0.0;
    }
}

// prior dx for one kept block (marginalization_factor.cpp:362-382)
__device__ inline void prior_block_dx(int gs, const double* x, const double* x0, double* dx) {
    if (gs != 7) { for (int k = 0; k < gs; ++k) dx[k] = x[k] - x0[k]; return; }
    for (int k = 0; k < 3; ++k) dx[k] = x[k] - x0[k];
    const Q4 dq = qmul(qinv(qload(x0 + 3)), qload(x + 3));
    const double s = dq.w >= 0 ? 2.0 : -2.0;
    dx[3] = s * dq.x; dx[4] = s * dq.y; dx[5] = s * dq.z;
}

}  // namespace vd
