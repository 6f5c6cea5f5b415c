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
struct VisJ { double r[2]; double Ji[12], Jj[12], Jex[12], Jl[2], Jt[2]; };

__device__ __forceinline__ void visual_eval(const double* c, const M3& Ri, V3 Pi, const M3& Rj, V3 Pj, const M3& Ric, V3 tic,
                                            double lam, double td, double s, double k_tr, int use_td, VisJ& o) {
    V3 pi{c[0], c[1], c[2]}, pj{c[3], c[4], c[5]};
    const V3 veli{c[6], c[7], 0.0}, velj{c[8], c[9], 0.0};
    if (use_td) {
        const double ti = td - c[10] + k_tr * c[12], tj = td - c[11] + k_tr * c[13];
        pi = pi - ti * veli;
        pj = pj - tj * velj;
    }
    const double il = 1.0 / lam;
    const V3 xci = il * pi;
    const V3 xbi = mul(Ric, xci) + tic;
    const V3 xw = mul(Ri, xbi) + Pi;
    const V3 xbj = mulT(Rj, xw - Pj);
    const V3 xcj = mulT(Ric, xbj - tic);
    const double iz = 1.0 / xcj.z;
    o.r[0] = s * (xcj.x * iz - pj.x);
    o.r[1] = s * (xcj.y * iz - pj.y);
    // E = s [[1/z 0 -x/z^2],[0 1/z -y/z^2]]
    const V3 e0{s * iz, 0.0, -s * xcj.x * iz * iz}, e1{0.0, s * iz, -s * xcj.y * iz * iz};
    // C = E Ric^T ; A = C Rj^T ; B = A Ri ; T = B Ric       (all 2x3, row vectors)
    const V3 c0 = rowmulT(e0, Ric), c1 = rowmulT(e1, Ric);
    const V3 a0 = rowmulT(c0, Rj), a1 = rowmulT(c1, Rj);
    const V3 b0 = rowmul(a0, Ri), b1 = rowmul(a1, Ri);
    const V3 t0 = rowmul(b0, Ric), t1 = rowmul(b1, Ric);
    // pose i: [A | -B [xbi]x] ;  row^T [v]x = (row x v)^T
    const V3 ri0 = cross(xbi, b0), ri1 = cross(xbi, b1);
    o.Ji[0] = a0.x; o.Ji[1] = a0.y; o.Ji[2] = a0.z; o.Ji[3] = ri0.x; o.Ji[4] = ri0.y; o.Ji[5] = ri0.z;
    o.Ji[6] = a1.x; o.Ji[7] = a1.y; o.Ji[8] = a1.z; o.Ji[9] = ri1.x; o.Ji[10] = ri1.y; o.Ji[11] = ri1.z;
    // pose j: [-A | C [xbj]x]
    const V3 rj0 = cross(c0, xbj), rj1 = cross(c1, xbj);
    o.Jj[0] = -a0.x; o.Jj[1] = -a0.y; o.Jj[2] = -a0.z; o.Jj[3] = rj0.x; o.Jj[4] = rj0.y; o.Jj[5] = rj0.z;
    o.Jj[6] = -a1.x; o.Jj[7] = -a1.y; o.Jj[8] = -a1.z; o.Jj[9] = rj1.x; o.Jj[10] = rj1.y; o.Jj[11] = rj1.z;
    // extrinsic: [B - C | -T [xci]x + E [T xci + w]x],  T xci + w == xcj
    const V3 rx0 = cross(xci, t0) + cross(e0, xcj), rx1 = cross(xci, t1) + cross(e1, xcj);
    o.Jex[0] = b0.x - c0.x; o.Jex[1] = b0.y - c0.y; o.Jex[2] = b0.z - c0.z; o.Jex[3] = rx0.x; o.Jex[4] = rx0.y; o.Jex[5] = rx0.z;
    o.Jex[6] = b1.x - c1.x; o.Jex[7] = b1.y - c1.y; o.Jex[8] = b1.z - c1.z; o.Jex[9] = rx1.x; o.Jex[10] = rx1.y; o.Jex[11] = rx1.z;
    // inverse depth: E T pts_i_td (-1/lam^2) = -(T xci)/lam
    o.Jl[0] = -dot(t0, xci) * il;
    o.Jl[1] = -dot(t1, xci) * il;
    if (use_td) {
        o.Jt[0] = -dot(t0, veli) * il + s * velj.x;
        o.Jt[1] = -dot(t1, veli) * il + s * velj.y;
    } else { o.Jt[0] = 0.0; o.Jt[1] = 0.0; }
}

// ------------------------------------------------------------------------------------------------
// A14 LiDAR point factors in window-pose form: p_b = R_bl p_l + t_bl, p_w = R p_b + P
//   plane (lidarFactor.hpp:106-138): r = n.p_w + d            J = [n^T | (p_b x R^T n)^T]
//   edge  (lidarFactor.hpp:12-55):   r = ((p_w-a)x(p_w-b))/|a-b|   J = [-[dh]x | [dh]x R [p_b]x], dh=(a-b)/|a-b|
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void plane_eval(V3 pl, V3 n, double d, const M3& Rbl, V3 tbl, const M3& R, V3 P, double& r, double* J6) {
    const V3 pb = mul(Rbl, pl) + tbl;
    const V3 pw = mul(R, pb) + P;
    r = dot(n, pw) + d;
    const V3 rn = mulT(R, n);
    const V3 jr = cross(pb, rn);
    J6[0] = n.x; J6[1] = n.y; J6[2] = n.z; J6[3] = jr.x; J6[4] = jr.y; J6[5] = jr.z;
}

__device__ __forceinline__ void edge_eval(V3 pl, V3 a, V3 b, const M3& Rbl, V3 tbl, const M3& R, V3 P, double* r3, double* J18) {
    const V3 pb = mul(Rbl, pl) + tbl;
    const V3 pw = mul(R, pb) + P;
    const V3 de = a - b;
    const double inv = 1.0 / sqrt(dot(de, de));
    const V3 nu = cross(pw - a, pw - b);
    r3[0] = nu.x * inv; r3[1] = nu.y * inv; r3[2] = nu.z * inv;
    const V3 dh = inv * de;
    // G = -[dh]x  (rows g0,g1,g2)
    const V3 g0{0.0, dh.z, -dh.y}, g1{-dh.z, 0.0, dh.x}, g2{dh.y, -dh.x, 0.0};
    // rotation part: G (-R [pb]x): row_i = -(g_i^T R) [pb]x = -( (R^T g_i) x pb ) = pb x (R^T g_i)
    const V3 q0 = cross(pb, mulT(R, g0)), q1 = cross(pb, mulT(R, g1)), q2 = cross(pb, mulT(R, g2));
    J18[0] = g0.x; J18[1] = g0.y; J18[2] = g0.z; J18[3] = q0.x; J18[4] = q0.y; J18[5] = q0.z;
    J18[6] = g1.x; J18[7] = g1.y; J18[8] = g1.z; J18[9] = q1.x; J18[10] = q1.y; J18[11] = q1.z;
    J18[12] = g2.x; J18[13] = g2.y; J18[14] = g2.z; J18[15] = q2.x; J18[16] = q2.y; J18[17] = q2.z;
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

__device__ inline void imu_raw(const double* c, V3 G, const double* pi, const double* sbi, const double* pj, const double* sbj,
                               double* r /*15*/, double* J /*15x30 row-major, or nullptr*/) {
    const V3 Pi{pi[0], pi[1], pi[2]}, Pj{pj[0], pj[1], pj[2]};
    const Q4 Qi = qload(pi + 3), Qj = qload(pj + 3);
    const V3 Vi{sbi[0], sbi[1], sbi[2]}, Bai{sbi[3], sbi[4], sbi[5]}, Bgi{sbi[6], sbi[7], sbi[8]};
    const V3 Vj{sbj[0], sbj[1], sbj[2]}, Baj{sbj[3], sbj[4], sbj[5]}, Bgj{sbj[6], sbj[7], sbj[8]};
    const V3 dp{c[0], c[1], c[2]}, dv{c[7], c[8], c[9]};
    const Q4 dq{c[6], c[3], c[4], c[5]};
    const double dt = c[16];
    const M3 Jp_ba = loadM3(c + 17), Jp_bg = loadM3(c + 26), Jq_bg = loadM3(c + 35), Jv_ba = loadM3(c + 44), Jv_bg = loadM3(c + 53);
    const V3 dba = Bai - V3{c[10], c[11], c[12]}, dbg = Bgi - V3{c[13], c[14], c[15]};
    const V3 th = mul(Jq_bg, dbg);
    const Q4 qt = qmul(dq, Q4{1.0, 0.5 * th.x, 0.5 * th.y, 0.5 * th.z});    // corrected delta_q, not normalised
    const V3 vt = dv + mul(Jv_ba, dba) + mul(Jv_bg, dbg);
    const V3 pt = dp + mul(Jp_ba, dba) + mul(Jp_bg, dbg);
    const Q4 Qi_inv = qinv(Qi);
    const V3 tp = qrot(Qi_inv, (0.5 * dt * dt) * G + Pj - Pi - dt * Vi);
    const V3 tv = qrot(Qi_inv, dt * G + Vj - Vi);
    const Q4 qij = qmul(Qi_inv, Qj);
    const Q4 qe = qmul(qinv(qt), qij);
    r[0] = tp.x - pt.x; r[1] = tp.y - pt.y; r[2] = tp.z - pt.z;
    r[3] = 2 * qe.x; r[4] = 2 * qe.y; r[5] = 2 * qe.z;
    r[6] = tv.x - vt.x; r[7] = tv.y - vt.y; r[8] = tv.z - vt.z;
    r[9] = Baj.x - Bai.x; r[10] = Baj.y - Bai.y; r[11] = Baj.z - Bai.z;
    r[12] = Bgj.x - Bgi.x; r[13] = Bgj.y - Bgi.y; r[14] = Bgj.z - Bgi.z;
    if (!J) return;
    for (int i = 0; i < 450; ++i) J[i] = 0.0;
    const double qiv[4] = {Qi_inv.x, Qi_inv.y, Qi_inv.z, Qi_inv.w};
    const M3 RiT = quatR(qiv);
    M3 I3; for (int i = 0; i < 9; ++i) I3.m[i] = 0; I3.m[0] = I3.m[4] = I3.m[8] = 1;
    // pose_i
    put33(J, 30, 0, 0, RiT, -1.0);
    put33(J, 30, 0, 3, skewm(tp), 1.0);
    {   // -(Qleft(Qj^-1 Qi) Qright(qt)) bottom-right: -v pv^T + (w I + [v]x)(pw I - [pv]x)
        const Q4 q = qmul(qinv(Qj), Qi);
        M3 blk = mm(ql33(q), qr33(qt));
        const double v[3] = {q.x, q.y, q.z}, pv[3] = {qt.x, qt.y, qt.z};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) blk.m[3 * i + j] -= v[i] * pv[j];
        put33(J, 30, 3, 3, blk, -1.0);
    }
    put33(J, 30, 6, 3, skewm(tv), 1.0);
    // speedbias_i
    put33(J, 30, 0, 6, RiT, -dt);
    put33(J, 30, 0, 9, Jp_ba, -1.0);
    put33(J, 30, 0, 12, Jp_bg, -1.0);
    put33(J, 30, 3, 12, mm(ql33(qmul(qmul(qinv(Qj), Qi), dq)), Jq_bg), -1.0);
    put33(J, 30, 6, 6, RiT, -1.0);
    put33(J, 30, 6, 9, Jv_ba, -1.0);
    put33(J, 30, 6, 12, Jv_bg, -1.0);
    put33(J, 30, 9, 9, I3, -1.0);
    put33(J, 30, 12, 12, I3, -1.0);
    // pose_j
    put33(J, 30, 0, 15, RiT, 1.0);
    put33(J, 30, 3, 18, ql33(qe), 1.0);
    // speedbias_j
    put33(J, 30, 6, 21, RiT, 1.0);
    put33(J, 30, 9, 24, I3, 1.0);
    put33(J, 30, 12, 27, I3, 1.0);
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
// One thread differentiates w.r.t. ONE 7-dof pose block (seeded), so N = 7.
// ------------------------------------------------------------------------------------------------
struct J7 {
    double a, v[7];
    __device__ J7() : a(0) { for (int i = 0; i < 7; ++i) v[i] = 0; }
    __device__ J7(double s) : a(s) { for (int i = 0; i < 7; ++i) v[i] = 0; }
};
__device__ inline J7 operator+(const J7& x, const J7& y) { J7 r; r.a = x.a + y.a; for (int i = 0; i < 7; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
__device__ inline J7 operator-(const J7& x, const J7& y) { J7 r; r.a = x.a - y.a; for (int i = 0; i < 7; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
__device__ inline J7 operator-(const J7& x) { J7 r; r.a = -x.a; for (int i = 0; i < 7; ++i) r.v[i] = -x.v[i]; return r; }
__device__ inline J7 operator*(const J7& x, const J7& y) { J7 r; r.a = x.a * y.a; for (int i = 0; i < 7; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
__device__ inline J7 operator/(const J7& x, const J7& y) { J7 r; const double inv = 1.0 / y.a; r.a = x.a * inv; for (int i = 0; i < 7; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv; return r; }
__device__ inline J7 jsin(const J7& x) { J7 r; r.a = sin(x.a); const double c = cos(x.a); for (int i = 0; i < 7; ++i) r.v[i] = c * x.v[i]; return r; }
__device__ inline J7 jacos(const J7& x) { J7 r; r.a = acos(x.a); const double d = -1.0 / sqrt(1.0 - x.a * x.a); for (int i = 0; i < 7; ++i) r.v[i] = d * x.v[i]; return r; }
struct JQ { J7 w, x, y, z; };
struct JV { J7 x, y, z; };
__device__ inline JQ jq_load(const double* p, bool seed) {
    JQ q; q.x = J7(p[3]); q.y = J7(p[4]); q.z = J7(p[5]); q.w = J7(p[6]);
    if (seed) { q.x.v[3] = 1; q.y.v[4] = 1; q.z.v[5] = 1; q.w.v[6] = 1; }
    return q;
}
__device__ inline JV jv_load(const double* p, bool seed) {
    JV v; v.x = J7(p[0]); v.y = J7(p[1]); v.z = J7(p[2]);
    if (seed) { v.x.v[0] = 1; v.y.v[1] = 1; v.z.v[2] = 1; }
    return v;
}
__device__ inline JQ jq_mul(const JQ& a, const JQ& b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
__device__ inline JQ jq_inv(const JQ& q) { J7 n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z; return {q.w / n2, -(q.x / n2), -(q.y / n2), -(q.z / n2)}; }
__device__ inline JV jv_cross(const JV& a, const JV& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline JV jq_rot(const JQ& q, const JV& v) {   // Eigen _transformVector
    JV u{q.x, q.y, q.z};
    JV uv = jv_cross(u, v);
    uv = {uv.x + uv.x, uv.y + uv.y, uv.z + uv.z};
    JV w = jv_cross(u, uv);
    return {v.x + q.w * uv.x + w.x, v.y + q.w * uv.y + w.y, v.z + q.w * uv.z + w.z};
}
__device__ inline JQ jq_slerp(const JQ& a, double t, const JQ& b) {   // Eigen::QuaternionBase::slerp
    const double one = 1.0 - 2.220446049250313e-16;
    J7 d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
    J7 ad = d.a < 0.0 ? -d : d;
    J7 s0, s1;
    if (ad.a >= one) { s0 = J7(1.0 - t); s1 = J7(t); }
    else {
        J7 th = jacos(ad), st = jsin(th);
        s0 = jsin(th * J7(1.0 - t)) / st;
        s1 = jsin(th * J7(t)) / st;
    }
    if (d.a < 0.0) s1 = -s1;
    return {s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}

// ICP: c[10] = ta tb tc td ti tj PIJ(3) s ; poses a,b,c,d ; `seed` in 0..3 selects the differentiated block
__device__ inline void icp_eval(const double* c, const double* pa, const double* pb, const double* pc, const double* pd, int seed, double* r3, double* J21 /*3x7*/) {
    const double ta = c[0], tb = c[1], tc = c[2], td = c[3], ti = c[4], tj = c[5];
    JQ Qa = jq_load(pa, seed == 0), Qb = jq_load(pb, seed == 1), Qc = jq_load(pc, seed == 2), Qd = jq_load(pd, seed == 3);
    JV Pa = jv_load(pa, seed == 0), Pb = jv_load(pb, seed == 1), Pc = jv_load(pc, seed == 2), Pd = jv_load(pd, seed == 3);
    JQ Qi = jq_slerp(Qa, (ti - ta) / (tb - ta), Qb);
    JQ Qj = jq_slerp(Qc, (tj - tc) / (td - tc), Qd);
    const J7 wab(tb - ta), tia(ti - ta), wcd(td - tc), tjc(tj - tc);
    JV Pi{Pa.x + (Pb.x - Pa.x) / wab * tia, Pa.y + (Pb.y - Pa.y) / wab * tia, Pa.z + (Pb.z - Pa.z) / wab * tia};
    JV Pj{Pc.x + (Pd.x - Pc.x) / wcd * tjc, Pc.y + (Pd.y - Pc.y) / wcd * tjc, Pc.z + (Pd.z - Pc.z) / wcd * tjc};
    JQ temQ = jq_mul(jq_inv(Qj), Qi);
    JV dP{Pj.x - Pi.x, Pj.y - Pi.y, Pj.z - Pi.z};
    JV tem = jq_rot(jq_inv(Qi), dP);
    JV df{J7(c[6]) - tem.x, J7(c[7]) - tem.y, J7(c[8]) - tem.z};
    JV RES = jq_rot(temQ, df);
    J7 o0 = RES.x * J7(c[9]), o2 = RES.z * J7(c[9]);
    r3[0] = o0.a; r3[1] = 0.0; r3[2] = o2.a;
    for (int k = 0; k < 7; ++k) { J21[k] = o0.v[k]; J21[7 + k] = 0.0; J21[14 + k] = o2.v[k]; }
}
// LPS: c[7] = tl tr tk q(x y z w) ; poses a,b ; seed in 0..1
__device__ inline void lps_eval(const double* c, const double* pa, const double* pb, int seed, double* r3, double* J21) {
    JQ Qa = jq_load(pa, seed == 0), Qb = jq_load(pb, seed == 1);
    JQ Qi = jq_slerp(Qa, (c[2] - c[0]) / (c[1] - c[0]), Qb);
    JQ Q1{J7(c[6]), J7(c[3]), J7(c[4]), J7(c[5])};
    JQ Q12 = jq_mul(jq_inv(Qi), Q1);
    J7 o0 = J7(2.0) * Q12.x / J7(0.01), o1 = J7(2.0) * Q12.y / J7(0.01), o2 = J7(2.0) * Q12.z / J7(0.01);
    r3[0] = o0.a; r3[1] = o1.a; r3[2] = o2.a;
    for (int k = 0; k < 7; ++k) { J21[k] = o0.v[k]; J21[7 + k] = o1.v[k]; J21[14 + k] = o2.v[k]; }
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
