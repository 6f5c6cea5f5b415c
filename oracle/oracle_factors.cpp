// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).  PARITY UNPINNED.
//
// Evaluate()-compatible CPU restatement of every factor on the hot path.  Each function fills
// `r` (num_residuals) and, if J != nullptr, the row-major GLOBAL-size Jacobian blocks laid out one
// after another exactly as ceres hands `double** jacobians` to CostFunction::Evaluate.
#include "oracle.hpp"

namespace orc {

// ------------------------------------------------------------------------------------------------
// small dense helpers for the 15x15 IMU information square root
// ------------------------------------------------------------------------------------------------
// General inverse by Gauss-Jordan with partial pivoting (Eigen's covariance.inverse() is
// PartialPivLU based: imu_factor.h:64).
static bool inverse_n(int n, const double* A, double* Ainv) {
    std::vector<double> a(A, A + n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ainv[i * n + j] = (i == j);
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(a[r * n + c]) > std::fabs(a[p * n + c])) p = r;
        if (a[p * n + c] == 0.0) return false;
        if (p != c) for (int j = 0; j < n; ++j) { std::swap(a[p * n + j], a[c * n + j]); std::swap(Ainv[p * n + j], Ainv[c * n + j]); }
        double inv = 1.0 / a[c * n + c];
        for (int j = 0; j < n; ++j) { a[c * n + j] *= inv; Ainv[c * n + j] *= inv; }
        for (int r = 0; r < n; ++r) if (r != c) {
            double f = a[r * n + c];
            if (f != 0.0) for (int j = 0; j < n; ++j) { a[r * n + j] -= f * a[c * n + j]; Ainv[r * n + j] -= f * Ainv[c * n + j]; }
        }
    }
    return true;
}
// lower Cholesky, row-major n x n; returns false if not PD
bool cholesky_lower(int n, const double* A, double* Lo) {
    std::fill(Lo, Lo + n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= Lo[j * n + k] * Lo[j * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        Lo[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= Lo[i * n + k] * Lo[j * n + k];
            Lo[i * n + j] = s / d;
        }
    }
    return true;
}
// sqrt_info = LLT(cov^-1).matrixL()^T   (imu_factor.h:64), row-major upper triangular
bool imu_sqrt_info(const double* cov, double* U) {
    double inv[225], Lo[225];
    if (!inverse_n(15, cov, inv)) return false;
    // symmetrise (Eigen's LLT reads the lower triangle only)
    for (int i = 0; i < 15; ++i) for (int j = 0; j < i; ++j) inv[j * 15 + i] = inv[i * 15 + j];
    if (!cholesky_lower(15, inv, Lo)) return false;
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) U[i * 15 + j] = Lo[j * 15 + i];
    return true;
}

static inline void put33(double* J, int ld, int r0, int c0, const Mat3& m) {
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) J[(r0 + i) * ld + c0 + j] = m(i, j);
}

// ------------------------------------------------------------------------------------------------
// A4  IMUFactor::Evaluate  imu_factor.h:19-181  (+ IntegrationBase::evaluate integration_base.h:175-201)
// blocks: pose_i(7) speedbias_i(9) pose_j(7) speedbias_j(9); J = [15x7 | 15x9 | 15x7 | 15x9] consecutive
// ------------------------------------------------------------------------------------------------
void imu_evaluate(const double* c, const double* G3, const double* pi, const double* sbi,
                  const double* pj, const double* sbj, double* r, double* J) {
    const Vec3 Pi = vec_from(pi), Pj = vec_from(pj);
    const Quat Qi = quat_from_block(pi), Qj = quat_from_block(pj);
    const Vec3 Vi = vec_from(sbi), Bai = vec_from(sbi + 3), Bgi = vec_from(sbi + 6);
    const Vec3 Vj = vec_from(sbj), Baj = vec_from(sbj + 3), Bgj = vec_from(sbj + 6);
    const Vec3 G = vec_from(G3);
    const Vec3 delta_p = vec_from(c + 0);
    const Quat delta_q = {c[6], c[3], c[4], c[5]};
    const Vec3 delta_v = vec_from(c + 7);
    const Vec3 lin_ba = vec_from(c + 10), lin_bg = vec_from(c + 13);
    const double sum_dt = c[16];
    Mat3 dp_dba, dp_dbg, dq_dbg, dv_dba, dv_dbg;
    std::memcpy(dp_dba.m, c + 17, 72); std::memcpy(dp_dbg.m, c + 26, 72); std::memcpy(dq_dbg.m, c + 35, 72);
    std::memcpy(dv_dba.m, c + 44, 72); std::memcpy(dv_dbg.m, c + 53, 72);
    const double* cov = c + 62;

    const Vec3 dba = Bai - lin_ba, dbg = Bgi - lin_bg;
    // integration_base.h:191-193
    const Quat corrected_delta_q = qmul(delta_q, deltaQ(mat_vec(dq_dbg, dbg)));
    const Vec3 corrected_delta_v = delta_v + mat_vec(dv_dba, dba) + mat_vec(dv_dbg, dbg);
    const Vec3 corrected_delta_p = delta_p + mat_vec(dp_dba, dba) + mat_vec(dp_dbg, dbg);
    const Quat Qi_inv = qinv(Qi);
    // integration_base.h:195-199
    const Vec3 tp = qrot(Qi_inv, G * (0.5 * sum_dt * sum_dt) + Pj - Pi - Vi * sum_dt);
    const Vec3 tv = qrot(Qi_inv, G * sum_dt + Vj - Vi);
    const Quat qe = qmul(qinv(corrected_delta_q), qmul(Qi_inv, Qj));
    double raw[15];
    raw[0] = tp.x - corrected_delta_p.x; raw[1] = tp.y - corrected_delta_p.y; raw[2] = tp.z - corrected_delta_p.z;
    raw[3] = 2 * qe.x; raw[4] = 2 * qe.y; raw[5] = 2 * qe.z;
    raw[6] = tv.x - corrected_delta_v.x; raw[7] = tv.y - corrected_delta_v.y; raw[8] = tv.z - corrected_delta_v.z;
    raw[9] = Baj.x - Bai.x; raw[10] = Baj.y - Bai.y; raw[11] = Baj.z - Bai.z;
    raw[12] = Bgj.x - Bgi.x; raw[13] = Bgj.y - Bgi.y; raw[14] = Bgj.z - Bgi.z;

    double U[225];
    bool ok = imu_sqrt_info(cov, U);
    if (!ok) { for (int i = 0; i < 225; ++i) U[i] = NAN; }
    for (int i = 0; i < 15; ++i) { double s = 0; for (int k = 0; k < 15; ++k) s += U[i * 15 + k] * raw[k]; r[i] = s; }
    if (!J) return;

    const Mat3 RiT = quat_R(Qi_inv);  // Qi.inverse().toRotationMatrix()
    double Ji[15 * 7] = {0}, Jsi[15 * 9] = {0}, Jj[15 * 7] = {0}, Jsj[15 * 9] = {0};
    // O_P=0 O_R=3 O_V=6 O_BA=9 O_BG=12 (parameters.h:80-87)
    // pose_i  imu_factor.h:88-114
    put33(Ji, 7, 0, 0, mat_scale(RiT, -1.0));
    put33(Ji, 7, 0, 3, skew(tp));
    {
        double Lq[16], Rq[16];
        qleft44(qmul(qinv(Qj), Qi), Lq);
        qright44(corrected_delta_q, Rq);
        Mat3 blk;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += Lq[4 * (i + 1) + k] * Rq[4 * k + j + 1]; blk(i, j) = -s; }
        put33(Ji, 7, 3, 3, blk);
    }
    put33(Ji, 7, 6, 3, skew(tv));
    // speedbias_i  imu_factor.h:115-145
    put33(Jsi, 9, 0, 0, mat_scale(RiT, -sum_dt));
    put33(Jsi, 9, 0, 3, mat_scale(dp_dba, -1.0));
    put33(Jsi, 9, 0, 6, mat_scale(dp_dbg, -1.0));
    put33(Jsi, 9, 3, 6, mat_scale(mat_mul(qleft33(qmul(qmul(qinv(Qj), Qi), delta_q)), dq_dbg), -1.0));
    put33(Jsi, 9, 6, 0, mat_scale(RiT, -1.0));
    put33(Jsi, 9, 6, 3, mat_scale(dv_dba, -1.0));
    put33(Jsi, 9, 6, 6, mat_scale(dv_dbg, -1.0));
    put33(Jsi, 9, 9, 3, mat_scale(mat_ident(), -1.0));
    put33(Jsi, 9, 12, 6, mat_scale(mat_ident(), -1.0));
    // pose_j  imu_factor.h:146-163
    put33(Jj, 7, 0, 0, RiT);
    put33(Jj, 7, 3, 3, qleft33(qmul(qinv(corrected_delta_q), qmul(Qi_inv, Qj))));
    // speedbias_j  imu_factor.h:164-177
    put33(Jsj, 9, 6, 0, RiT);
    put33(Jsj, 9, 9, 3, mat_ident());
    put33(Jsj, 9, 12, 6, mat_ident());
    // J <- sqrt_info * J
    auto premul = [&](const double* in, int nc, double* out) {
        for (int i = 0; i < 15; ++i) for (int j = 0; j < nc; ++j) { double s = 0; for (int k = 0; k < 15; ++k) s += U[i * 15 + k] * in[k * nc + j]; out[i * nc + j] = s; }
    };
    premul(Ji, 7, J);
    premul(Jsi, 9, J + 105);
    premul(Jj, 7, J + 105 + 135);
    premul(Jsj, 9, J + 105 + 135 + 105);
}

// ------------------------------------------------------------------------------------------------
// A6 ProjectionTdFactor::Evaluate projection_td_factor.cpp:34-141 (use_td=1)
// A7 ProjectionFactor::Evaluate   projection_factor.cpp:21-121    (use_td=0: no td terms)
// blocks: pose_i(7) pose_j(7) ex(7) inv_depth(1) td(1);  J = [2x7|2x7|2x7|2x1|2x1] (td block zero for A7)
// ------------------------------------------------------------------------------------------------
void visual_evaluate(const double* c, double sqrt_info, double tr_over_row, int use_td,
                     const double* pi, const double* pj, const double* ex, double inv_dep, double td,
                     double* r, double* J) {
    const Vec3 Pi = vec_from(pi), Pj = vec_from(pj), tic = vec_from(ex);
    const Quat Qi = quat_from_block(pi), Qj = quat_from_block(pj), qic = quat_from_block(ex);
    Vec3 pts_i = vec_from(c), pts_j = vec_from(c + 3);
    const Vec3 vel_i = {c[6], c[7], 0.0}, vel_j = {c[8], c[9], 0.0};
    Vec3 pts_i_td = pts_i, pts_j_td = pts_j;
    if (use_td) {
        pts_i_td = pts_i - vel_i * (td - c[10] + tr_over_row * c[12]);
        pts_j_td = pts_j - vel_j * (td - c[11] + tr_over_row * c[13]);
    }
    const Vec3 pts_camera_i = pts_i_td * (1.0 / inv_dep);  // Eigen: vector / scalar
    const Vec3 pci = {pts_i_td.x / inv_dep, pts_i_td.y / inv_dep, pts_i_td.z / inv_dep};
    (void)pts_camera_i;
    const Vec3 pts_imu_i = qrot(qic, pci) + tic;
    const Vec3 pts_w = qrot(Qi, pts_imu_i) + Pi;
    const Vec3 pts_imu_j = qrot(qinv(Qj), pts_w - Pj);
    const Vec3 pcj = qrot(qinv(qic), pts_imu_j - tic);
    const double dep_j = pcj.z;
    r[0] = sqrt_info * (pcj.x / dep_j - pts_j_td.x);
    r[1] = sqrt_info * (pcj.y / dep_j - pts_j_td.y);
    if (!J) return;
    const Mat3 Ri = quat_R(Qi), Rj = quat_R(Qj), ric = quat_R(qic);
    double red[6] = {sqrt_info / dep_j, 0, -sqrt_info * pcj.x / (dep_j * dep_j),
                     0, sqrt_info / dep_j, -sqrt_info * pcj.y / (dep_j * dep_j)};
    auto reduce_into = [&](const Mat3& left, const Mat3& right, double* out /*2x7 row-major*/) {
        for (int i = 0; i < 2; ++i) {
            for (int j = 0; j < 3; ++j) {
                double a = 0, b = 0;
                for (int k = 0; k < 3; ++k) { a += red[3 * i + k] * left(k, j); b += red[3 * i + k] * right(k, j); }
                out[7 * i + j] = a; out[7 * i + 3 + j] = b;
            }
            out[7 * i + 6] = 0.0;
        }
    };
    const Mat3 ricT = mat_T(ric), RjT = mat_T(Rj);
    const Mat3 ricT_RjT = mat_mul(ricT, RjT);
    // pose_i  :92-102
    reduce_into(ricT_RjT, mat_mul(mat_mul(ricT_RjT, Ri), mat_scale(skew(pts_imu_i), -1.0)), J);
    // pose_j  :104-114
    reduce_into(mat_scale(ricT_RjT, -1.0), mat_mul(ricT, skew(pts_imu_j)), J + 14);
    // ex      :115-125
    const Mat3 tmp_r = mat_mul(mat_mul(ricT_RjT, Ri), ric);
    {
        Mat3 left = mat_mul(ricT, mat_sub(mat_mul(RjT, Ri), mat_ident()));
        Vec3 inner = mat_vec(RjT, mat_vec(Ri, tic) + Pi - Pj) - tic;
        Mat3 right = mat_add(mat_add(mat_scale(mat_mul(tmp_r, skew(pci)), -1.0), skew(mat_vec(tmp_r, pci))), skew(mat_vec(ricT, inner)));
        reduce_into(left, right, J + 28);
    }
    // inverse depth :126-130
    {
        Vec3 v = mat_vec(tmp_r, pts_i_td);
        double s = -1.0 / (inv_dep * inv_dep);
        J[42] = (red[0] * v.x + red[1] * v.y + red[2] * v.z) * s;
        J[43] = (red[3] * v.x + red[4] * v.y + red[5] * v.z) * s;
    }
    // td :131-136
    if (use_td) {
        Vec3 v = mat_vec(tmp_r, vel_i);
        J[44] = (red[0] * v.x + red[1] * v.y + red[2] * v.z) / inv_dep * -1.0 + sqrt_info * vel_j.x;
        J[45] = (red[3] * v.x + red[4] * v.y + red[5] * v.z) / inv_dep * -1.0 + sqrt_info * vel_j.y;
    } else {
        J[44] = J[45] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// A8 MarginalizationFactor::Evaluate  marginalization_factor.cpp:352-400
// params[i] -> global-size block i; J (if non-null): blocks consecutive, block i is n x gsize_i row-major
// ------------------------------------------------------------------------------------------------
static inline int gsize_of(int kind) { return kind == VIL_BLK_POSE || kind == VIL_BLK_EX ? 7 : (kind == VIL_BLK_SPEEDBIAS ? 9 : 1); }
static inline int lsize_of(int kind) { return kind == VIL_BLK_POSE || kind == VIL_BLK_EX ? 6 : (kind == VIL_BLK_SPEEDBIAS ? 9 : 1); }

void prior_dx(const vil_prior& pr, const double* const* params, double* dx) {
    int xoff = 0;
    for (int b = 0; b < pr.nblk; ++b) {
        const int gs = gsize_of(pr.blk_kind[b]);
        const double* x = params[b];
        const double* x0 = pr.x0 + xoff;
        const int idx = pr.blk_col[b];
        if (gs != 7) {
            for (int k = 0; k < gs; ++k) dx[idx + k] = x[k] - x0[k];
        } else {
            for (int k = 0; k < 3; ++k) dx[idx + k] = x[k] - x0[k];
            Quat q0 = {x0[6], x0[3], x0[4], x0[5]}, q = {x[6], x[3], x[4], x[5]};
            Quat dq = qmul(qinv(q0), q);
            double sgn = (dq.w >= 0) ? 2.0 : -2.0;  // :376-380 (positify is a no-op, utility.h:37-44)
            dx[idx + 3] = sgn * dq.x; dx[idx + 4] = sgn * dq.y; dx[idx + 5] = sgn * dq.z;
        }
        xoff += gs;
    }
}

void prior_evaluate(const vil_prior& pr, const double* const* params, double* r, double* J) {
    const int n = pr.n;
    std::vector<double> dx(n, 0.0);
    prior_dx(pr, params, dx.data());
    for (int i = 0; i < n; ++i) {
        double s = pr.r0[i];
        for (int k = 0; k < n; ++k) s += pr.J0[(size_t)k * n + i] * dx[k];  // J0 column-major
        r[i] = s;
    }
    if (!J) return;
    size_t off = 0;
    for (int b = 0; b < pr.nblk; ++b) {
        const int gs = gsize_of(pr.blk_kind[b]), ls = lsize_of(pr.blk_kind[b]);
        const int idx = pr.blk_col[b];
        for (int i = 0; i < n; ++i) for (int c2 = 0; c2 < gs; ++c2)
            J[off + (size_t)i * gs + c2] = (c2 < ls) ? pr.J0[(size_t)(idx + c2) * n + i] : 0.0;
        off += (size_t)n * gs;
    }
}

// ------------------------------------------------------------------------------------------------
// A11 LidarICPConstraint_b::operator()  lidar_backend.h:107-169, AutoDiff<3;7,7,7,7>
// ------------------------------------------------------------------------------------------------
template <class T>
static void icp_functor(const double* c, const T* A, const T* B, const T* C, const T* Dp, T* res) {
    const double ta = c[0], tb = c[1], tc = c[2], td = c[3], ti = c[4], tj = c[5];
    Qt<T> Qa{A[6], A[3], A[4], A[5]}, Qb{B[6], B[3], B[4], B[5]}, Qc{C[6], C[3], C[4], C[5]}, Qd{Dp[6], Dp[3], Dp[4], Dp[5]};
    V3<T> Pa{A[0], A[1], A[2]}, Pb{B[0], B[1], B[2]}, Pc{C[0], C[1], C[2]}, Pd{Dp[0], Dp[1], Dp[2]};
    const double t_i = (ti - ta) / (tb - ta);
    const double t_j = (tj - tc) / (td - tc);
    Qt<T> Qi = qslerp(Qa, t_i, Qb);
    Qt<T> Qj = qslerp(Qc, t_j, Qd);
    // Qi.normalized(); Qj.normalized();  -- results discarded in the reference (:125-126)
    // Pi = Pa + (Pb - Pa) / T(tb - ta) * T(ti - ta)
    V3<T> dab = Pb - Pa, dcd = Pd - Pc;
    V3<T> Pi{Pa.x + dab.x / T(tb - ta) * T(ti - ta), Pa.y + dab.y / T(tb - ta) * T(ti - ta), Pa.z + dab.z / T(tb - ta) * T(ti - ta)};
    V3<T> Pj{Pc.x + dcd.x / T(td - tc) * T(tj - tc), Pc.y + dcd.y / T(td - tc) * T(tj - tc), Pc.z + dcd.z / T(td - tc) * T(tj - tc)};
    Qt<T> temQ = qmul(qinv(Qj), Qi);
    V3<T> temPIJ = qrot(qinv(Qi), Pj - Pi);
    V3<T> PIJ{T(c[6]), T(c[7]), T(c[8])};
    V3<T> RES = qrot(temQ, PIJ - temPIJ);
    res[0] = RES.x * T(c[9]);
    res[1] = T(0.0);
    res[2] = RES.z * T(c[9]);
}

void icp_evaluate(const double* c, const double* pa, const double* pb, const double* pc, const double* pd,
                  double* r, double* J) {
    typedef Jet<28> JT;
    JT A[7], B[7], C[7], Dd[7], res[3];
    for (int k = 0; k < 7; ++k) { A[k] = JT(pa[k], k); B[k] = JT(pb[k], 7 + k); C[k] = JT(pc[k], 14 + k); Dd[k] = JT(pd[k], 21 + k); }
    icp_functor<JT>(c, A, B, C, Dd, res);
    for (int i = 0; i < 3; ++i) r[i] = res[i].a;
    if (!J) return;
    for (int b = 0; b < 4; ++b) for (int i = 0; i < 3; ++i) for (int k = 0; k < 7; ++k) J[b * 21 + i * 7 + k] = res[i].v[b * 7 + k];
}

// ------------------------------------------------------------------------------------------------
// A12 LPSConstraint::operator()  lidar_backend.h:45-80, AutoDiff<3;7,7>
// ------------------------------------------------------------------------------------------------
template <class T>
static void lps_functor(const double* c, const T* A, const T* B, T* res) {
    const double tl = c[0], tr = c[1], tk = c[2];
    Qt<T> Qa{A[6], A[3], A[4], A[5]}, Qb{B[6], B[3], B[4], B[5]};
    const double t_i = (tk - tl) / (tr - tl);
    Qt<T> Qi = qslerp(Qa, t_i, Qb);
    Qt<T> Q1{T(c[6]), T(c[3]), T(c[4]), T(c[5])};
    Qt<T> Q12 = qmul(qinv(Qi), Q1);
    res[0] = T(2.0) * Q12.x / T(0.01);
    res[1] = T(2.0) * Q12.y / T(0.01);
    res[2] = T(2.0) * Q12.z / T(0.01);
}

void lps_evaluate(const double* c, const double* pa, const double* pb, double* r, double* J) {
    typedef Jet<14> JT;
    JT A[7], B[7], res[3];
    for (int k = 0; k < 7; ++k) { A[k] = JT(pa[k], k); B[k] = JT(pb[k], 7 + k); }
    lps_functor<JT>(c, A, B, res);
    for (int i = 0; i < 3; ++i) r[i] = res[i].a;
    if (!J) return;
    for (int b = 0; b < 2; ++b) for (int i = 0; i < 3; ++i) for (int k = 0; k < 7; ++k) J[b * 21 + i * 7 + k] = res[i].v[b * 7 + k];
}

// ------------------------------------------------------------------------------------------------
// A14 LiDAR point factors, WINDOW-POSE FORM (a build definition):
//   p_b = RLB^T (p_l - TLB),  p_w = Q_k p_b + P_k,  then the lidarFactor.hpp residual on p_w
//   edge  (lidarFactor.hpp:12-55, s = 1):  r = ((p_w - a) x (p_w - b)) / |a - b|
//   plane (lidarFactor.hpp:106-138):       r = n . p_w + d
// Jacobian w.r.t. the right-multiplicative pose tangent, 7th column 0 (like the analytic factors).
// ------------------------------------------------------------------------------------------------
static inline Vec3 lidar_to_body(const double* q_lb, const double* t_lb, const Vec3& pl) {
    Quat qlb = {q_lb[3], q_lb[0], q_lb[1], q_lb[2]};
    return qrot(qinv(qlb), pl - vec_from(t_lb));
}

void edge_evaluate(const double* c, const double* q_lb, const double* t_lb, const double* pose, double* r, double* J) {
    const Vec3 pb = lidar_to_body(q_lb, t_lb, vec_from(c));
    const Vec3 a = vec_from(c + 3), b = vec_from(c + 6);
    const Quat Q = quat_from_block(pose);
    const Vec3 pw = qrot(Q, pb) + vec_from(pose);
    const Vec3 nu = cross(pw - a, pw - b);
    const Vec3 de = a - b;
    const double den = std::sqrt(dot(de, de));
    r[0] = nu.x / den; r[1] = nu.y / den; r[2] = nu.z / den;
    if (!J) return;
    const Mat3 dr_dpw = mat_scale(skew(de), -1.0 / den);
    const Mat3 dpw_dth = mat_scale(mat_mul(quat_R(Q), skew(pb)), -1.0);
    const Mat3 right = mat_mul(dr_dpw, dpw_dth);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { J[7 * i + j] = dr_dpw(i, j); J[7 * i + 3 + j] = right(i, j); }
        J[7 * i + 6] = 0.0;
    }
}

void plane_evaluate(const double* c, const double* q_lb, const double* t_lb, const double* pose, double* r, double* J) {
    const Vec3 pb = lidar_to_body(q_lb, t_lb, vec_from(c));
    const Vec3 n = vec_from(c + 3);
    const Quat Q = quat_from_block(pose);
    const Vec3 pw = qrot(Q, pb) + vec_from(pose);
    r[0] = dot(n, pw) + c[6];
    if (!J) return;
    const Mat3 dpw_dth = mat_scale(mat_mul(quat_R(Q), skew(pb)), -1.0);
    J[0] = n.x; J[1] = n.y; J[2] = n.z;
    for (int j = 0; j < 3; ++j) J[3 + j] = n.x * dpw_dth(0, j) + n.y * dpw_dth(1, j) + n.z * dpw_dth(2, j);
    J[6] = 0.0;
}

// LidarPlaneFactor (lidarFactor.hpp:57-104), s = 1: the constructor normalises (j - l) x (j - m); residual (lp - j) . ljm_norm --
// a plane-normal factor with n = ljm_norm, d = -n . j.  Dead code in the reference (localMapping.cpp:748-765 commented out).
void plane3_evaluate(const double* c, const double* q_lb, const double* t_lb, const double* pose, double* r, double* J) {
    const Vec3 pj = vec_from(c + 3), pl = vec_from(c + 6), pm = vec_from(c + 9);
    Vec3 n = cross(pj - pl, pj - pm);
    const double inv = 1.0 / std::sqrt(dot(n, n));
    n = n * inv;
    const double c7[7] = {c[0], c[1], c[2], n.x, n.y, n.z, -dot(n, pj)};
    plane_evaluate(c7, q_lb, t_lb, pose, r, J);
}
// LidarDistanceFactor (:141-172): r = point_w - closed_point.  Dead code in the reference (localMapping.cpp:668-685).
void distance_evaluate(const double* c, const double* q_lb, const double* t_lb, const double* pose, double* r, double* J) {
    const Vec3 pb = lidar_to_body(q_lb, t_lb, vec_from(c));
    const Quat Q = quat_from_block(pose);
    const Vec3 pw = qrot(Q, pb) + vec_from(pose);
    r[0] = pw.x - c[3]; r[1] = pw.y - c[4]; r[2] = pw.z - c[5];
    if (!J) return;
    const Mat3 dpw_dth = mat_scale(mat_mul(quat_R(Q), skew(pb)), -1.0);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { J[7 * i + j] = i == j ? 1.0 : 0.0; J[7 * i + 3 + j] = dpw_dth(i, j); }
        J[7 * i + 6] = 0.0;
    }
}

// Literal lidarFactor.hpp functors on the LiDAR->world transform (q_wl, t_wl), used by the tests
// to check that the window-pose form above reproduces the reference residuals.
void edge_residual_ref(const double* cp, const double* a3, const double* b3, const double* q_wl_xyzw, const double* t_wl, double* r) {
    Quat q = {q_wl_xyzw[3], q_wl_xyzw[0], q_wl_xyzw[1], q_wl_xyzw[2]};
    Quat ident = {1, 0, 0, 0};
    q = qslerp(ident, 1.0, q);  // s = 1.0 (localMapping.cpp:664)
    Vec3 lp = qrot(q, vec_from(cp)) + vec_from(t_wl);
    Vec3 nu = cross(lp - vec_from(a3), lp - vec_from(b3));
    Vec3 de = vec_from(a3) - vec_from(b3);
    double den = std::sqrt(dot(de, de));
    r[0] = nu.x / den; r[1] = nu.y / den; r[2] = nu.z / den;
}
void plane_residual_ref(const double* cp, const double* n3, double d, const double* q_wl_xyzw, const double* t_wl, double* r) {
    Quat q = {q_wl_xyzw[3], q_wl_xyzw[0], q_wl_xyzw[1], q_wl_xyzw[2]};
    Vec3 pw = qrot(q, vec_from(cp)) + vec_from(t_wl);
    r[0] = dot(vec_from(n3), pw) + d;
}

// ------------------------------------------------------------------------------------------------
// A9 robust-loss corrector  marginalization_factor.cpp:37-67 (mirror of ceres::internal::Corrector)
// rho[0..2] = rho(s), rho'(s), rho''(s);  ceres CauchyLoss / HuberLoss definitions.
// ------------------------------------------------------------------------------------------------
void loss_evaluate(int kind, double a, double s, double rho[3]) {
    if (kind == VIL_LOSS_CAUCHY) {
        const double b = a * a, cc = 1.0 / b;
        const double sum = 1.0 + s * cc, inv = 1.0 / sum;
        rho[0] = b * std::log(sum); rho[1] = inv > 2.2250738585072014e-308 ? inv : 2.2250738585072014e-308; rho[2] = -cc * (inv * inv);
    } else if (kind == VIL_LOSS_HUBER) {
        const double b = a * a;
        if (s > b) {
            const double rr = std::sqrt(s);
            rho[0] = 2.0 * a * rr - b; rho[1] = a / rr > 2.2250738585072014e-308 ? a / rr : 2.2250738585072014e-308; rho[2] = -rho[1] / (2.0 * s);
        } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
    } else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}

// corrects r (nr) and a row-major nr x nc Jacobian in place; returns rho(s)
double apply_corrector(int kind, double a, int nr, double* r, int nblocks, double* const* Jb, const int* ncols) {
    double sq = 0; for (int i = 0; i < nr; ++i) sq += r[i] * r[i];
    if (kind == VIL_LOSS_NONE) return sq;
    double rho[3];
    loss_evaluate(kind, a, sq, rho);
    const double sqrt_rho1 = std::sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
    else {
        const double Dd = 1.0 + 2.0 * sq * rho[2] / rho[1];
        const double alpha = 1.0 - std::sqrt(Dd);
        residual_scaling = sqrt_rho1 / (1 - alpha);
        alpha_sq_norm = alpha / sq;
    }
    for (int b = 0; b < nblocks; ++b) {
        double* Jm = Jb[b];
        if (!Jm) continue;
        const int nc = ncols[b];
        if (alpha_sq_norm != 0.0) {
            std::vector<double> rtJ(nc, 0.0);
            for (int i = 0; i < nr; ++i) for (int c2 = 0; c2 < nc; ++c2) rtJ[c2] += r[i] * Jm[i * nc + c2];
            for (int i = 0; i < nr; ++i) for (int c2 = 0; c2 < nc; ++c2) Jm[i * nc + c2] = sqrt_rho1 * (Jm[i * nc + c2] - alpha_sq_norm * r[i] * rtJ[c2]);
        } else {
            for (int i = 0; i < nr * nc; ++i) Jm[i] *= sqrt_rho1;
        }
    }
    for (int i = 0; i < nr; ++i) r[i] *= residual_scaling;
    return rho[0];
}

// ------------------------------------------------------------------------------------------------
// A5 IntegrationBase::{propagate, midPointIntegration}  integration_base.h:54-158
// state: delta_p(3) delta_q(xyzw,4) delta_v(3) | jacobian 15x15 | covariance 15x15 (row-major) | sum_dt
// ------------------------------------------------------------------------------------------------
void preint_init(Preint& s, const double* acc0, const double* gyr0, const double* ba, const double* bg, const double noise4[4]) {
    std::memset(&s, 0, sizeof s);
    s.dq[3] = 1.0;
    for (int i = 0; i < 15; ++i) s.jac[i * 15 + i] = 1.0;
    for (int k = 0; k < 3; ++k) { s.acc0[k] = acc0[k]; s.gyr0[k] = gyr0[k]; s.ba[k] = ba[k]; s.bg[k] = bg[k]; }
    for (int k = 0; k < 4; ++k) s.noise[k] = noise4[k];  // ACC_N GYR_N ACC_W GYR_W
}

void preint_push(Preint& s, double dt, const double* acc1, const double* gyr1) {
    const Vec3 a0 = vec_from(s.acc0), g0 = vec_from(s.gyr0), a1 = vec_from(acc1), g1 = vec_from(gyr1);
    const Vec3 ba = vec_from(s.ba), bg = vec_from(s.bg);
    const Quat dq = {s.dq[3], s.dq[0], s.dq[1], s.dq[2]};
    const Vec3 dp = vec_from(s.dp), dv = vec_from(s.dv);
    // :63-70
    const Vec3 un_acc_0 = qrot(dq, a0 - ba);
    const Vec3 un_gyr = (g0 + g1) * 0.5 - bg;
    const Quat rq = qmul(dq, Quat{1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2});
    const Vec3 un_acc_1 = qrot(rq, a1 - ba);
    const Vec3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
    const Vec3 rp = dp + dv * dt + un_acc * (0.5 * dt * dt);
    const Vec3 rv = dv + un_acc * dt;
    // :75-125
    const Vec3 w_x = (g0 + g1) * 0.5 - bg, a_0_x = a0 - ba, a_1_x = a1 - ba;
    const Mat3 R_w_x = skew(w_x), R_a_0_x = skew(a_0_x), R_a_1_x = skew(a_1_x);
    const Mat3 Rd = quat_R(dq), Rr = quat_R(rq), I3 = mat_ident();
    double F[225] = {0.0}, V[15 * 18] = {0.0};
    auto putF = [&](int r0, int c0, const Mat3& m) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F[(r0 + i) * 15 + c0 + j] = m(i, j); };
    auto putV = [&](int r0, int c0, const Mat3& m) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[(r0 + i) * 18 + c0 + j] = m(i, j); };
    const Mat3 ImW = mat_sub(I3, mat_scale(R_w_x, dt));
    putF(0, 0, I3);
    putF(0, 3, mat_add(mat_scale(mat_mul(Rd, R_a_0_x), -0.25 * dt * dt), mat_scale(mat_mul(mat_mul(Rr, R_a_1_x), ImW), -0.25 * dt * dt)));
    putF(0, 6, mat_scale(I3, dt));
    putF(0, 9, mat_scale(mat_add(Rd, Rr), -0.25 * dt * dt));
    putF(0, 12, mat_scale(mat_mul(Rr, R_a_1_x), -0.25 * dt * dt * -dt));
    putF(3, 3, ImW);
    putF(3, 12, mat_scale(I3, -dt));
    putF(6, 3, mat_add(mat_scale(mat_mul(Rd, R_a_0_x), -0.5 * dt), mat_scale(mat_mul(mat_mul(Rr, R_a_1_x), ImW), -0.5 * dt)));
    putF(6, 6, I3);
    putF(6, 9, mat_scale(mat_add(Rd, Rr), -0.5 * dt));
    putF(6, 12, mat_scale(mat_mul(Rr, R_a_1_x), -0.5 * dt * -dt));
    putF(9, 9, I3);
    putF(12, 12, I3);
    const Mat3 V03 = mat_scale(mat_mul(Rr, R_a_1_x), 0.25 * -1.0 * dt * dt * 0.5 * dt);
    const Mat3 V63 = mat_scale(mat_mul(Rr, R_a_1_x), 0.5 * -1.0 * dt * 0.5 * dt);
    putV(0, 0, mat_scale(Rd, 0.25 * dt * dt));
    putV(0, 3, V03);
    putV(0, 6, mat_scale(Rr, 0.25 * dt * dt));
    putV(0, 9, V03);
    putV(3, 3, mat_scale(I3, 0.5 * dt));
    putV(3, 9, mat_scale(I3, 0.5 * dt));
    putV(6, 0, mat_scale(Rd, 0.5 * dt));
    putV(6, 3, V63);
    putV(6, 6, mat_scale(Rr, 0.5 * dt));
    putV(6, 9, V63);
    putV(9, 12, mat_scale(I3, dt));
    putV(12, 15, mat_scale(I3, dt));
    // jacobian = F * jacobian ; covariance = F cov F^T + V noise V^T
    double nj[225], FC[225], nc[225];
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) {
        double s1 = 0, s2 = 0;
        for (int k = 0; k < 15; ++k) { s1 += F[i * 15 + k] * s.jac[k * 15 + j]; s2 += F[i * 15 + k] * s.cov[k * 15 + j]; }
        nj[i * 15 + j] = s1; FC[i * 15 + j] = s2;
    }
    double nd[18];
    for (int k = 0; k < 3; ++k) {
        nd[k] = s.noise[0] * s.noise[0]; nd[3 + k] = s.noise[1] * s.noise[1]; nd[6 + k] = s.noise[0] * s.noise[0];
        nd[9 + k] = s.noise[1] * s.noise[1]; nd[12 + k] = s.noise[2] * s.noise[2]; nd[15 + k] = s.noise[3] * s.noise[3];
    }
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) {
        double s1 = 0;
        for (int k = 0; k < 15; ++k) s1 += FC[i * 15 + k] * F[j * 15 + k];
        for (int k = 0; k < 18; ++k) s1 += V[i * 18 + k] * nd[k] * V[j * 18 + k];
        nc[i * 15 + j] = s1;
    }
    std::memcpy(s.jac, nj, sizeof s.jac);
    std::memcpy(s.cov, nc, sizeof s.cov);
    // :147-157
    s.dp[0] = rp.x; s.dp[1] = rp.y; s.dp[2] = rp.z;
    Quat rn = quat_normalized(rq);
    s.dq[0] = rn.x; s.dq[1] = rn.y; s.dq[2] = rn.z; s.dq[3] = rn.w;
    s.dv[0] = rv.x; s.dv[1] = rv.y; s.dv[2] = rv.z;
    s.sum_dt += dt;
    for (int k = 0; k < 3; ++k) { s.acc0[k] = acc1[k]; s.gyr0[k] = gyr1[k]; }
}

// pack into the 287-double constant record of vilsolve.h
void preint_pack(const Preint& s, double* c) {
    for (int k = 0; k < 3; ++k) { c[k] = s.dp[k]; c[7 + k] = s.dv[k]; c[10 + k] = s.ba[k]; c[13 + k] = s.bg[k]; }
    for (int k = 0; k < 4; ++k) c[3 + k] = s.dq[k];
    c[16] = s.sum_dt;
    auto blk = [&](int r0, int c0, double* out) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out[3 * i + j] = s.jac[(r0 + i) * 15 + c0 + j]; };
    blk(0, 9, c + 17); blk(0, 12, c + 26); blk(3, 12, c + 35); blk(6, 9, c + 44); blk(6, 12, c + 53);
    std::memcpy(c + 62, s.cov, sizeof s.cov);
}

}  // namespace orc
