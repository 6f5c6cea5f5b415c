// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the arithmetic of mVIL-Fusion's sliding-window backend
// (vils_estimator/src/factor/*, lidar_backend.h, lidar_mapping/src/lidarFactor.hpp).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// PARITY UNPINNED: the reference ships no tests / golden vectors and cannot be compiled in this
// image (Eigen, Ceres, ROS absent), so this restatement is pinned only by the in-tree formulas it
// cites, by the reference's own finite-difference recipe (projection_factor.cpp:176-224) and by
// algebraic identities (tests/test_oracle_*.py).
//
// Small fixed-size linear algebra + quaternion helpers that follow Eigen's semantics where the
// reference depends on them (utility/utility.h and Eigen::Quaternion).
#pragma once
#include <cmath>
#include <cstring>

namespace orc {

// ---- forward-mode dual number (what ceres::Jet does for the two AutoDiff factors) -------------
template <int N>
struct Jet {
    double a;
    double v[N];
    Jet() : a(0) { for (int i = 0; i < N; ++i) v[i] = 0; }
    Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; }
    Jet(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; v[k] = 1.0; }
};
template <int N> inline Jet<N> operator+(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a + y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] + y.v[i]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a - y.a; for (int i = 0; i < N; ++i) r.v[i] = x.v[i] - y.v[i]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& x) { Jet<N> r; r.a = -x.a; for (int i = 0; i < N; ++i) r.v[i] = -x.v[i]; return r; }
template <int N> inline Jet<N> operator*(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; r.a = x.a * y.a; for (int i = 0; i < N; ++i) r.v[i] = x.a * y.v[i] + x.v[i] * y.a; return r; }
template <int N> inline Jet<N> operator/(const Jet<N>& x, const Jet<N>& y) { Jet<N> r; double inv = 1.0 / y.a; r.a = x.a * inv; for (int i = 0; i < N; ++i) r.v[i] = (x.v[i] - r.a * y.v[i]) * inv; return r; }
template <int N> inline Jet<N> sin(const Jet<N>& x) { Jet<N> r; r.a = std::sin(x.a); double c = std::cos(x.a); for (int i = 0; i < N; ++i) r.v[i] = c * x.v[i]; return r; }
template <int N> inline Jet<N> cos(const Jet<N>& x) { Jet<N> r; r.a = std::cos(x.a); double s = -std::sin(x.a); for (int i = 0; i < N; ++i) r.v[i] = s * x.v[i]; return r; }
template <int N> inline Jet<N> acos(const Jet<N>& x) { Jet<N> r; r.a = std::acos(x.a); double d = -1.0 / std::sqrt(1.0 - x.a * x.a); for (int i = 0; i < N; ++i) r.v[i] = d * x.v[i]; return r; }
template <int N> inline Jet<N> abs(const Jet<N>& x) { return x.a < 0.0 ? -x : x; }
template <int N> inline Jet<N> sqrt(const Jet<N>& x) { Jet<N> r; r.a = std::sqrt(x.a); double d = 0.5 / r.a; for (int i = 0; i < N; ++i) r.v[i] = d * x.v[i]; return r; }
inline double scalar(double x) { return x; }
template <int N> inline double scalar(const Jet<N>& x) { return x.a; }
using std::sin; using std::cos; using std::acos; using std::abs; using std::sqrt;

// ---- generic 3-vectors / quaternions (w,x,y,z members; Hamilton; Eigen semantics) -------------
template <class T> struct V3 { T x, y, z; };
template <class T> inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> inline V3<T> operator*(const V3<T>& a, const T& s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> inline V3<T> cross(const V3<T>& a, const V3<T>& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <class T> inline T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

template <class T> struct Qt { T w, x, y, z; };
// Eigen::Quaternion operator* (Hamilton product)
template <class T> inline Qt<T> qmul(const Qt<T>& a, const Qt<T>& b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
            a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
// Eigen::Quaternion::inverse(): conjugate / squaredNorm
template <class T> inline Qt<T> qinv(const Qt<T>& q) {
    T n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
// Eigen::Quaternion * Vector3 (_transformVector): v + w*uv + vec x uv, uv = 2 vec x v
template <class T> inline V3<T> qrot(const Qt<T>& q, const V3<T>& v) {
    V3<T> u{q.x, q.y, q.z};
    V3<T> uv = cross(u, v);
    uv = uv + uv;
    return v + uv * q.w + cross(u, uv);
}
// Eigen::QuaternionBase::slerp (Eigen 3.3 Geometry/Quaternion.h)
template <class T> inline Qt<T> qslerp(const Qt<T>& a, double t, const Qt<T>& b) {
    const double one = 1.0 - 2.220446049250313e-16;
    T d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
    T absD = abs(d);
    T s0, s1;
    if (scalar(absD) >= one) {
        s0 = T(1.0 - t);
        s1 = T(t);
    } else {
        T theta = acos(absD);
        T sinTheta = sin(theta);
        s0 = sin(theta * T(1.0 - t)) / sinTheta;
        s1 = sin(theta * T(t)) / sinTheta;
    }
    if (scalar(d) < 0.0) s1 = -s1;
    return {s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}

// ---- plain double helpers ---------------------------------------------------------------------
typedef V3<double> Vec3;
typedef Qt<double> Quat;
struct Mat3 { double m[9]; double& operator()(int r, int c) { return m[3 * r + c]; } double operator()(int r, int c) const { return m[3 * r + c]; } };

inline Quat quat_from_block(const double* p7) { return {p7[6], p7[3], p7[4], p7[5]}; }  // [.. qx qy qz qw]
inline Vec3 vec_from(const double* p) { return {p[0], p[1], p[2]}; }
inline Mat3 mat_zero() { Mat3 r; std::memset(r.m, 0, sizeof r.m); return r; }
inline Mat3 mat_ident() { Mat3 r = mat_zero(); r(0, 0) = r(1, 1) = r(2, 2) = 1; return r; }
inline Mat3 mat_mul(const Mat3& a, const Mat3& b) { Mat3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += a(i, k) * b(k, j); r(i, j) = s; } return r; }
inline Mat3 mat_T(const Mat3& a) { Mat3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = a(j, i); return r; }
inline Mat3 mat_add(const Mat3& a, const Mat3& b) { Mat3 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] + b.m[i]; return r; }
inline Mat3 mat_sub(const Mat3& a, const Mat3& b) { Mat3 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] - b.m[i]; return r; }
inline Mat3 mat_scale(const Mat3& a, double s) { Mat3 r; for (int i = 0; i < 9; ++i) r.m[i] = a.m[i] * s; return r; }
inline Vec3 mat_vec(const Mat3& a, const Vec3& v) { return {a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z, a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z, a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z}; }
// utility.h:27-34
inline Mat3 skew(const Vec3& q) { Mat3 r; r(0, 0) = 0; r(0, 1) = -q.z; r(0, 2) = q.y; r(1, 0) = q.z; r(1, 1) = 0; r(1, 2) = -q.x; r(2, 0) = -q.y; r(2, 1) = q.x; r(2, 2) = 0; return r; }
// Eigen::Quaternion::toRotationMatrix()
inline Mat3 quat_R(const Quat& q) {
    Mat3 r;
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    r(0, 0) = 1 - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy;
    r(1, 0) = txy + twz; r(1, 1) = 1 - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1 - (txx + tyy);
    return r;
}
inline Quat quat_normalized(const Quat& q) { double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z); return {q.w / n, q.x / n, q.y / n, q.z / n}; }
// utility.h:12-24 (first order, NOT normalised)
inline Quat deltaQ(const Vec3& th) { return {1.0, th.x / 2, th.y / 2, th.z / 2}; }
// bottom-right 3x3 of utility.h:47-55 Qleft(q):  w I + [v]x ;  of :57-64 Qright(q):  w I - [v]x
inline Mat3 qleft33(const Quat& q) { Mat3 r = skew({q.x, q.y, q.z}); r(0, 0) += q.w; r(1, 1) += q.w; r(2, 2) += q.w; return r; }
inline Mat3 qright33(const Quat& q) { Mat3 r = mat_scale(skew({q.x, q.y, q.z}), -1.0); r(0, 0) += q.w; r(1, 1) += q.w; r(2, 2) += q.w; return r; }
// full 4x4 (w x y z order) as in utility.h, used for the product Qleft*Qright in imu_factor.h:100
inline void qleft44(const Quat& q, double* m) {
    m[0] = q.w; m[1] = -q.x; m[2] = -q.y; m[3] = -q.z;
    Mat3 b = qleft33(q);
    const double v[3] = {q.x, q.y, q.z};
    for (int i = 0; i < 3; ++i) { m[4 * (i + 1)] = v[i]; for (int j = 0; j < 3; ++j) m[4 * (i + 1) + j + 1] = b(i, j); }
}
inline void qright44(const Quat& q, double* m) {
    m[0] = q.w; m[1] = -q.x; m[2] = -q.y; m[3] = -q.z;
    Mat3 b = qright33(q);
    const double v[3] = {q.x, q.y, q.z};
    for (int i = 0; i < 3; ++i) { m[4 * (i + 1)] = v[i]; for (int j = 0; j < 3; ++j) m[4 * (i + 1) + j + 1] = b(i, j); }
}
// utility.h:66-82 (degrees)
inline Vec3 R2ypr(const Mat3& R) {
    const double n0 = R(0, 0), n1 = R(1, 0), n2 = R(2, 0);
    const double o0 = R(0, 1), o1 = R(1, 1);
    const double a0 = R(0, 2), a1 = R(1, 2);
    double y = std::atan2(n1, n0);
    double p = std::atan2(-n2, n0 * std::cos(y) + n1 * std::sin(y));
    double r = std::atan2(a0 * std::sin(y) - a1 * std::cos(y), -o0 * std::sin(y) + o1 * std::cos(y));
    return {y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0};
}
// utility.h:84-112
inline Mat3 ypr2R(const Vec3& ypr) {
    double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
    Mat3 Rz = mat_zero(), Ry = mat_zero(), Rx = mat_zero();
    Rz(0, 0) = std::cos(y); Rz(0, 1) = -std::sin(y); Rz(1, 0) = std::sin(y); Rz(1, 1) = std::cos(y); Rz(2, 2) = 1;
    Ry(0, 0) = std::cos(p); Ry(0, 2) = std::sin(p); Ry(1, 1) = 1; Ry(2, 0) = -std::sin(p); Ry(2, 2) = std::cos(p);
    Rx(0, 0) = 1; Rx(1, 1) = std::cos(r); Rx(1, 2) = -std::sin(r); Rx(2, 1) = std::sin(r); Rx(2, 2) = std::cos(r);
    return mat_mul(mat_mul(Rz, Ry), Rx);
}
// Eigen: Quaterniond(Matrix3d) (Shepperd's method as in Eigen's quaternionbase_assign_impl)
inline Quat quat_from_R(const Mat3& R) {
    Quat q;
    double t = R(0, 0) + R(1, 1) + R(2, 2);
    if (t > 0) {
        t = std::sqrt(t + 1.0);
        q.w = 0.5 * t; t = 0.5 / t;
        q.x = (R(2, 1) - R(1, 2)) * t; q.y = (R(0, 2) - R(2, 0)) * t; q.z = (R(1, 0) - R(0, 1)) * t;
    } else {
        int i = 0;
        if (R(1, 1) > R(0, 0)) i = 1;
        if (R(2, 2) > R(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
        double v[3];
        v[i] = 0.5 * t; t = 0.5 / t;
        q.w = (R(k, j) - R(j, k)) * t;
        v[j] = (R(j, i) + R(i, j)) * t; v[k] = (R(k, i) + R(i, k)) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}

}  // namespace orc
