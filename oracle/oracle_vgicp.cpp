// ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product path).
// PINNED by the reference's own known-answer test: the vendored fast_gicp ships src/test/gicp_test.cpp with two real scans
// and their relative pose (fast_gicp-master.zip: data/251370668.pcd, data/251371071.pcd, data/relative.txt); the restatement
// passes it forward and backward at the test's tolerances (0.05 m, 1 degree, converged) --
// tests/golden/vgicp/fast_gicp_kat.npz, tests/test_oracle_vgicp.py::test_reference_known_answer.  fast_gicp itself cannot be
// built here (Eigen / PCL absent), so parity finer than that tolerance rests on the numpy / finite-difference pins.
//
// CPU restatement of the voxelised GICP registration vendored in the reference under
// vils_estimator/src/lidar_functions/fast_gicp (third party: SMRT-AIST fast_gicp, unpinned snapshot):
//   GaussianVoxelMap / AdditiveGaussianVoxel      include/fast_gicp/gicp/fast_vgicp_voxel.hpp:107-170
//   FastVGICP::update_correspondences / linearize / compute_error   gicp/impl/fast_vgicp_impl.hpp:73-196
//   LsqRegistration::computeTransformation / is_converged / step_gn / step_lm   gicp/impl/lsq_registration_impl.hpp:48-165
//   so3_exp                                        so3/so3.hpp:53-77
//   FastGICP::calculate_covariances (k nearest neighbours, PLANE regularisation)   gicp/impl/fast_gicp_impl.hpp:241-300
// Same sequential accumulation order as the reference with one thread.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../include/vilvgicp.h"

namespace {

struct Voxel { int num = 0; double mean[3] = {0, 0, 0}; double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; };
struct Key { int x, y, z; bool operator==(const Key& o) const { return x == o.x && y == o.y && z == o.z; } };
struct KeyHash { size_t operator()(const Key& k) const { uint64_t h = (uint64_t)(uint32_t)k.x * 0x9E3779B97F4A7C15ull; h ^= (uint64_t)(uint32_t)k.y * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2); h ^= (uint64_t)(uint32_t)k.z * 0x165667B19E3779F9ull + (h << 6) + (h >> 2); return (size_t)h; } };

struct Corr { int src; const Voxel* vox; double M[9]; };

struct Ctx {
    double res = 1.0;
    std::unordered_map<Key, Voxel, KeyHash> vox;
    std::vector<float> sxyz; std::vector<double> scov;
    std::vector<Corr> corr;
};

// (x / resolution - 0.5).floor()   fast_vgicp_voxel.hpp:161-163
inline Key voxel_coord(const double* p, double res) { return {(int)std::floor(p[0] / res - 0.5), (int)std::floor(p[1] / res - 0.5), (int)std::floor(p[2] / res - 0.5)}; }

inline void offsets_of(int mode, std::vector<Key>& off) {      // fast_vgicp_voxel.hpp:10-43
    off.clear();
    if (mode == VGICP_DIRECT1) off.push_back({0, 0, 0});
    else if (mode == VGICP_DIRECT7) { off = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}}; }
    else for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < 3; ++k) off.push_back({i - 1, j - 1, k - 1});
}

inline bool inv3(const double* a, double* o) {
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
    const double id = 1.0 / det;
    o[0] = c0 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c1 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c2 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return std::isfinite(id);
}

// the 4x4 RCR of the reference has a zero 4th row / column except (3,3) = 1: its inverse is the 3x3 inverse, (3,3) then zeroed
void update_correspondences(Ctx& c, const double* T, int mode) {
    std::vector<Key> off; offsets_of(mode, off);
    c.corr.clear();
    const int n = (int)c.sxyz.size() / 3;
    for (int i = 0; i < n; ++i) {
        const double a[3] = {(double)c.sxyz[3 * i], (double)c.sxyz[3 * i + 1], (double)c.sxyz[3 * i + 2]};
        double ta[3];
        for (int r = 0; r < 3; ++r) ta[r] = T[4 * r] * a[0] + T[4 * r + 1] * a[1] + T[4 * r + 2] * a[2] + T[4 * r + 3];
        const Key k = voxel_coord(ta, c.res);
        for (const Key& o : off) {
            auto it = c.vox.find({k.x + o.x, k.y + o.y, k.z + o.z});
            if (it == c.vox.end()) continue;
            Corr cr; cr.src = i; cr.vox = &it->second;
            const double* ca = &c.scov[9 * i];
            double RC[9], RCR[9];
            for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) RC[3 * r + q] = T[4 * r] * ca[q] + T[4 * r + 1] * ca[3 + q] + T[4 * r + 2] * ca[6 + q];
            for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) RCR[3 * r + q] = cr.vox->cov[3 * r + q] + RC[3 * r] * T[4 * q] + RC[3 * r + 1] * T[4 * q + 1] + RC[3 * r + 2] * T[4 * q + 2];
            inv3(RCR, cr.M);
            c.corr.push_back(cr);
        }
    }
}

double accumulate(const Ctx& c, const double* T, double* H, double* b) {
    double sum = 0.0;
    if (H) { std::memset(H, 0, 36 * sizeof(double)); std::memset(b, 0, 6 * sizeof(double)); }
    for (const Corr& cr : c.corr) {
        const int i = cr.src;
        const double a[3] = {(double)c.sxyz[3 * i], (double)c.sxyz[3 * i + 1], (double)c.sxyz[3 * i + 2]};
        double ta[3], e[3], Me[3];
        for (int r = 0; r < 3; ++r) ta[r] = T[4 * r] * a[0] + T[4 * r + 1] * a[1] + T[4 * r + 2] * a[2] + T[4 * r + 3];
        for (int r = 0; r < 3; ++r) e[r] = cr.vox->mean[r] - ta[r];
        const double w = std::sqrt((double)cr.vox->num);
        for (int r = 0; r < 3; ++r) Me[r] = cr.M[3 * r] * e[0] + cr.M[3 * r + 1] * e[1] + cr.M[3 * r + 2] * e[2];
        sum += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);
        if (!H) continue;
        // J = [skew(ta) | -I]  (3 x 6)
        const double J[18] = {0, -ta[2], ta[1], -1, 0, 0, ta[2], 0, -ta[0], 0, -1, 0, -ta[1], ta[0], 0, 0, 0, -1};
        double MJ[18];
        for (int r = 0; r < 3; ++r) for (int q = 0; q < 6; ++q) MJ[6 * r + q] = cr.M[3 * r] * J[q] + cr.M[3 * r + 1] * J[6 + q] + cr.M[3 * r + 2] * J[12 + q];
        for (int p = 0; p < 6; ++p) {
            for (int q = 0; q < 6; ++q) H[6 * p + q] += w * (J[p] * MJ[q] + J[6 + p] * MJ[6 + q] + J[12 + p] * MJ[12 + q]);
            b[p] += w * (J[p] * Me[0] + J[6 + p] * Me[1] + J[12 + p] * Me[2]);
        }
    }
    return sum;
}

// symmetric positive definite 6x6 solve (the reference uses Eigen::LDLT; any accurate factorisation gives the same d)
bool solve6(const double* A, const double* rhs, double* x) {
    double L[36] = {0};
    for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
        if (!(d > 0.0)) return false;
        L[6 * j + j] = std::sqrt(d);
        for (int i = j + 1; i < 6; ++i) { double s = A[6 * i + j]; for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k]; L[6 * i + j] = s / L[6 * j + j]; }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) { double s = rhs[i]; for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k]; y[i] = s / L[6 * i + i]; }
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k]; x[i] = s / L[6 * i + i]; }
    return true;
}

void so3_exp_R(const double* w, double* R) {      // so3.hpp:53-77 + Quaternion::toRotationMatrix
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double im, re;
    if (t2 < 1e-10) { const double t4 = t2 * t2; im = 0.5 - 1.0 / 48.0 * t2 + 1.0 / 3840.0 * t4; re = 1.0 - 1.0 / 8.0 * t2 + 1.0 / 384.0 * t4; }
    else { const double t = std::sqrt(t2), h = 0.5 * t; im = std::sin(h) / t; re = std::cos(h); }
    const double qw = re, qx = im * w[0], qy = im * w[1], qz = im * w[2];
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz, twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy; R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx; R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

void compose(const double* d, const double* x0, double* xi) {      // delta * x0 with delta = [so3_exp(d[0:3]) | d[3:6]]
    double R[9]; so3_exp_R(d, R);
    for (int r = 0; r < 3; ++r) {
        for (int q = 0; q < 4; ++q) xi[4 * r + q] = R[3 * r] * x0[q] + R[3 * r + 1] * x0[4 + q] + R[3 * r + 2] * x0[8 + q];
        xi[4 * r + 3] += d[3 + r];
    }
    xi[12] = 0; xi[13] = 0; xi[14] = 0; xi[15] = 1;
}

bool is_converged(const double* d, double reps, double teps) {    // lsq_registration_impl.hpp:76-86
    double R[9]; so3_exp_R(d, R);
    double m = 0;
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) m = std::fmax(m, std::fabs(R[3 * r + q] - (r == q ? 1.0 : 0.0)) / reps);
    for (int r = 0; r < 3; ++r) m = std::fmax(m, std::fabs(d[3 + r]) / teps);
    return m < 1;
}

// smallest-eigenvalue eigenvector of a symmetric 3x3 (cyclic Jacobi); the PLANE regularisation U diag(1,1,1e-3) V^T of a
// symmetric PSD matrix is I - (1 - 1e-3) n n^T with n that vector (JacobiSVD of a symmetric matrix: U = V up to signs)
void smallest_eigvec3(const double* Cin, double* nrm) {
    double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(A, Cin, sizeof A);
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off <= 1e-40 * (A[0] * A[0] + A[4] * A[4] + A[8] * A[8]) || off == 0.0) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            const double apq = A[3 * p + q];
            if (apq == 0.0) continue;
            const double tau = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
            const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
            const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
            for (int k = 0; k < 3; ++k) { const double akp = A[3 * k + p], akq = A[3 * k + q]; A[3 * k + p] = c * akp - s * akq; A[3 * k + q] = s * akp + c * akq; }
            for (int k = 0; k < 3; ++k) { const double apk = A[3 * p + k], aqk = A[3 * q + k]; A[3 * p + k] = c * apk - s * aqk; A[3 * q + k] = s * apk + c * aqk; }
            for (int k = 0; k < 3; ++k) { const double vkp = V[3 * k + p], vkq = V[3 * k + q]; V[3 * k + p] = c * vkp - s * vkq; V[3 * k + q] = s * vkp + c * vkq; }
        }
    }
    int m = 0; if (A[4] < A[3 * m + m]) m = 1; if (A[8] < A[3 * m + m]) m = 2;
    for (int k = 0; k < 3; ++k) nrm[k] = V[3 * k + m];
}

__attribute__((optimize("fp-contract=off"))) void covariances(int n, const float* xyz, int k, double* out) {
    std::vector<std::pair<float, int>> d((size_t)n);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {                                  // float squared distances, like FLANN's L2_Simple
            const float dx = xyz[3 * i] - xyz[3 * j], dy = xyz[3 * i + 1] - xyz[3 * j + 1], dz = xyz[3 * i + 2] - xyz[3 * j + 2];
            d[j] = {dx * dx + dy * dy + dz * dz, j};
        }
        const int kk = std::min(k, n);
        std::partial_sort(d.begin(), d.begin() + kk, d.end());
        double mean[3] = {0, 0, 0}, C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = 0; q < kk; ++q) for (int r = 0; r < 3; ++r) mean[r] += (double)xyz[3 * d[q].second + r];
        for (int r = 0; r < 3; ++r) mean[r] /= k;                       // rowwise().mean() over the k columns
        for (int q = 0; q < kk; ++q) {
            const double c[3] = {(double)xyz[3 * d[q].second] - mean[0], (double)xyz[3 * d[q].second + 1] - mean[1], (double)xyz[3 * d[q].second + 2] - mean[2]};
            for (int r = 0; r < 3; ++r) for (int t = 0; t < 3; ++t) C[3 * r + t] += c[r] * c[t];
        }
        for (int r = 0; r < 9; ++r) C[r] /= k;
        double nv[3]; smallest_eigvec3(C, nv);
        for (int r = 0; r < 3; ++r) for (int t = 0; t < 3; ++t) out[9 * (size_t)i + 3 * r + t] = (r == t ? 1.0 : 0.0) - (1.0 - 1e-3) * nv[r] * nv[t];
    }
}

}  // namespace

extern "C" {

struct vgicp_ctx { Ctx c; };

int orc_vgicp_create(int32_t, vgicp_ctx** out) { *out = new vgicp_ctx(); return 0; }
void orc_vgicp_destroy(vgicp_ctx* c) { delete c; }
void orc_vgicp_default_options(vgicp_options* o) {
    o->neighbor_mode = VGICP_DIRECT1; o->optimizer = VGICP_LM; o->max_iterations = 64; o->lm_max_iterations = 10;
    o->rotation_epsilon = 2e-3; o->transformation_epsilon = 5e-4; o->lm_init_lambda_factor = 1e-9;
}
int orc_vgicp_covariances(vgicp_ctx*, int32_t n, const float* xyz, int32_t k, double* out) { covariances(n, xyz, k, out); return 0; }
int orc_vgicp_set_target(vgicp_ctx* c, int32_t n, const float* xyz, const double* cov9_in, double resolution) {
    std::vector<double> own; const double* cov9 = cov9_in;
    if (!cov9) { own.resize(9 * (size_t)n); covariances(n, xyz, 20, own.data()); cov9 = own.data(); }
    c->c.res = resolution; c->c.vox.clear(); c->c.corr.clear();
    for (int i = 0; i < n; ++i) {                                   // create_voxelmap + AdditiveGaussianVoxel::append
        const double p[3] = {(double)xyz[3 * i], (double)xyz[3 * i + 1], (double)xyz[3 * i + 2]};
        Voxel& v = c->c.vox[voxel_coord(p, resolution)];
        v.num++;
        for (int k = 0; k < 3; ++k) v.mean[k] += p[k];
        for (int k = 0; k < 9; ++k) v.cov[k] += cov9[9 * i + k];
    }
    for (auto& kv : c->c.vox) { Voxel& v = kv.second; for (int k = 0; k < 3; ++k) v.mean[k] /= v.num; for (int k = 0; k < 9; ++k) v.cov[k] /= v.num; }   // finalize
    return 0;
}
int orc_vgicp_set_source(vgicp_ctx* c, int32_t n, const float* xyz, const double* cov9_in) {
    std::vector<double> own; const double* cov9 = cov9_in;
    if (!cov9) { own.resize(9 * (size_t)n); covariances(n, xyz, 20, own.data()); cov9 = own.data(); }
    c->c.sxyz.assign(xyz, xyz + 3 * (size_t)n); c->c.scov.assign(cov9, cov9 + 9 * (size_t)n); c->c.corr.clear();
    return 0;
}
int orc_vgicp_linearize(vgicp_ctx* c, const double* T, int32_t mode, double* err, double* H, double* b, int32_t* n_corr) {
    update_correspondences(c->c, T, mode);
    *err = accumulate(c->c, T, (H && b) ? H : nullptr, b);
    if (n_corr) *n_corr = (int32_t)c->c.corr.size();
    return 0;
}
int orc_vgicp_compute_error(vgicp_ctx* c, const double* T, double* err) { *err = accumulate(c->c, T, nullptr, nullptr); return 0; }

int orc_vgicp_profile_enable(vgicp_ctx*, int32_t) { return 0; }
int orc_vgicp_profile_read(vgicp_ctx*, int64_t* n, double* ms) { *n = 0; *ms = 0.0; return 0; }

int orc_vgicp_align(vgicp_ctx* c, const double* guess, const vgicp_options* o, double* T_out, vgicp_summary* out) {
    double x0[16]; std::memcpy(x0, guess, sizeof x0);
    double lambda = -1.0;
    bool converged = false;
    std::memset(out, 0, sizeof *out);
    for (int k = 0; k < 36; ++k) out->final_hessian[k] = (k % 7 == 0) ? 1.0 : 0.0;
    int it = 0;
    for (; it < o->max_iterations && !converged; ++it) {
        double H[36], b[6], d[6], nb[6];
        int32_t nc = 0;
        double y0; orc_vgicp_linearize(c, x0, o->neighbor_mode, &y0, H, b, &nc);
        out->n_correspondences = nc; out->final_error = y0;
        for (int k = 0; k < 6; ++k) nb[k] = -b[k];
        bool stepped = false;
        if (o->optimizer == VGICP_GN) {
            if (!solve6(H, nb, d)) { out->lm_failed = 1; break; }
            double xi[16]; compose(d, x0, xi); std::memcpy(x0, xi, sizeof x0); std::memcpy(out->final_hessian, H, sizeof H); stepped = true;
        } else {
            if (lambda < 0.0) { double m = 0; for (int k = 0; k < 6; ++k) m = std::fmax(m, std::fabs(H[7 * k])); lambda = o->lm_init_lambda_factor * m; }
            double nu = 2.0;
            for (int i = 0; i < o->lm_max_iterations; ++i) {
                double Hl[36]; std::memcpy(Hl, H, sizeof H);
                for (int k = 0; k < 6; ++k) Hl[7 * k] += lambda;
                if (!solve6(Hl, nb, d)) { lambda = nu * lambda; nu = 2 * nu; continue; }
                double xi[16]; compose(d, x0, xi);
                double yi; orc_vgicp_compute_error(c, xi, &yi);
                double den = 0; for (int k = 0; k < 6; ++k) den += d[k] * (lambda * d[k] - b[k]);
                const double rho = (y0 - yi) / den;
                if (rho < 0) {
                    if (is_converged(d, o->rotation_epsilon, o->transformation_epsilon)) { stepped = true; break; }
                    lambda = nu * lambda; nu = 2 * nu; continue;
                }
                std::memcpy(x0, xi, sizeof x0);
                const double f = 1 - std::pow(2 * rho - 1, 3);
                lambda = lambda * std::fmax(1.0 / 3.0, f);
                std::memcpy(out->final_hessian, H, sizeof H);
                stepped = true;
                break;
            }
        }
        if (!stepped) { out->lm_failed = 1; ++it; break; }
        converged = is_converged(d, o->rotation_epsilon, o->transformation_epsilon);
    }
    out->iterations = it; out->converged = converged ? 1 : 0;
    std::memcpy(T_out, x0, sizeof x0);
    return 0;
}

}  // extern "C"
