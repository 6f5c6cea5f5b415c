// ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product path).  PARITY UNPINNED: the
// reference ships no tests or golden vectors for lidar_mapping and cannot be built here (Eigen / Ceres / PCL / ROS absent).
//
// CPU restatement of the scan-to-map registration of lidar_mapping/src/localMapping.cpp:590-791:
//   pointAssociateToMap :170-179 ; corner points: 5 nearest map points, PCA line test :613-660 -> LidarEdgeFactor(cp, a, b, 1.0)
//   surf points: 10 nearest re-ranked by |intensity difference| :688-703, plane fit + 0.2 m validity :705-741 -> LidarPlaneNormFactor
//   two rounds of association + solve (HuberLoss 0.1, max 4 iterations) :594-600, :766-777 -- the solve is the oracle's own
//   trust-region restatement (oracle_solver.cpp) on a one-pose window (see include/vilmap.h for the parameterisation note).
// Nearest-neighbour search: exhaustive by default (the checker), or an exact kd-tree (median split, 15-point leaves -- the
// structure pcl::KdTreeFLANN builds) selected by orc_vmap_set_search(ctx, 1) for the CPU baseline; both produce the float
// distances pcl would, tests/test_oracle_map.py holds them equal.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/vilmap.h"

extern "C" int orc_solve(const vil_problem* p, vil_state* s, const vil_options* o, vil_summary* sum);

namespace {

struct KdNode { int lo, hi, dim, left, right; float split; };
struct KdTree {
    std::vector<KdNode> nodes; std::vector<int> idx; const float* pts = nullptr;
    int build_rec(int lo, int hi) {
        const int id = (int)nodes.size(); nodes.push_back({lo, hi, -1, -1, -1, 0.f});
        if (hi - lo <= 15) return id;
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = lo; i < hi; ++i) for (int d = 0; d < 3; ++d) { const float v = pts[4 * idx[i] + d]; mn[d] = std::min(mn[d], v); mx[d] = std::max(mx[d], v); }
        int dim = 0; if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1; if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
        const int mid = (lo + hi) / 2;
        std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, [&](int a, int b) { return pts[4 * a + dim] < pts[4 * b + dim]; });
        const float split = pts[4 * idx[mid] + dim];
        const int l = build_rec(lo, mid), r = build_rec(mid, hi);
        nodes[id].dim = dim; nodes[id].split = split; nodes[id].left = l; nodes[id].right = r;
        return id;
    }
    void build(const std::vector<float>& map) {
        pts = map.data(); const int n = (int)map.size() / 4;
        idx.resize((size_t)n); for (int i = 0; i < n; ++i) idx[i] = i;
        nodes.clear(); nodes.reserve((size_t)(n / 4 + 8));
        if (n) build_rec(0, n);
    }
    // left subtree: coordinate <= split, right subtree: coordinate >= split (nth_element)
    __attribute__((optimize("fp-contract=off"))) void search(int id, const float* s, int k, std::pair<float, int>* best, int& cnt) const {
        const KdNode& nd = nodes[id];
        if (nd.dim < 0) {
            for (int i = nd.lo; i < nd.hi; ++i) {
                const int j = idx[i];
                const float dx = s[0] - pts[4 * j], dy = s[1] - pts[4 * j + 1], dz = s[2] - pts[4 * j + 2];
                const std::pair<float, int> c{dx * dx + dy * dy + dz * dz, j};
                if (cnt == k && !(c < best[k - 1])) continue;
                int pos = cnt < k ? cnt++ : k - 1;
                while (pos > 0 && c < best[pos - 1]) { best[pos] = best[pos - 1]; --pos; }
                best[pos] = c;
            }
            return;
        }
        const float diff = s[nd.dim] - nd.split;
        const int near = diff <= 0.f ? nd.left : nd.right, far = diff <= 0.f ? nd.right : nd.left;
        search(near, s, k, best, cnt);
        if (cnt < k || diff * diff <= best[k - 1].first) search(far, s, k, best, cnt);
    }
};
struct Ctx { std::vector<float> cmap, smap; int search = 0; KdTree kc, ks; };

void quat_to_R(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

void jacobi3(double* A, double* V) {
    V[0] = 1; V[1] = 0; V[2] = 0; V[3] = 0; V[4] = 1; V[5] = 0; V[6] = 0; V[7] = 0; V[8] = 1;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off <= 1e-40 * (A[0] * A[0] + A[4] * A[4] + A[8] * A[8]) || off == 0.0) break;
        for (int p = 0; p < 2; ++p) for (int q = p + 1; q < 3; ++q) {
            const double apq = A[3 * p + q];
            if (apq == 0.0) continue;
            const double tau = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
            const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
            const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
            for (int r = 0; r < 3; ++r) { const double akp = A[3 * r + p], akq = A[3 * r + q]; A[3 * r + p] = c * akp - s * akq; A[3 * r + q] = s * akp + c * akq; }
            for (int r = 0; r < 3; ++r) { const double apk = A[3 * p + r], aqk = A[3 * q + r]; A[3 * p + r] = c * apk - s * aqk; A[3 * q + r] = s * apk + c * aqk; }
            for (int r = 0; r < 3; ++r) { const double vkp = V[3 * r + p], vkq = V[3 * r + q]; V[3 * r + p] = c * vkp - s * vkq; V[3 * r + q] = s * vkp + c * vkq; }
        }
    }
}

// min |A x - b| for a 5 x 3 A (row-major in P, overwritten) by column-pivoted Householder QR -- what
// matA0.colPivHouseholderQr().solve(matB0) does (localMapping.cpp:716)
void qr_solve_5x3(double* P, double* b, double* x) {
    int perm[3] = {0, 1, 2};
    for (int k = 0; k < 3; ++k) {
        int pv = k; double best = -1.0;
        for (int c = k; c < 3; ++c) { double s = 0; for (int r = k; r < 5; ++r) s += P[3 * r + c] * P[3 * r + c]; if (s > best) { best = s; pv = c; } }
        if (pv != k) { for (int r = 0; r < 5; ++r) { const double tmp = P[3 * r + k]; P[3 * r + k] = P[3 * r + pv]; P[3 * r + pv] = tmp; } const int tp = perm[k]; perm[k] = perm[pv]; perm[pv] = tp; }
        const double nrm = std::sqrt(best), akk = P[3 * k + k], alpha = akk > 0 ? -nrm : nrm;
        const double v0 = akk - alpha;
        double vv = v0 * v0;
        for (int r = k + 1; r < 5; ++r) vv += P[3 * r + k] * P[3 * r + k];
        if (vv > 0) {
            const double beta = 2.0 / vv;
            for (int c = k + 1; c < 3; ++c) {
                double s = v0 * P[3 * k + c];
                for (int r = k + 1; r < 5; ++r) s += P[3 * r + k] * P[3 * r + c];
                s *= beta;
                P[3 * k + c] -= s * v0;
                for (int r = k + 1; r < 5; ++r) P[3 * r + c] -= s * P[3 * r + k];
            }
            double s = v0 * b[k];
            for (int r = k + 1; r < 5; ++r) s += P[3 * r + k] * b[r];
            s *= beta;
            b[k] -= s * v0;
            for (int r = k + 1; r < 5; ++r) b[r] -= s * P[3 * r + k];
        }
        P[3 * k + k] = alpha;
    }
    double y[3];
    y[2] = b[2] / P[8];
    y[1] = (b[1] - P[5] * y[2]) / P[4];
    y[0] = (b[0] - P[1] * y[1] - P[2] * y[2]) / P[0];
    for (int k = 0; k < 3; ++k) x[perm[k]] = y[k];
}

// k nearest map points of (sx, sy, sz): (float squared distance, index) ascending, ties by index
__attribute__((optimize("fp-contract=off"))) void knn(const std::vector<float>& map, float sx, float sy, float sz, int k, std::vector<std::pair<float, int>>& out) {
    const int n = (int)map.size() / 4;
    out.resize((size_t)n);
    for (int j = 0; j < n; ++j) {
        const float dx = sx - map[4 * j], dy = sy - map[4 * j + 1], dz = sz - map[4 * j + 2];
        out[j] = {dx * dx + dy * dy + dz * dz, j};
    }
    std::partial_sort(out.begin(), out.begin() + std::min(k, n), out.end());
    out.resize((size_t)std::min(k, n));
}

void knn_any(const Ctx& c, bool surf, float sx, float sy, float sz, int k, std::vector<std::pair<float, int>>& out) {
    if (!c.search) { knn(surf ? c.smap : c.cmap, sx, sy, sz, k, out); return; }
    const KdTree& t = surf ? c.ks : c.kc;
    out.resize((size_t)k); int cnt = 0; const float s[3] = {sx, sy, sz};
    t.search(0, s, k, out.data(), cnt);
    out.resize((size_t)cnt);
}

void associate(const Ctx& c, int nc, const float* corner, int ns, const float* surf, const double* q, const double* t,
               int32_t* n_edge, double* edge9, int32_t* n_plane, double* plane7) {
    double R[9]; quat_to_R(q, R);
    auto to_map = [&](const float* p, float* s) {
        const double x = p[0], y = p[1], z = p[2];
        s[0] = (float)(R[0] * x + R[1] * y + R[2] * z + t[0]); s[1] = (float)(R[3] * x + R[4] * y + R[5] * z + t[1]); s[2] = (float)(R[6] * x + R[7] * y + R[8] * z + t[2]);
    };
    std::vector<std::pair<float, int>> nb;
    *n_edge = 0; *n_plane = 0;
    for (int i = 0; i < nc && (int)c.cmap.size() / 4 >= 5; ++i) {
        float s[3]; to_map(corner + 4 * i, s);
        knn_any(c, false, s[0], s[1], s[2], 5, nb);
        if (!(nb[4].first < 1.0f)) continue;
        double cen[3] = {0, 0, 0}, A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, V[9];
        for (int j = 0; j < 5; ++j) for (int r = 0; r < 3; ++r) cen[r] += (double)c.cmap[4 * nb[j].second + r];
        for (int r = 0; r < 3; ++r) cen[r] /= 5.0;
        for (int j = 0; j < 5; ++j) {
            const double d[3] = {(double)c.cmap[4 * nb[j].second] - cen[0], (double)c.cmap[4 * nb[j].second + 1] - cen[1], (double)c.cmap[4 * nb[j].second + 2] - cen[2]};
            for (int r = 0; r < 3; ++r) for (int u = 0; u < 3; ++u) A[3 * r + u] += d[r] * d[u];
        }
        jacobi3(A, V);
        int m2 = 0; if (A[4] > A[0]) m2 = 1; if (A[8] > A[4 * m2]) m2 = 2;
        const double l2 = A[4 * m2], l1 = std::fmax(A[4 * ((m2 + 1) % 3)], A[4 * ((m2 + 2) % 3)]);
        if (!(l2 > 3.0 * l1)) continue;
        double* o = edge9 + 9 * (size_t)(*n_edge);
        for (int r = 0; r < 3; ++r) { o[r] = corner[4 * i + r]; o[3 + r] = 0.1 * V[3 * r + m2] + cen[r]; o[6 + r] = -0.1 * V[3 * r + m2] + cen[r]; }
        ++*n_edge;
    }
    for (int i = 0; i < ns && (int)c.smap.size() / 4 >= 10; ++i) {
        float s[3]; to_map(surf + 4 * i, s);
        knn_any(c, true, s[0], s[1], s[2], 10, nb);
        const float d5 = nb[4].first;
        std::vector<std::pair<float, int>> rk(10);
        for (int m = 0; m < 10; ++m) rk[m] = {std::fabs(c.smap[4 * nb[m].second + 3] - surf[4 * i + 3]), nb[m].second};
        std::sort(rk.begin(), rk.end());
        if (!(d5 < 1.0f)) continue;
        double P[15], Q[15], rhs[5] = {-1, -1, -1, -1, -1}, nv[3] = {0, 0, 0};
        for (int j = 0; j < 5; ++j) for (int u = 0; u < 3; ++u) P[3 * j + u] = (double)c.smap[4 * rk[j].second + u];
        std::memcpy(Q, P, sizeof Q);
        qr_solve_5x3(Q, rhs, nv);
        double nx = nv[0], ny = nv[1], nz = nv[2];
        const double nn = std::sqrt(nx * nx + ny * ny + nz * nz), d = 1.0 / nn;
        nx /= nn; ny /= nn; nz /= nn;
        bool ok = std::isfinite(d) && std::isfinite(nx);
        for (int j = 0; j < 5; ++j) ok = ok && !(std::fabs(nx * P[3 * j] + ny * P[3 * j + 1] + nz * P[3 * j + 2] + d) > 0.2);
        if (!ok) continue;
        double* o = plane7 + 7 * (size_t)(*n_plane);
        for (int u = 0; u < 3; ++u) o[u] = surf[4 * i + u];
        o[3] = nx; o[4] = ny; o[5] = nz; o[6] = d;
        ++*n_plane;
    }
}

}  // namespace

extern "C" {

struct vmap_ctx { Ctx c; };

int orc_vmap_create(int32_t, vmap_ctx** out) { *out = new vmap_ctx(); return 0; }
void orc_vmap_destroy(vmap_ctx* c) { delete c; }
int orc_vmap_set_map(vmap_ctx* c, int32_t nc, const float* corner, int32_t ns, const float* surf) {
    c->c.cmap.assign(corner, corner + 4 * (size_t)nc); c->c.smap.assign(surf, surf + 4 * (size_t)ns);
    if (c->c.search) { c->c.kc.build(c->c.cmap); c->c.ks.build(c->c.smap); }          // kdtree->setInputCloud, localMapping.cpp:590-591
    return 0;
}
int orc_vmap_set_search(vmap_ctx* c, int32_t kd) {
    c->c.search = kd != 0;
    if (c->c.search) { c->c.kc.build(c->c.cmap); c->c.ks.build(c->c.smap); }
    return 0;
}
int orc_vmap_associate(vmap_ctx* c, int32_t nc, const float* corner, int32_t ns, const float* surf, const double* q, const double* t,
                       int32_t* n_edge, double* edge9, int32_t* n_plane, double* plane7) {
    associate(c->c, nc, corner, ns, surf, q, t, n_edge, edge9, n_plane, plane7);
    return 0;
}
int orc_vmap_align(vmap_ctx* c, vil_ctx*, int32_t nc, const float* corner, int32_t ns, const float* surf, double* q, double* t, const vil_options* opts, vmap_summary* out) {
    std::memset(out, 0, sizeof *out);
    if (!((int)c->c.cmap.size() / 4 > 10 && (int)c->c.smap.size() / 4 > 50)) return 0;
    std::vector<double> edge(9 * (size_t)std::max(1, nc)), plane(7 * (size_t)std::max(1, ns));
    for (int round = 0; round < 2; ++round) {
        int32_t ne = 0, np = 0;
        associate(c->c, nc, corner, ns, surf, q, t, &ne, edge.data(), &np, plane.data());
        std::vector<int32_t> epose((size_t)std::max(1, ne), 0), ppose((size_t)std::max(1, np), 0);
        vil_problem p; std::memset(&p, 0, sizeof p);
        uint8_t pose_const = 0, sb_const = 1;
        p.K = 1; p.L = 0; p.pose_const = &pose_const; p.sb_const = &sb_const; p.ex_const = 1; p.td_const = 1; p.use_td = 0;
        p.n_edge = ne; p.edge_pose = epose.data(); p.edge_const = edge.data();
        p.n_plane = np; p.plane_pose = ppose.data(); p.plane_const = plane.data();
        p.q_lb[3] = 1.0; p.sqrt_info_px = 230.0; p.G[2] = 9.8;
        double pose[7] = {t[0], t[1], t[2], q[0], q[1], q[2], q[3]}, sb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ex[7] = {0, 0, 0, 0, 0, 0, 1}, td = 0.0, lam = 0.0;
        vil_state s; std::memset(&s, 0, sizeof s);
        s.K = 1; s.L = 0; s.pose = pose; s.speedbias = sb; s.ex_pose = ex; s.td = &td; s.inv_depth = &lam;
        vil_summary sum;
        const int st = orc_solve(&p, &s, opts, &sum);
        if (st != 0) return st;
        t[0] = pose[0]; t[1] = pose[1]; t[2] = pose[2]; q[0] = pose[3]; q[1] = pose[4]; q[2] = pose[5]; q[3] = pose[6];
        out->t_prepare_ms += sum.t_prepare_ms; out->t_solve_ms += sum.t_solve_ms + sum.t_readback_ms;
        out->rounds = round + 1; out->n_edge = ne; out->n_plane = np; out->iterations = sum.iterations; out->initial_cost = sum.initial_cost; out->final_cost = sum.final_cost;
    }
    return 0;
}

int orc_vmap_profile_enable(vmap_ctx*, int32_t) { return 0; }
int orc_vmap_profile_read(vmap_ctx*, int64_t* n, double* ms) { n[0] = n[1] = 0; ms[0] = ms[1] = 0.0; return 0; }

}  // extern "C"
