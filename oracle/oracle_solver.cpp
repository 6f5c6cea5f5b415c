// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of what estimator.cpp:1126-1419 asks ceres to do:
//   trust-region (TRADITIONAL_DOGLEG) non-linear least squares with a DENSE_SCHUR linear solver,
//   Jacobi scaling, robust-loss corrector (marginalization_factor.cpp:37-67), local
//   parameterisation "first 6 of 7 columns" (pose_local_parameterization.cpp:20-27).
// Ceres itself is a third-party dependency absent from /root/reference (version unpinned,
// find_package(Ceres) vils_estimator/CMakeLists.txt:31, API <= 2.1); the loop below restates its
// published algorithm (trust_region_minimizer.cc / dogleg_strategy.cc).
// The trajectory is therefore NOT pinned by the reference -- only the factor arithmetic and the
// optimum are.
#include <chrono>
#include <cstdio>

#include <memory>
#include <thread>
#include "oracle.hpp"

namespace orc {

#ifdef ORC_TIMERS
double g_tm[8];
struct Tm { int k; std::chrono::steady_clock::time_point t0; Tm(int k_) : k(k_), t0(std::chrono::steady_clock::now()) {} ~Tm() { g_tm[k] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } };
#define ORC_TM(k) Tm tm_##k(k)
#else
#define ORC_TM(k) do {} while (0)
#endif

struct WState {
    int K, L;
    std::vector<double> pose, sb, lam;
    double ex[7];
    double td;
    void from(const vil_state* s) {
        K = s->K; L = s->L;
        pose.assign(s->pose, s->pose + 7 * K);
        sb.assign(s->speedbias, s->speedbias + 9 * K);
        lam.assign(s->inv_depth, s->inv_depth + L);
        std::memcpy(ex, s->ex_pose, sizeof ex);
        td = s->td[0];
    }
    void to(vil_state* s) const {
        std::copy(pose.begin(), pose.end(), s->pose);
        std::copy(sb.begin(), sb.end(), s->speedbias);
        std::copy(lam.begin(), lam.end(), s->inv_depth);
        std::memcpy(s->ex_pose, ex, sizeof ex);
        s->td[0] = td;
    }
};

struct System {
    Layout lay;
    std::vector<double> Hcc, bc, hll, bl, E;
    std::vector<uint8_t> cconst, lconst;   // 1 => constant (not optimised)
    double cost;
    double *Ep, *hp, *bp;                  // landmark rows the accumulation writes to (own storage, or a parent's: see below)
    explicit System(int K, int L) : lay(K, L), Hcc((size_t)lay.D * lay.D), bc(lay.D), hll(L), bl(L), E((size_t)L * lay.D), cconst(lay.D), lconst(L), cost(0) {
        Ep = E.data(); hp = hll.data(); bp = bl.data();
    }
    // worker-thread accumulator of the all-cores variant: private camera block, landmark rows SHARED with `parent`
    // (each worker owns a disjoint set of landmarks, like the reference's marginalisation threads own disjoint factors)
    System(const System& parent, int) : lay(parent.lay), Hcc((size_t)lay.D * lay.D), bc(lay.D), cost(0) { Ep = parent.Ep; hp = parent.hp; bp = parent.bp; }
    void zero() {
        std::fill(Hcc.begin(), Hcc.end(), 0.0); std::fill(bc.begin(), bc.end(), 0.0);
        std::fill(hll.begin(), hll.end(), 0.0); std::fill(bl.begin(), bl.end(), 0.0);
        std::fill(E.begin(), E.end(), 0.0); cost = 0;
    }
};

// block descriptor for accumulation: col >= 0 camera column, col == -1 constant, col <= -2 landmark (-2-l)
struct Blk { int col; int n; int ld; const double* J; };

static void accumulate(System& sys, int nr, const double* r, int nb, const Blk* b) {
    const int D = sys.lay.D;
    for (int a = 0; a < nb; ++a) {
        if (b[a].col == -1) continue;
        if (b[a].col <= -2) {
            const int l = -2 - b[a].col;
            double h = 0, g = 0;
            for (int k = 0; k < nr; ++k) { double j = b[a].J[k * b[a].ld]; h += j * j; g += j * r[k]; }
            sys.hp[l] += h; sys.bp[l] += g;
            for (int c = 0; c < nb; ++c) {
                if (b[c].col < 0) continue;
                for (int jj = 0; jj < b[c].n; ++jj) {
                    double s = 0;
                    for (int k = 0; k < nr; ++k) s += b[a].J[k * b[a].ld] * b[c].J[k * b[c].ld + jj];
                    sys.Ep[(size_t)l * D + b[c].col + jj] += s;
                }
            }
            continue;
        }
        for (int i = 0; i < b[a].n; ++i) {
            double g = 0;
            for (int k = 0; k < nr; ++k) g += b[a].J[k * b[a].ld + i] * r[k];
            sys.bc[b[a].col + i] += g;
        }
        for (int c = 0; c < nb; ++c) {
            if (b[c].col < 0) continue;
            // full (both triangles); blocks with equal col handled once each ordered pair
            for (int i = 0; i < b[a].n; ++i) for (int jj = 0; jj < b[c].n; ++jj) {
                double s = 0;
                for (int k = 0; k < nr; ++k) s += b[a].J[k * b[a].ld + i] * b[c].J[k * b[c].ld + jj];
                sys.Hcc[(size_t)(b[a].col + i) * D + b[c].col + jj] += s;
            }
        }
    }
}

// d(q (x) [1, dth/2]) / d dth in (x y z w) order, 4x3:  for the mathematically-correct ICP/LPS option
static void tangent_fix(const double* pose, int nr, double* J7 /* nr x 7 row-major, in place */) {
    const double x = pose[3], y = pose[4], z = pose[5], w = pose[6];
    const double P[12] = {0.5 * w, -0.5 * z, 0.5 * y,
                          0.5 * z, 0.5 * w, -0.5 * x,
                          -0.5 * y, 0.5 * x, 0.5 * w,
                          -0.5 * x, -0.5 * y, -0.5 * z};
    for (int i = 0; i < nr; ++i) {
        double q4[4] = {J7[7 * i + 3], J7[7 * i + 4], J7[7 * i + 5], J7[7 * i + 6]};
        for (int j = 0; j < 3; ++j) J7[7 * i + 3 + j] = q4[0] * P[j] + q4[1] * P[3 + j] + q4[2] * P[6 + j] + q4[3] * P[9 + j];
        J7[7 * i + 6] = 0.0;
    }
}

struct Ctx {
    const vil_problem* p;
    vil_options o;
    Layout lay;
    Ctx(const vil_problem* p_, const vil_options* o_) : p(p_), o(*o_), lay(p_->K, p_->L) {}
    int col_pose(int k) const { return (p->pose_const && p->pose_const[k]) ? -1 : lay.pose(k); }
    int col_sb(int k) const { return (p->sb_const && p->sb_const[k]) ? -1 : lay.sb(k); }
    int col_ex() const { return p->ex_const ? -1 : lay.ex(); }
    int col_td() const { return (p->td_const || !p->use_td) ? -1 : lay.td(); }
    int col_lm(int l) const { return (p->lm_const && p->lm_const[l]) ? -1 : -2 - l; }
};

// ---- bulk factor classes (visual, LiDAR points): worker t of T ---------------------------------------------------------
static double bulk_part(const Ctx& c, const WState& x, System* sys, int t, int T) {
    const vil_problem* p = c.p;
    double cost = 0;
    // visual  estimator.cpp:1189-1242
    for (int f = 0; f < p->n_vis; ++f) {
        const int i = p->vis_i[f], j = p->vis_j[f], l = p->vis_l[f];
        if (T > 1 && l % T != t) continue;                 // landmark ownership: rows of E / hll / bl are written by one worker only
        double r[2], J[VIL_VIS_NJ];
        visual_evaluate(p->vis_const + (size_t)f * VIL_VIS_CONST, p->sqrt_info_px, p->tr_over_row, p->use_td,
                        &x.pose[7 * i], &x.pose[7 * j], x.ex, x.lam[l], x.td, r, sys ? J : nullptr);
        double* Jb[5] = {J, J + 14, J + 28, J + 42, J + 44};
        const int nc[5] = {7, 7, 7, 1, 1};
        double rho0;
        if (sys) rho0 = apply_corrector(c.o.visual_loss, c.o.visual_loss_scale, 2, r, 5, Jb, nc);
        else { double sq = r[0] * r[0] + r[1] * r[1]; double rho[3]; loss_evaluate(c.o.visual_loss, c.o.visual_loss_scale, sq, rho); rho0 = rho[0]; }
        cost += 0.5 * rho0;
        if (sys) {
            Blk b[5] = {{c.col_pose(i), 6, 7, J}, {c.col_pose(j), 6, 7, J + 14}, {c.col_ex(), 6, 7, J + 28}, {c.col_lm(l), 1, 1, J + 42}, {c.col_td(), 1, 1, J + 44}};
            accumulate(*sys, 2, r, 5, b);
        }
    }
    // LiDAR edge / plane points (extended mode), Huber(0.1) as localMapping.cpp:597
    for (int f = (int)((int64_t)p->n_edge * t / T); f < (int)((int64_t)p->n_edge * (t + 1) / T); ++f) {
        const int k = p->edge_pose[f];
        double r[3], J[21];
        edge_evaluate(p->edge_const + (size_t)f * VIL_EDGE_CONST, p->q_lb, p->t_lb, &x.pose[7 * k], r, sys ? J : nullptr);
        double* Jb[1] = {J}; const int nc[1] = {7};
        double rho0;
        if (sys) rho0 = apply_corrector(c.o.lidar_loss, c.o.lidar_loss_scale, 3, r, 1, Jb, nc);
        else { double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2]; double rho[3]; loss_evaluate(c.o.lidar_loss, c.o.lidar_loss_scale, sq, rho); rho0 = rho[0]; }
        cost += 0.5 * rho0;
        if (sys) { Blk b[1] = {{c.col_pose(k), 6, 7, J}}; accumulate(*sys, 3, r, 1, b); }
    }
    for (int f = (int)((int64_t)p->n_plane * t / T); f < (int)((int64_t)p->n_plane * (t + 1) / T); ++f) {
        const int k = p->plane_pose[f];
        double r[1], J[7];
        plane_evaluate(p->plane_const + (size_t)f * VIL_PLANE_CONST, p->q_lb, p->t_lb, &x.pose[7 * k], r, sys ? J : nullptr);
        double* Jb[1] = {J}; const int nc[1] = {7};
        double rho0;
        if (sys) rho0 = apply_corrector(c.o.lidar_loss, c.o.lidar_loss_scale, 1, r, 1, Jb, nc);
        else { double sq = r[0] * r[0]; double rho[3]; loss_evaluate(c.o.lidar_loss, c.o.lidar_loss_scale, sq, rho); rho0 = rho[0]; }
        cost += 0.5 * rho0;
        if (sys) { Blk b[1] = {{c.col_pose(k), 6, 7, J}}; accumulate(*sys, 1, r, 1, b); }
    }
    return cost;
}

// Number of threads of the "all cores" CPU-baseline variant.  1 = the reference's configuration (ceres::Solve
// with the default num_threads = 1, estimator.cpp:1400-1411) and the ONLY setting used for parity.
static int g_threads = 1;
void set_threads(int n) { g_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }

static double bulk(const Ctx& c, const WState& x, System* sys) {
    const int T = g_threads;
    if (T == 1) return bulk_part(c, x, sys, 0, 1);
    std::vector<double> costs(T, 0.0);
    std::vector<std::unique_ptr<System>> priv(T);
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) {
        if (sys) priv[t].reset(new System(*sys, 0));
        th.emplace_back([&, t]() { costs[t] = bulk_part(c, x, sys ? priv[t].get() : nullptr, t, T); });
    }
    costs[0] = bulk_part(c, x, sys, 0, T);
    for (auto& q : th) q.join();
    double cost = 0;
    for (int t = 0; t < T; ++t) cost += costs[t];
    if (sys) for (int t = 1; t < T; ++t) {
        for (size_t i = 0; i < sys->Hcc.size(); ++i) sys->Hcc[i] += priv[t]->Hcc[i];
        for (size_t i = 0; i < sys->bc.size(); ++i) sys->bc[i] += priv[t]->bc[i];
    }
    return cost;
}

// Evaluate every residual block at `x`; if sys != nullptr also build the (corrected) normal equations.
static double linearize(const Ctx& c, const WState& x, System* sys) {
    const vil_problem* p = c.p;
    double cost = 0;
    if (sys) { ORC_TM(0); sys->zero(); }
    // prior (no loss)  estimator.cpp:1171-1177
    if (p->prior.n > 0) {
        const vil_prior& pr = p->prior;
        std::vector<const double*> params(pr.nblk);
        for (int b = 0; b < pr.nblk; ++b) {
            switch (pr.blk_kind[b]) {
                case VIL_BLK_POSE: params[b] = &x.pose[7 * pr.blk_index[b]]; break;
                case VIL_BLK_SPEEDBIAS: params[b] = &x.sb[9 * pr.blk_index[b]]; break;
                case VIL_BLK_EX: params[b] = x.ex; break;
                default: params[b] = &x.td; break;
            }
        }
        const int n = pr.n;
        std::vector<double> r(n), dx(n);
        prior_dx(pr, params.data(), dx.data());
        for (int i = 0; i < n; ++i) { double s = pr.r0[i]; for (int k = 0; k < n; ++k) s += pr.J0[(size_t)k * n + i] * dx[k]; r[i] = s; }
        double sq = 0; for (int i = 0; i < n; ++i) sq += r[i] * r[i];
        cost += 0.5 * sq;
        if (sys) {
            // J block b = J0[:, col_b : col_b+local] (column-major J0 => column pointer, ld = 1 per row step n)
            // accumulate generically with row-major copies of the needed columns
            std::vector<double> Jrm((size_t)n * n);  // row-major n x n (prior-column order)
            for (int i = 0; i < n; ++i) for (int k = 0; k < n; ++k) Jrm[(size_t)i * n + k] = pr.J0[(size_t)k * n + i];
            std::vector<Blk> bl(pr.nblk);
            for (int b = 0; b < pr.nblk; ++b) {
                int col, nloc;
                switch (pr.blk_kind[b]) {
                    case VIL_BLK_POSE: col = c.col_pose(pr.blk_index[b]); nloc = 6; break;
                    case VIL_BLK_SPEEDBIAS: col = c.col_sb(pr.blk_index[b]); nloc = 9; break;
                    case VIL_BLK_EX: col = c.col_ex(); nloc = 6; break;
                    default: col = c.col_td(); nloc = 1; break;
                }
                bl[b] = {col, nloc, n, &Jrm[pr.blk_col[b]]};
            }
            accumulate(*sys, n, r.data(), pr.nblk, bl.data());
        }
    }
    // IMU (no loss)  estimator.cpp:1179-1186
    for (int f = 0; f < p->n_imu; ++f) {
        const double* cc = p->imu_const + (size_t)f * VIL_IMU_CONST;
        if (cc[16] > 10.0) continue;
        const int i = p->imu_i[f], j = p->imu_j[f];
        double r[15], J[VIL_IMU_NJ];
        imu_evaluate(cc, p->G, &x.pose[7 * i], &x.sb[9 * i], &x.pose[7 * j], &x.sb[9 * j], r, sys ? J : nullptr);
        double sq = 0; for (int k = 0; k < 15; ++k) sq += r[k] * r[k];
        cost += 0.5 * sq;
        if (sys) {
            Blk b[4] = {{c.col_pose(i), 6, 7, J}, {c.col_sb(i), 9, 9, J + 105}, {c.col_pose(j), 6, 7, J + 240}, {c.col_sb(j), 9, 9, J + 345}};
            accumulate(*sys, 15, r, 4, b);
        }
    }
    // ICP  estimator.cpp:1371-1396
    for (int f = 0; f < p->n_icp; ++f) {
        const int* id = p->icp_ids + 4 * f;
        double r[3], J[VIL_ICP_NJ];
        icp_evaluate(p->icp_const + (size_t)f * VIL_ICP_CONST, &x.pose[7 * id[0]], &x.pose[7 * id[1]], &x.pose[7 * id[2]], &x.pose[7 * id[3]], r, sys ? J : nullptr);
        if (sys && !c.o.autodiff_quirk) for (int b = 0; b < 4; ++b) tangent_fix(&x.pose[7 * id[b]], 3, J + 21 * b);
        double* Jb[4] = {J, J + 21, J + 42, J + 63};
        const int nc[4] = {7, 7, 7, 7};
        double rho0;
        if (sys) rho0 = apply_corrector(c.o.rel_loss, c.o.rel_loss_scale, 3, r, 4, Jb, nc);
        else { double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2]; double rho[3]; loss_evaluate(c.o.rel_loss, c.o.rel_loss_scale, sq, rho); rho0 = rho[0]; }
        cost += 0.5 * rho0;
        if (sys) {
            Blk b[4];
            for (int q = 0; q < 4; ++q) b[q] = {c.col_pose(id[q]), 6, 7, J + 21 * q};
            accumulate(*sys, 3, r, 4, b);
        }
    }
    // LPS  estimator.cpp:1298-1324
    for (int f = 0; f < p->n_lps; ++f) {
        const int* id = p->lps_ids + 2 * f;
        double r[3], J[VIL_LPS_NJ];
        lps_evaluate(p->lps_const + (size_t)f * VIL_LPS_CONST, &x.pose[7 * id[0]], &x.pose[7 * id[1]], r, sys ? J : nullptr);
        if (sys && !c.o.autodiff_quirk) for (int b = 0; b < 2; ++b) tangent_fix(&x.pose[7 * id[b]], 3, J + 21 * b);
        double* Jb[2] = {J, J + 21};
        const int nc[2] = {7, 7};
        double rho0;
        if (sys) rho0 = apply_corrector(c.o.rel_loss, c.o.rel_loss_scale, 3, r, 2, Jb, nc);
        else { double sq = r[0] * r[0] + r[1] * r[1] + r[2] * r[2]; double rho[3]; loss_evaluate(c.o.rel_loss, c.o.rel_loss_scale, sq, rho); rho0 = rho[0]; }
        cost += 0.5 * rho0;
        if (sys) {
            Blk b[2] = {{c.col_pose(id[0]), 6, 7, J}, {c.col_pose(id[1]), 6, 7, J + 21}};
            accumulate(*sys, 3, r, 2, b);
        }
    }
    { ORC_TM(1); cost += bulk(c, x, sys); }
    if (sys) {
        sys->cost = cost;
        const int D = c.lay.D;
        std::fill(sys->cconst.begin(), sys->cconst.end(), 0);
        for (int k = 0; k < p->K; ++k) {
            if (c.col_pose(k) < 0) for (int q = 0; q < 6; ++q) sys->cconst[c.lay.pose(k) + q] = 1;
            if (c.col_sb(k) < 0) for (int q = 0; q < 9; ++q) sys->cconst[c.lay.sb(k) + q] = 1;
        }
        if (c.col_ex() < 0) for (int q = 0; q < 6; ++q) sys->cconst[c.lay.ex() + q] = 1;
        if (c.col_td() < 0) sys->cconst[c.lay.td()] = 1;
        for (int l = 0; l < p->L; ++l) sys->lconst[l] = (c.col_lm(l) == -1);
        (void)D;
    }
    return cost;
}

// per-landmark list of camera columns it touches (anchor, observers, ex, td)
static void landmark_cols(const Ctx& c, std::vector<std::vector<int>>& cols) {
    const vil_problem* p = c.p;
    cols.assign(p->L, {});
    std::vector<std::vector<int>> frames(p->L);
    for (int f = 0; f < p->n_vis; ++f) {
        auto& v = frames[p->vis_l[f]];
        if (std::find(v.begin(), v.end(), p->vis_i[f]) == v.end()) v.push_back(p->vis_i[f]);
        if (std::find(v.begin(), v.end(), p->vis_j[f]) == v.end()) v.push_back(p->vis_j[f]);
    }
    for (int l = 0; l < p->L; ++l) {
        std::sort(frames[l].begin(), frames[l].end());
        for (int k : frames[l]) if (c.col_pose(k) >= 0) for (int q = 0; q < 6; ++q) cols[l].push_back(c.lay.pose(k) + q);
        if (!frames[l].empty()) {
            if (c.col_ex() >= 0) for (int q = 0; q < 6; ++q) cols[l].push_back(c.lay.ex() + q);
            if (c.col_td() >= 0) cols[l].push_back(c.lay.td());
        }
    }
}

// Plus():  pose_local_parameterization.cpp:3-18 for 7-blocks, plain addition otherwise
static void state_plus(const Ctx& c, const WState& x, const double* dc, const double* dl, WState& out) {
    out = x;
    auto pose_plus = [](const double* in, const double* d, double* o) {
        for (int k = 0; k < 3; ++k) o[k] = in[k] + d[k];
        Quat q = quat_normalized(qmul(quat_from_block(in), deltaQ({d[3], d[4], d[5]})));
        o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
    };
    for (int k = 0; k < x.K; ++k) {
        if (c.col_pose(k) >= 0) pose_plus(&x.pose[7 * k], dc + c.lay.pose(k), &out.pose[7 * k]);
        if (c.col_sb(k) >= 0) for (int q = 0; q < 9; ++q) out.sb[9 * k + q] = x.sb[9 * k + q] + dc[c.lay.sb(k) + q];
    }
    if (c.col_ex() >= 0) pose_plus(x.ex, dc + c.lay.ex(), out.ex);
    if (c.col_td() >= 0) out.td = x.td + dc[c.lay.td()];
    for (int l = 0; l < x.L; ++l) if (c.col_lm(l) != -1) out.lam[l] = x.lam[l] + dl[l];
}

static void free_norms(const Ctx& c, const WState& x, const WState& y, double* xnorm, double* stepnorm) {
    double xn = 0, sn = 0;
    auto acc = [&](const double* a, const double* b, int n) { for (int k = 0; k < n; ++k) { xn += a[k] * a[k]; sn += (a[k] - b[k]) * (a[k] - b[k]); } };
    for (int k = 0; k < x.K; ++k) {
        if (c.col_pose(k) >= 0) acc(&x.pose[7 * k], &y.pose[7 * k], 7);
        if (c.col_sb(k) >= 0) acc(&x.sb[9 * k], &y.sb[9 * k], 9);
    }
    if (c.col_ex() >= 0) acc(x.ex, y.ex, 7);
    if (c.col_td() >= 0) acc(&x.td, &y.td, 1);
    for (int l = 0; l < x.L; ++l) if (c.col_lm(l) != -1) acc(&x.lam[l], &y.lam[l], 1);
    *xnorm = std::sqrt(xn); *stepnorm = std::sqrt(sn);
}

// ---- the DENSE_SCHUR + dogleg machinery in one object -----------------------------------------
struct Dogleg {
    const Ctx& c;
    int D, L;
    std::vector<std::vector<int>> lcols;
    std::vector<double> Sc, Sl;          // Jacobi scaling (fixed after iteration 0)
    std::vector<double> dc, dl;          // diagonal_ (dogleg scaling), recomputed per linearisation
    std::vector<double> gradc, gradl;    // gradient_ in dogleg space
    std::vector<double> gnc, gnl;        // gauss_newton_step_ in dogleg space
    double alpha = 0, radius, mu, dogleg_step_norm = 0;
    bool reuse = false;
    Dogleg(const Ctx& c_) : c(c_), D(c_.lay.D), L(c_.lay.L), Sc(D, 1.0), Sl(L, 1.0), dc(D), dl(L), gradc(D), gradl(L), gnc(D), gnl(L) {
        landmark_cols(c, lcols);
        radius = c.o.initial_radius; mu = c.o.min_mu;
    }
    void init_scaling(const System& s) {
        if (!c.o.jacobi_scaling) return;
        for (int i = 0; i < D; ++i) Sc[i] = 1.0 / (1.0 + std::sqrt(s.Hcc[(size_t)i * D + i]));
        for (int l = 0; l < L; ++l) Sl[l] = 1.0 / (1.0 + std::sqrt(s.hll[l]));
    }
    // v^T H v over free parameters (H = J^T J of the corrected Jacobian), v in ORIGINAL coordinates
    double quad(const System& s, const double* vc, const double* vl) const {
        double q = 0;
        for (int i = 0; i < D; ++i) { if (s.cconst[i]) continue; double row = 0; const double* Hr = &s.Hcc[(size_t)i * D]; for (int j = 0; j < D; ++j) row += Hr[j] * vc[j]; q += vc[i] * row; }
        for (int l = 0; l < L; ++l) {
            if (s.lconst[l]) continue;
            double ev = 0; for (int col : lcols[l]) ev += s.E[(size_t)l * D + col] * vc[col];
            q += 2.0 * vl[l] * ev + s.hll[l] * vl[l] * vl[l];
        }
        return q;
    }
    // returns false if no valid GN step could be computed
    bool compute_system(const System& s) {
        ORC_TM(2);
        // diagonal_ = sqrt(clamp(colnorm^2 of the Jacobi-scaled Jacobian, 1e-6, 1e32))
        for (int i = 0; i < D; ++i) { double v = Sc[i] * Sc[i] * s.Hcc[(size_t)i * D + i]; dc[i] = std::sqrt(std::min(std::max(v, 1e-6), 1e32)); }
        for (int l = 0; l < L; ++l) { double v = Sl[l] * Sl[l] * s.hll[l]; dl[l] = std::sqrt(std::min(std::max(v, 1e-6), 1e32)); }
        // gradient_ = (J~^T r) / diagonal_
        for (int i = 0; i < D; ++i) gradc[i] = s.cconst[i] ? 0.0 : Sc[i] * s.bc[i] / dc[i];
        for (int l = 0; l < L; ++l) gradl[l] = s.lconst[l] ? 0.0 : Sl[l] * s.bl[l] / dl[l];
        // Cauchy point: alpha = |gradient_|^2 / |J~ (gradient_/diagonal_)|^2
        {
            std::vector<double> vc(D), vl(L);
            double g2 = 0;
            for (int i = 0; i < D; ++i) { vc[i] = Sc[i] * gradc[i] / dc[i]; g2 += gradc[i] * gradc[i]; }
            for (int l = 0; l < L; ++l) { vl[l] = Sl[l] * gradl[l] / dl[l]; g2 += gradl[l] * gradl[l]; }
            alpha = g2 / quad(s, vc.data(), vl.data());
        }
        // Gauss-Newton step through the Schur complement, mu retry loop (dogleg_strategy.cc)
        bool ok = false;
        std::vector<double> Sred((size_t)D * D), rhs(D), Lo((size_t)D * D), xc(D), xl(L);
        while (true) {
            for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) Sred[(size_t)i * D + j] = (s.cconst[i] || s.cconst[j]) ? 0.0 : Sc[i] * s.Hcc[(size_t)i * D + j] * Sc[j];
            for (int i = 0; i < D; ++i) { if (s.cconst[i]) { Sred[(size_t)i * D + i] = 1.0; rhs[i] = 0.0; } else { Sred[(size_t)i * D + i] += mu * dc[i] * dc[i]; rhs[i] = Sc[i] * s.bc[i]; } }
            std::vector<double> piv(L, 0.0), et;
            for (int l = 0; l < L; ++l) {
                if (s.lconst[l]) continue;
                const double p = Sl[l] * Sl[l] * s.hll[l] + mu * dl[l] * dl[l];
                piv[l] = p;
                const double gl = Sl[l] * s.bl[l];
                const auto& cl = lcols[l];
                const int nc = (int)cl.size();
                et.resize(nc);
                for (int a = 0; a < nc; ++a) et[a] = Sl[l] * s.E[(size_t)l * D + cl[a]] * Sc[cl[a]];
                const double ip = 1.0 / p;
                for (int a = 0; a < nc; ++a) {
                    const double ea = et[a] * ip;
                    rhs[cl[a]] -= ea * gl;
                    double* row = &Sred[(size_t)cl[a] * D];
                    for (int b = 0; b < nc; ++b) row[cl[b]] -= ea * et[b];
                }
            }
            bool chol; { ORC_TM(3); chol = cholesky_lower(D, Sred.data(), Lo.data()); }
            if (chol) {
                // forward / backward substitution
                for (int i = 0; i < D; ++i) { double v = rhs[i]; for (int k = 0; k < i; ++k) v -= Lo[(size_t)i * D + k] * xc[k]; xc[i] = v / Lo[(size_t)i * D + i]; }
                for (int i = D - 1; i >= 0; --i) { double v = xc[i]; for (int k = i + 1; k < D; ++k) v -= Lo[(size_t)k * D + i] * xc[k]; xc[i] = v / Lo[(size_t)i * D + i]; }
                bool finite = true;
                for (int i = 0; i < D; ++i) finite = finite && std::isfinite(xc[i]);
                for (int l = 0; l < L; ++l) {
                    if (s.lconst[l]) { xl[l] = 0; continue; }
                    double ev = 0; for (int col : lcols[l]) ev += Sl[l] * s.E[(size_t)l * D + col] * Sc[col] * xc[col];
                    xl[l] = (Sl[l] * s.bl[l] - ev) / piv[l];
                    finite = finite && std::isfinite(xl[l]);
                }
                if (finite) { ok = true; break; }
            }
            mu *= 10.0;
            if (!(mu < c.o.max_mu)) break;
        }
        if (!ok) return false;
        mu = std::max(c.o.min_mu, 2.0 * mu / 10.0);
        for (int i = 0; i < D; ++i) gnc[i] = -xc[i] * dc[i];
        for (int l = 0; l < L; ++l) gnl[l] = -xl[l] * dl[l];
        return true;
    }
    // traditional dogleg in dogleg space -> step in ORIGINAL coordinates
    void dogleg_step(double* stepc, double* stepl) {
        double gn2 = 0, g2 = 0, gdotgn = 0;
        for (int i = 0; i < D; ++i) { gn2 += gnc[i] * gnc[i]; g2 += gradc[i] * gradc[i]; gdotgn += gradc[i] * gnc[i]; }
        for (int l = 0; l < L; ++l) { gn2 += gnl[l] * gnl[l]; g2 += gradl[l] * gradl[l]; gdotgn += gradl[l] * gnl[l]; }
        const double gn_norm = std::sqrt(gn2), g_norm = std::sqrt(g2);
        double cg, cn;  // step = cg * gradient_ + cn * gn
        if (gn_norm <= radius) { cg = 0; cn = 1; dogleg_step_norm = gn_norm; }
        else if (g_norm * alpha >= radius) { cg = -(radius / g_norm); cn = 0; dogleg_step_norm = radius; }
        else {
            const double b_dot_a = -alpha * gdotgn;
            const double a2 = (alpha * g_norm) * (alpha * g_norm);
            const double bma2 = a2 - 2 * b_dot_a + gn2;
            const double cc = b_dot_a - a2;
            const double dd = std::sqrt(cc * cc + bma2 * (radius * radius - a2));
            const double beta = (cc <= 0) ? (dd - cc) / bma2 : (radius * radius - a2) / (dd + cc);
            cg = -alpha * (1 - beta); cn = beta; dogleg_step_norm = radius;
        }
        for (int i = 0; i < D; ++i) stepc[i] = Sc[i] * (cg * gradc[i] + cn * gnc[i]) / dc[i];
        for (int l = 0; l < L; ++l) stepl[l] = Sl[l] * (cg * gradl[l] + cn * gnl[l]) / dl[l];
    }
    void accepted(double q) { if (q < 0.25) radius *= 0.5; if (q > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm); radius = std::min(c.o.max_radius, radius); reuse = false; }
    void rejected() { radius *= 0.5; reuse = true; }
    void invalid() { mu *= 10.0; reuse = false; }
};

int solve(const vil_problem* p, vil_state* st, const vil_options* o, vil_summary* sum) {
    Ctx c(p, o);
    const int D = c.lay.D, L = p->L;
    WState x; x.from(st);
    WState cand = x;
    System sys(p->K, L);
    Dogleg dl(c);
    std::vector<double> stepc(D), stepl(L);
    auto t0 = std::chrono::steady_clock::now();
    std::memset(sum, 0, sizeof *sum);

    double cost = linearize(c, x, &sys);
    dl.init_scaling(sys);
    sum->initial_cost = cost;
    int iter = 0, nsucc = 0, invalid_run = 0;
    int term = VIL_TERM_NONE;
    if (!std::isfinite(cost)) { sum->termination = VIL_TERM_FAILURE; return VIL_ERR_NON_FINITE; }
    auto grad_max = [&]() { double m = 0; for (int i = 0; i < D; ++i) if (!sys.cconst[i]) m = std::max(m, std::fabs(sys.bc[i])); for (int l = 0; l < L; ++l) if (!sys.lconst[l]) m = std::max(m, std::fabs(sys.bl[l])); return m; };
    while (true) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (o->max_time_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= o->max_time_s) { term = VIL_TERM_MAX_TIME; break; }
        if (iter >= o->max_iterations) { term = VIL_TERM_MAX_ITERATIONS; break; }
        if (grad_max() <= o->gradient_tolerance) { term = VIL_TERM_GRADIENT_TOLERANCE; break; }
        if (dl.radius <= 1e-32) { term = VIL_TERM_FAILURE; break; }
        ++iter;
        bool valid = true;
        if (!dl.reuse) valid = dl.compute_system(sys);
        double model_change = 0;
        if (valid) {
            dl.dogleg_step(stepc.data(), stepl.data());
            // model_cost_change = -(|J d|^2 / 2 + r^T J d)
            double gd = 0;
            for (int i = 0; i < D; ++i) if (!sys.cconst[i]) gd += sys.bc[i] * stepc[i];
            for (int l = 0; l < L; ++l) if (!sys.lconst[l]) gd += sys.bl[l] * stepl[l];
            model_change = -(0.5 * dl.quad(sys, stepc.data(), stepl.data()) + gd);
            valid = model_change > 0.0;
        }
        if (!valid) {
            if (++invalid_run >= 5) { term = VIL_TERM_FAILURE; break; }
            dl.invalid();
            if (iter <= VIL_MAX_TRACE) { sum->cost_trace[iter - 1] = cost; sum->radius_trace[iter - 1] = dl.radius; }
            continue;
        }
        invalid_run = 0;
        state_plus(c, x, stepc.data(), stepl.data(), cand);
        const double cand_cost = linearize(c, cand, nullptr);
        double xnorm, stepnorm;
        free_norms(c, x, cand, &xnorm, &stepnorm);
        if (iter <= VIL_MAX_TRACE) { sum->cost_trace[iter - 1] = cost; sum->radius_trace[iter - 1] = dl.radius; }
        if (stepnorm <= o->parameter_tolerance * (xnorm + o->parameter_tolerance)) { term = VIL_TERM_PARAMETER_TOLERANCE; break; }
        if (std::fabs(cost - cand_cost) <= o->function_tolerance * cost) { term = VIL_TERM_FUNCTION_TOLERANCE; break; }
        const double rel = (cost - cand_cost) / model_change;
        if (std::isfinite(cand_cost) && rel > o->min_relative_decrease) {
            x = cand;
            cost = linearize(c, x, &sys);
            ++nsucc;
            dl.accepted(rel);
            if (iter <= VIL_MAX_TRACE) sum->cost_trace[iter - 1] = cost;
        } else {
            dl.rejected();
        }
    }
    sum->iterations = iter;
    sum->successful_steps = nsucc;
    sum->termination = term;
    sum->final_cost = cost;
    sum->t_solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    bool finite = std::isfinite(cost);
    for (double v : x.pose) finite = finite && std::isfinite(v);
    if (!finite) return VIL_ERR_NON_FINITE;
    x.to(st);
    return term == VIL_TERM_FAILURE ? VIL_ERR_NOT_POSITIVE_DEFINITE : VIL_OK;
}

// robustified cost + Schur-reduced normal equations at mu = 0, no scaling (parity surface)
int linearize_api(const vil_problem* p, const vil_state* st, const vil_options* o, double* cost, double* S, double* g) {
    Ctx c(p, o);
    const int D = c.lay.D, L = p->L;
    WState x; x.from(st);
    System sys(p->K, L);
    *cost = linearize(c, x, &sys);
    std::vector<std::vector<int>> lcols;
    landmark_cols(c, lcols);
    for (int i = 0; i < D; ++i) { g[i] = sys.cconst[i] ? 0.0 : sys.bc[i]; for (int j = 0; j < D; ++j) S[(size_t)i * D + j] = (sys.cconst[i] || sys.cconst[j]) ? 0.0 : sys.Hcc[(size_t)i * D + j]; }
    for (int l = 0; l < L; ++l) {
        if (sys.lconst[l] || sys.hll[l] == 0.0) continue;
        const double ip = 1.0 / sys.hll[l];
        for (int a : lcols[l]) {
            const double ea = sys.E[(size_t)l * D + a] * ip;
            g[a] -= ea * sys.bl[l];
            for (int b : lcols[l]) S[(size_t)a * D + b] -= ea * sys.E[(size_t)l * D + b];
        }
    }
    return VIL_OK;
}

// unreduced pieces, for tests of the device reduction (Hcc D x D, bc D, hll L, bl L, E L x D)
int linearize_full_api(const vil_problem* p, const vil_state* st, const vil_options* o, double* cost, double* Hcc, double* bc, double* hll, double* bl, double* E) {
    Ctx c(p, o);
    WState x; x.from(st);
    System sys(p->K, p->L);
    *cost = linearize(c, x, &sys);
    if (Hcc) std::copy(sys.Hcc.begin(), sys.Hcc.end(), Hcc);
    if (bc) std::copy(sys.bc.begin(), sys.bc.end(), bc);
    if (hll) std::copy(sys.hll.begin(), sys.hll.end(), hll);
    if (bl) std::copy(sys.bl.begin(), sys.bl.end(), bl);
    if (E) std::copy(sys.E.begin(), sys.E.end(), E);
    return VIL_OK;
}

double cost_api(const vil_problem* p, const vil_state* st, const vil_options* o) {
    Ctx c(p, o);
    WState x; x.from(st);
    return linearize(c, x, nullptr);
}

// estimator.cpp:960-1011  double2vector(): yaw + translation gauge fix
int gauge_fix(const double* pose0_before, vil_state* s) {
    const Mat3 R0 = quat_R(quat_from_block(pose0_before));
    const Vec3 origin_R0 = R2ypr(R0);
    const Vec3 origin_P0 = vec_from(pose0_before);
    const Mat3 R00m = quat_R(quat_from_block(s->pose));
    const Vec3 origin_R00 = R2ypr(R00m);
    const double y_diff = origin_R0.x - origin_R00.x;
    Mat3 rot_diff = ypr2R({y_diff, 0, 0});
    if (std::fabs(std::fabs(origin_R0.y) - 90) < 1.0 || std::fabs(std::fabs(origin_R00.y) - 90) < 1.0)
        rot_diff = mat_mul(R0, mat_T(R00m));
    const Vec3 p0 = vec_from(s->pose);
    for (int i = 0; i < s->K; ++i) {
        double* pp = s->pose + 7 * i;
        Mat3 R = mat_mul(rot_diff, quat_R(quat_normalized(quat_from_block(pp))));
        Vec3 P = mat_vec(rot_diff, vec_from(pp) - p0) + origin_P0;
        Quat q = quat_from_R(R);  // vector2double(): Quaterniond q{Rs[i]}
        pp[0] = P.x; pp[1] = P.y; pp[2] = P.z; pp[3] = q.x; pp[4] = q.y; pp[5] = q.z; pp[6] = q.w;
        double* sb = s->speedbias + 9 * i;
        Vec3 V = mat_vec(rot_diff, vec_from(sb));
        sb[0] = V.x; sb[1] = V.y; sb[2] = V.z;
    }
    Quat qe = quat_normalized(quat_from_block(s->ex_pose));
    qe = quat_from_R(quat_R(qe));
    s->ex_pose[3] = qe.x; s->ex_pose[4] = qe.y; s->ex_pose[5] = qe.z; s->ex_pose[6] = qe.w;
    return VIL_OK;
}

}  // namespace orc
