// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).  PARITY UNPINNED.
//
// CPU restatement of the marginalisation step of Estimator::optimization():
//   estimator.cpp:1484-1683  (which factors / which blocks are dropped, the i->i-1 address shift)
//   marginalization_factor.cpp:3-69    ResidualBlockInfo::Evaluate (+ loss corrector)
//   marginalization_factor.cpp:141-174 ThreadsConstructA  (A = sum J^T J, b = sum J^T r, NUM_THREADS=4)
//   marginalization_factor.cpp:176-316 marginalize()      (eig-based A_mm^-1, Schur, eig -> J0, r0)
//   marginalization_factor.cpp:318-338 getParameterBlocks
// The block ORDER and the eigenvector BASIS are implementation-defined in the reference
// (unordered_map over pointer values): compare J0^T J0 and J0^T r0, never J0.
#include <thread>

#include "oracle.hpp"

namespace orc {

// cyclic Jacobi eigen-decomposition of a symmetric n x n row-major matrix: A = V diag(w) V^T,
// eigenvalues ascending (like Eigen::SelfAdjointEigenSolver), V column k = eigenvector k.
// Slow (O(n^3) per sweep) but independent of the QL solver below: kept as the cross-check (tests) only.
void sym_eig_jacobi(int n, const double* Ain, double* w, double* V) {
    std::vector<double> A(Ain, Ain + (size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = (i == j);
    double fro = 0; for (size_t i = 0; i < A.size(); ++i) fro += A[i] * A[i];
    const double tol = 1e-30 * (fro > 0 ? fro : 1.0);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[(size_t)p * n + q] * A[(size_t)p * n + q];
        if (off <= tol) break;
        for (int p = 0; p < n - 1; ++p) for (int q = p + 1; q < n; ++q) {
            const double apq = A[(size_t)p * n + q];
            if (apq == 0.0) continue;
            const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
            const double tau = (aqq - app) / (2.0 * apq);
            const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
            const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = t * cs;
            for (int k = 0; k < n; ++k) {
                const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                A[(size_t)k * n + p] = cs * akp - sn * akq;
                A[(size_t)k * n + q] = sn * akp + cs * akq;
            }
            for (int k = 0; k < n; ++k) {
                const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                A[(size_t)p * n + k] = cs * apk - sn * aqk;
                A[(size_t)q * n + k] = sn * apk + cs * aqk;
            }
            for (int k = 0; k < n; ++k) {
                const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                V[(size_t)k * n + p] = cs * vkp - sn * vkq;
                V[(size_t)k * n + q] = sn * vkp + cs * vkq;
            }
        }
    }
    std::vector<int> ord(n);
    for (int i = 0; i < n; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return A[(size_t)a * n + a] < A[(size_t)b * n + b]; });
    std::vector<double> Vs((size_t)n * n);
    for (int k = 0; k < n; ++k) { w[k] = A[(size_t)ord[k] * n + ord[k]]; for (int i = 0; i < n; ++i) Vs[(size_t)i * n + k] = V[(size_t)i * n + ord[k]]; }
    std::copy(Vs.begin(), Vs.end(), V);
}

// Householder tridiagonalisation + implicit-shift QL: the same two stages Eigen::SelfAdjointEigenSolver runs
// (marginalization_factor.cpp:275,301), so the CPU timing of marginalisation is not inflated by the eigen solver.
// Symmetric n x n row-major in, A = V diag(w) V^T, eigenvalues ascending, V column k = eigenvector k.
void sym_eig(int n, const double* Ain, double* w, double* V) {
    if (n <= 0) return;
    std::vector<double> z(Ain, Ain + (size_t)n * n), e(n, 0.0);
    double* d = w;
    auto Z = [&](int i, int j) -> double& { return z[(size_t)i * n + j]; };
    // ---- stage 1: reduce to tridiagonal (d diagonal, e sub-diagonal), accumulate the orthogonal transform in z
    for (int i = n - 1; i >= 1; --i) {
        const int l = i - 1;
        double h = 0.0;
        if (l > 0) {
            double scale = 0.0;
            for (int k = 0; k <= l; ++k) scale += std::fabs(Z(i, k));
            if (scale == 0.0) e[i] = Z(i, l);
            else {
                for (int k = 0; k <= l; ++k) { Z(i, k) /= scale; h += Z(i, k) * Z(i, k); }
                double f = Z(i, l);
                double g = f >= 0.0 ? -std::sqrt(h) : std::sqrt(h);
                e[i] = scale * g; h -= f * g; Z(i, l) = f - g;
                f = 0.0;
                for (int j = 0; j <= l; ++j) {
                    Z(j, i) = Z(i, j) / h;
                    g = 0.0;
                    for (int k = 0; k <= j; ++k) g += Z(j, k) * Z(i, k);
                    for (int k = j + 1; k <= l; ++k) g += Z(k, j) * Z(i, k);
                    e[j] = g / h;
                    f += e[j] * Z(i, j);
                }
                const double hh = f / (h + h);
                for (int j = 0; j <= l; ++j) {
                    f = Z(i, j);
                    e[j] = g = e[j] - hh * f;
                    for (int k = 0; k <= j; ++k) Z(j, k) -= f * e[k] + g * Z(i, k);
                }
            }
        } else e[i] = Z(i, l);
        d[i] = h;
    }
    d[0] = 0.0; e[0] = 0.0;
    for (int i = 0; i < n; ++i) {
        const int l = i - 1;
        if (d[i] != 0.0) {
            for (int j = 0; j <= l; ++j) {
                double g = 0.0;
                for (int k = 0; k <= l; ++k) g += Z(i, k) * Z(k, j);
                for (int k = 0; k <= l; ++k) Z(k, j) -= g * Z(k, i);
            }
        }
        d[i] = Z(i, i); Z(i, i) = 1.0;
        for (int j = 0; j <= l; ++j) Z(j, i) = Z(i, j) = 0.0;
    }
    // ---- stage 2: QL with implicit shifts on the tridiagonal; rotations applied to the TRANSPOSED accumulator
    //      (row i of zt = eigenvector i: the Givens updates are contiguous)
    std::vector<double> zt((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) zt[(size_t)j * n + i] = Z(i, j);
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    for (int l = 0; l < n; ++l) {
        int iter = 0, m;
        do {
            for (m = l; m < n - 1; ++m) {
                const double dd = std::fabs(d[m]) + std::fabs(d[m + 1]);
                if (std::fabs(e[m]) <= 2.3e-16 * dd) break;
            }
            if (m != l) {
                if (++iter > 200) break;
                double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
                double r = std::hypot(g, 1.0);
                g = d[m] - d[l] + e[l] / (g + (g >= 0.0 ? std::fabs(r) : -std::fabs(r)));
                double sn = 1.0, cs = 1.0, p = 0.0;
                int i;
                for (i = m - 1; i >= l; --i) {
                    double f = sn * e[i];
                    const double b = cs * e[i];
                    e[i + 1] = r = std::hypot(f, g);
                    if (r == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
                    sn = f / r; cs = g / r;
                    g = d[i + 1] - p;
                    r = (d[i] - g) * sn + 2.0 * cs * b;
                    p = sn * r;
                    d[i + 1] = g + p;
                    g = cs * r - b;
                    double* z0 = &zt[(size_t)i * n];
                    double* z1 = z0 + n;
                    for (int k = 0; k < n; ++k) {
                        const double f1 = z1[k], f0 = z0[k];
                        z1[k] = sn * f0 + cs * f1;
                        z0[k] = cs * f0 - sn * f1;
                    }
                }
                if (r == 0.0 && i >= l) continue;
                d[l] -= p; e[l] = g; e[m] = 0.0;
            }
        } while (m != l);
    }
    std::vector<int> ord(n);
    for (int i = 0; i < n; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return d[a] < d[b]; });
    std::vector<double> ws(n);
    for (int k = 0; k < n; ++k) { ws[k] = d[ord[k]]; for (int i = 0; i < n; ++i) V[(size_t)i * n + k] = zt[(size_t)ord[k] * n + i]; }
    std::copy(ws.begin(), ws.end(), w);
}

namespace {
struct PBlock { int kind, index; };          // parameter block id; landmark: kind = 100
inline bool operator==(const PBlock& a, const PBlock& b) { return a.kind == b.kind && a.index == b.index; }
struct RBlock {                               // a ResidualBlockInfo after Evaluate()
    int nr;
    std::vector<double> r;
    std::vector<PBlock> blocks;
    std::vector<int> lsize, gsize;
    std::vector<std::vector<double>> J;       // per block, nr x gsize row-major
};
inline int gs_of(int kind) { return kind == VIL_BLK_POSE || kind == VIL_BLK_EX ? 7 : (kind == VIL_BLK_SPEEDBIAS ? 9 : 1); }
inline int ls_of(int kind) { return kind == VIL_BLK_POSE || kind == VIL_BLK_EX ? 6 : (kind == VIL_BLK_SPEEDBIAS ? 9 : 1); }
const double* block_ptr(const vil_state* s, const PBlock& b) {
    switch (b.kind) {
        case VIL_BLK_POSE: return s->pose + 7 * b.index;
        case VIL_BLK_SPEEDBIAS: return s->speedbias + 9 * b.index;
        case VIL_BLK_EX: return s->ex_pose;
        case VIL_BLK_TD: return s->td;
        default: return s->inv_depth + b.index;
    }
}
void finish(RBlock& rb, int loss, double scale) {
    std::vector<double*> Jp(rb.blocks.size());
    for (size_t b = 0; b < rb.blocks.size(); ++b) Jp[b] = rb.J[b].data();
    apply_corrector(loss, scale, rb.nr, rb.r.data(), (int)rb.blocks.size(), Jp.data(), rb.gsize.data());
}
}  // namespace

int marginalize(const vil_problem* p, const vil_state* st, const vil_options* o, const vil_marg_spec* spec, vil_prior_out* out) {
    const int K = p->K;
    std::vector<RBlock> facs;
    std::vector<PBlock> dropped;
    auto add_blocks = [](RBlock& rb, std::initializer_list<PBlock> bl) {
        for (auto& b : bl) { rb.blocks.push_back(b); int k = b.kind == 100 ? VIL_BLK_TD : b.kind; rb.gsize.push_back(gs_of(k)); rb.lsize.push_back(ls_of(k)); }
    };
    auto split = [](RBlock& rb, const double* Jall) {
        size_t off = 0;
        for (size_t b = 0; b < rb.blocks.size(); ++b) { rb.J.emplace_back(Jall + off, Jall + off + (size_t)rb.nr * rb.gsize[b]); off += (size_t)rb.nr * rb.gsize[b]; }
    };
    // ---- prior factor --------------------------------------------------------------------------
    const int drop_pose = spec->flag == VIL_MARGIN_OLD ? 0 : K - 2;
    if (p->prior.n > 0) {
        const vil_prior& pr = p->prior;
        bool has_drop_pose = false;
        for (int b = 0; b < pr.nblk; ++b) if (pr.blk_kind[b] == VIL_BLK_POSE && pr.blk_index[b] == drop_pose) has_drop_pose = true;
        if (spec->flag == VIL_MARGIN_SECOND_NEW && !has_drop_pose) { out->n = -1; return VIL_OK; }  // estimator.cpp:1620-1621: prior kept as is
        RBlock rb; rb.nr = pr.n; rb.r.resize(pr.n);
        std::vector<const double*> params(pr.nblk);
        size_t nj = 0;
        for (int b = 0; b < pr.nblk; ++b) {
            PBlock pb{pr.blk_kind[b], (pr.blk_kind[b] == VIL_BLK_POSE || pr.blk_kind[b] == VIL_BLK_SPEEDBIAS) ? pr.blk_index[b] : 0};
            add_blocks(rb, {pb});
            params[b] = block_ptr(st, pb);
            nj += (size_t)pr.n * gs_of(pr.blk_kind[b]);
        }
        std::vector<double> Jall(nj);
        prior_evaluate(pr, params.data(), rb.r.data(), Jall.data());
        split(rb, Jall.data());
        facs.push_back(std::move(rb));
    } else if (spec->flag == VIL_MARGIN_SECOND_NEW) { out->n = -1; return VIL_OK; }

    if (spec->flag == VIL_MARGIN_OLD) {
        dropped.push_back({VIL_BLK_POSE, 0});
        dropped.push_back({VIL_BLK_SPEEDBIAS, 0});
        // ICP / LPS constraint touching frame 0  estimator.cpp:1508-1533
        if (spec->icp_marg >= 0) {
            const int* id = p->icp_ids + 4 * spec->icp_marg;
            RBlock rb; rb.nr = 3; rb.r.resize(3);
            add_blocks(rb, {{VIL_BLK_POSE, 0}, {VIL_BLK_POSE, id[1]}, {VIL_BLK_POSE, id[2]}, {VIL_BLK_POSE, id[3]}});
            double J[VIL_ICP_NJ];
            icp_evaluate(p->icp_const + (size_t)spec->icp_marg * VIL_ICP_CONST, st->pose, st->pose + 7 * id[1], st->pose + 7 * id[2], st->pose + 7 * id[3], rb.r.data(), J);
            split(rb, J); finish(rb, o->rel_loss, o->rel_loss_scale);
            facs.push_back(std::move(rb));
        }
        if (spec->lps_marg >= 0) {
            const int* id = p->lps_ids + 2 * spec->lps_marg;
            RBlock rb; rb.nr = 3; rb.r.resize(3);
            add_blocks(rb, {{VIL_BLK_POSE, 0}, {VIL_BLK_POSE, id[1]}});
            double J[VIL_LPS_NJ];
            lps_evaluate(p->lps_const + (size_t)spec->lps_marg * VIL_LPS_CONST, st->pose, st->pose + 7 * id[1], rb.r.data(), J);
            split(rb, J); finish(rb, o->rel_loss, o->rel_loss_scale);
            facs.push_back(std::move(rb));
        }
        // IMU(0,1)  estimator.cpp:1535-1545
        for (int f = 0; f < p->n_imu; ++f) {
            if (!(p->imu_i[f] == 0 && p->imu_j[f] == 1)) continue;
            const double* cc = p->imu_const + (size_t)f * VIL_IMU_CONST;
            if (!(cc[16] < 10.0)) continue;
            RBlock rb; rb.nr = 15; rb.r.resize(15);
            add_blocks(rb, {{VIL_BLK_POSE, 0}, {VIL_BLK_SPEEDBIAS, 0}, {VIL_BLK_POSE, 1}, {VIL_BLK_SPEEDBIAS, 1}});
            double J[VIL_IMU_NJ];
            imu_evaluate(cc, p->G, st->pose, st->speedbias, st->pose + 7, st->speedbias + 9, rb.r.data(), J);
            split(rb, J);
            facs.push_back(std::move(rb));
        }
        // visual factors anchored in frame 0  estimator.cpp:1547-1589 (constant landmarks are dropped too, App. C #8)
        int last_l = -1;
        for (int f = 0; f < p->n_vis; ++f) {
            if (p->vis_i[f] != 0) continue;
            const int l = p->vis_l[f], j = p->vis_j[f];
            if (l != last_l) { dropped.push_back({100, l}); last_l = l; }
            RBlock rb; rb.nr = 2; rb.r.resize(2);
            if (p->use_td) add_blocks(rb, {{VIL_BLK_POSE, 0}, {VIL_BLK_POSE, j}, {VIL_BLK_EX, 0}, {100, l}, {VIL_BLK_TD, 0}});
            else add_blocks(rb, {{VIL_BLK_POSE, 0}, {VIL_BLK_POSE, j}, {VIL_BLK_EX, 0}, {100, l}});
            double J[VIL_VIS_NJ];
            visual_evaluate(p->vis_const + (size_t)f * VIL_VIS_CONST, p->sqrt_info_px, p->tr_over_row, p->use_td,
                            st->pose, st->pose + 7 * j, st->ex_pose, st->inv_depth[l], st->td[0], rb.r.data(), J);
            split(rb, J); finish(rb, o->visual_loss, o->visual_loss_scale);
            facs.push_back(std::move(rb));
        }
        // LiDAR edge / plane point factors attached to frame 0 (extended mode of this build; the reference has
        // no such factors in the window).  marginalization_factor.cpp:176-316 folds EVERY factor that touches a dropped block:
        // these touch pose 0 only, so they add to A_mm / b_m and reach the prior through the Schur complement.
        for (int f = 0; f < p->n_edge; ++f) {
            if (p->edge_pose[f] != 0) continue;
            RBlock rb; rb.nr = 3; rb.r.resize(3);
            add_blocks(rb, {{VIL_BLK_POSE, 0}});
            double J[VIL_EDGE_NJ];
            edge_evaluate(p->edge_const + (size_t)f * VIL_EDGE_CONST, p->q_lb, p->t_lb, st->pose, rb.r.data(), J);
            split(rb, J); finish(rb, o->lidar_loss, o->lidar_loss_scale);
            facs.push_back(std::move(rb));
        }
        for (int f = 0; f < p->n_plane; ++f) {
            if (p->plane_pose[f] != 0) continue;
            RBlock rb; rb.nr = 1; rb.r.resize(1);
            add_blocks(rb, {{VIL_BLK_POSE, 0}});
            double J[VIL_PLANE_NJ];
            plane_evaluate(p->plane_const + (size_t)f * VIL_PLANE_CONST, p->q_lb, p->t_lb, st->pose, rb.r.data(), J);
            split(rb, J); finish(rb, o->lidar_loss, o->lidar_loss_scale);
            facs.push_back(std::move(rb));
        }
    } else {
        dropped.push_back({VIL_BLK_POSE, drop_pose});
    }

    // ---- index blocks: dropped first (m), kept after (n), canonical order --------------------------
    std::vector<PBlock> order = dropped;
    std::vector<PBlock> kept;
    for (auto& f : facs) for (auto& b : f.blocks) {
        if (std::find(order.begin(), order.end(), b) == order.end() && std::find(kept.begin(), kept.end(), b) == kept.end()) kept.push_back(b);
    }
    std::sort(kept.begin(), kept.end(), [](const PBlock& a, const PBlock& b) { return a.kind != b.kind ? a.kind < b.kind : a.index < b.index; });
    // only blocks that actually occur are dropped
    std::vector<PBlock> dr;
    for (auto& d : dropped) { bool occ = false; for (auto& f : facs) for (auto& b : f.blocks) if (b == d) occ = true; if (occ) dr.push_back(d); }
    order = dr;
    std::vector<int> idx;
    int pos = 0;
    for (auto& b : order) { idx.push_back(pos); pos += ls_of(b.kind == 100 ? VIL_BLK_TD : b.kind); }
    const int m = pos;
    for (auto& b : kept) { order.push_back(b); idx.push_back(pos); pos += ls_of(b.kind); }
    const int n = pos - m;
    auto find_idx = [&](const PBlock& b) { for (size_t i = 0; i < order.size(); ++i) if (order[i] == b) return idx[i]; return -1; };

    // ---- A = sum J^T J, b = sum J^T r  (ThreadsConstructA, round-robin over NUM_THREADS) ------------
    const int nth = std::max(1, spec->threads);
    std::vector<std::vector<double>> At(nth, std::vector<double>((size_t)pos * pos, 0.0)), bt(nth, std::vector<double>(pos, 0.0));
    auto worker = [&](int t) {
        auto& A = At[t]; auto& bb = bt[t];
        for (size_t fi = t; fi < facs.size(); fi += nth) {
            const RBlock& f = facs[fi];
            for (size_t i = 0; i < f.blocks.size(); ++i) {
                const int ii = find_idx(f.blocks[i]), si = f.lsize[i], gi = f.gsize[i];
                for (size_t j = i; j < f.blocks.size(); ++j) {
                    const int ij = find_idx(f.blocks[j]), sj = f.lsize[j], gj = f.gsize[j];
                    for (int a = 0; a < si; ++a) for (int c2 = 0; c2 < sj; ++c2) {
                        double s = 0;
                        for (int k = 0; k < f.nr; ++k) s += f.J[i][(size_t)k * gi + a] * f.J[j][(size_t)k * gj + c2];
                        A[(size_t)(ii + a) * pos + ij + c2] += s;
                        if (i != j) A[(size_t)(ij + c2) * pos + ii + a] = A[(size_t)(ii + a) * pos + ij + c2];
                    }
                }
                for (int a = 0; a < si; ++a) { double s = 0; for (int k = 0; k < f.nr; ++k) s += f.J[i][(size_t)k * gi + a] * f.r[k]; bb[ii + a] += s; }
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nth; ++t) th.emplace_back(worker, t);
        worker(0);
        for (auto& t : th) t.join();
    }
    std::vector<double> A((size_t)pos * pos, 0.0), b(pos, 0.0);
    for (int t = nth - 1; t >= 0; --t) { for (size_t i = 0; i < A.size(); ++i) A[i] += At[t][i]; for (int i = 0; i < pos; ++i) b[i] += bt[t][i]; }

    // ---- Schur complement with eigen-based pseudo inverse (marginalization_factor.cpp:273-290) -------
    const double eps = 1e-8;
    std::vector<double> Ar((size_t)n * n), br(n);
    if (m > 0) {
        std::vector<double> Amm((size_t)m * m), w(m), V((size_t)m * m), Ainv((size_t)m * m, 0.0);
        for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Amm[(size_t)i * m + j] = 0.5 * (A[(size_t)i * pos + j] + A[(size_t)j * pos + i]);
        sym_eig(m, Amm.data(), w.data(), V.data());
        for (int k = 0; k < m; ++k) {
            if (!(w[k] > eps)) continue;
            const double iw = 1.0 / w[k];
            for (int i = 0; i < m; ++i) { const double vi = V[(size_t)i * m + k] * iw; for (int j = 0; j < m; ++j) Ainv[(size_t)i * m + j] += vi * V[(size_t)j * m + k]; }
        }
        // T = Arm * Amm_inv (n x m)
        std::vector<double> T((size_t)n * m, 0.0);
        for (int i = 0; i < n; ++i) for (int k = 0; k < m; ++k) { const double a = A[(size_t)(m + i) * pos + k]; if (a == 0.0) continue; for (int j = 0; j < m; ++j) T[(size_t)i * m + j] += a * Ainv[(size_t)k * m + j]; }
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) { double s = A[(size_t)(m + i) * pos + m + j]; for (int k = 0; k < m; ++k) s -= T[(size_t)i * m + k] * A[(size_t)k * pos + m + j]; Ar[(size_t)i * n + j] = s; }
            double s = b[m + i]; for (int k = 0; k < m; ++k) s -= T[(size_t)i * m + k] * b[k]; br[i] = s;
        }
    } else {
        for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) Ar[(size_t)i * n + j] = A[(size_t)i * pos + j]; br[i] = b[i]; }
    }
    // ---- second eigen-decomposition -> linearized_jacobians / residuals (:301-309) -------------------
    std::vector<double> w2(n), V2((size_t)n * n);
    // Eigen's SelfAdjointEigenSolver reads the lower triangle only
    std::vector<double> Asym((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) Asym[(size_t)i * n + j] = Asym[(size_t)j * n + i] = Ar[(size_t)i * n + j];
    sym_eig(n, Asym.data(), w2.data(), V2.data());
    out->n = n; out->m = m; out->nblk = (int)kept.size();
    for (int k = 0; k < n; ++k) {
        const double Sv = w2[k] > eps ? w2[k] : 0.0, Sinv = w2[k] > eps ? 1.0 / w2[k] : 0.0;
        const double ss = std::sqrt(Sv), si = std::sqrt(Sinv);
        double vb = 0;
        for (int j = 0; j < n; ++j) { out->J0[(size_t)j * n + k] = ss * V2[(size_t)j * n + k]; vb += V2[(size_t)j * n + k] * br[j]; }  // column-major: J0(k,j)
        out->r0[k] = si * vb;
    }
    if (out->A) std::copy(Ar.begin(), Ar.end(), out->A);
    if (out->b) std::copy(br.begin(), br.end(), out->b);
    // ---- getParameterBlocks with the address shift as an index remap (estimator.cpp:1599-1611,1654-1677)
    int xoff = 0;
    for (size_t i = 0; i < kept.size(); ++i) {
        const PBlock& kb = kept[i];
        out->blk_kind[i] = kb.kind;
        int ni = kb.index;
        if (kb.kind == VIL_BLK_POSE || kb.kind == VIL_BLK_SPEEDBIAS) {
            if (spec->flag == VIL_MARGIN_OLD) ni = kb.index - 1;
            else ni = (kb.index == K - 1) ? K - 2 : kb.index;
        }
        out->blk_index[i] = ni;
        out->blk_col[i] = find_idx(kb) - m;
        const double* src = block_ptr(st, kb);
        for (int q = 0; q < gs_of(kb.kind); ++q) out->x0[xoff + q] = src[q];
        xoff += gs_of(kb.kind);
    }
    return VIL_OK;
}

}  // namespace orc
