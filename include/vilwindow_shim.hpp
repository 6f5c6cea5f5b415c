// vilwindow_shim.hpp -- header-only C++ bookkeeping either side of the solve: the landmark indexing of
// FeatureManager (feature_manager.cpp:27-215, :286-384) and the window shift of Estimator::slideWindow()
// (estimator.cpp:1689-1814), on plain arrays, so that the tables handed to vilsolve.h / vilpreint.h come out in the
// reference's order without Eigen / ROS types.  Host logic only; nothing here is on the measured path.
#ifndef VILWINDOW_SHIM_HPP
#define VILWINDOW_SHIM_HPP

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "vilpreint.h"
#include "vilsolve_shim.hpp"

namespace vil {

// one observation of a landmark (FeaturePerFrame, feature_manager.h:18-44): the 8 numbers of the feature message
// [x y z u v vx vy depth] (estimator_node.cpp:485-503) + the td it was taken with
struct FeatureObs { double point[3], uv[2], velocity[2], cur_td, depth; };

// a landmark track (FeaturePerId, feature_manager.h:46-79)
struct FeatureTrack {
    int feature_id, start_frame;
    std::vector<FeatureObs> obs;
    double estimated_depth;
    bool lidar_depth_flag;
    int solve_flag;                                   // 0 not solved yet, 1 solved, 2 failed (negative depth)
    int end_frame() const { return start_frame + (int)obs.size() - 1; }
};

class FeatureTable {
public:
    // window_size = WINDOW_SIZE (parameters.h:12), init_depth = INIT_DEPTH, min_parallax = MIN_PARALLAX (already / FOCAL_LENGTH)
    FeatureTable(int window_size, double init_depth, double min_parallax) : W_(window_size), init_depth_(init_depth), min_parallax_(min_parallax) {}

    // addFeatureCheckParallax (feature_manager.cpp:44-106): appends the observations of image `frame_count`; returns true when
    // the second-newest frame is a keyframe (=> MARGIN_OLD), false => MARGIN_SECOND_NEW.  obs8 = n x [x y z u v vx vy depth].
    // The reference receives the image as a std::map<int, ...> and walks it in ascending feature id (feature_manager.cpp:51),
    // which fixes the order of f_manager.feature, hence feature_index / para_Feature and the factor order: the observations
    // are visited in ascending id here too, whatever order the caller's arrays are in (first entry wins for a repeated id).
    bool add_frame(int frame_count, const int* ids, const double* obs8, int n, double td) {
        last_track_num = 0;
        std::vector<int> order(n);
        for (int k = 0; k < n; ++k) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ids[a] < ids[b]; });
        for (int q = 0; q < n; ++q) {
            const int k = order[q];
            if (q > 0 && ids[k] == ids[order[q - 1]]) continue;       // std::map keeps one entry per id
            FeatureObs o;
            const double* p = obs8 + 8 * (size_t)k;
            o.point[0] = p[0]; o.point[1] = p[1]; o.point[2] = p[2]; o.uv[0] = p[3]; o.uv[1] = p[4]; o.velocity[0] = p[5]; o.velocity[1] = p[6]; o.depth = p[7]; o.cur_td = td;
            FeatureTrack* tr = find(ids[k]);
            if (!tr) {
                FeatureTrack t;
                t.feature_id = ids[k]; t.start_frame = frame_count; t.solve_flag = 0;
                t.lidar_depth_flag = o.depth > 0; t.estimated_depth = o.depth > 0 ? o.depth : -1.0;      // feature_manager.h:62-76
                t.obs.push_back(o);
                tracks_.push_back(t);
            } else {
                tr->obs.push_back(o);
                ++last_track_num;
                if (o.depth > 0 && !tr->lidar_depth_flag) { tr->estimated_depth = o.depth; tr->lidar_depth_flag = true; tr->obs[0].depth = o.depth; }   // :76-81
            }
        }
        if (frame_count < 2 || last_track_num < 20) return true;
        double sum = 0.0; int num = 0;
        for (const FeatureTrack& t : tracks_)
            if (t.start_frame <= frame_count - 2 && t.end_frame() >= frame_count - 1) { sum += parallax(t, frame_count); ++num; }
        return num == 0 ? true : sum / num >= min_parallax_;
    }
    // which tracks are landmarks of the optimisation problem (feature_manager.cpp:36, estimator.cpp:1192-1194)
    bool in_problem(const FeatureTrack& t) const { return t.obs.size() >= 2 && t.start_frame < W_ - 2; }
    int count() const { int c = 0; for (const FeatureTrack& t : tracks_) c += in_problem(t); return c; }            // getFeatureCount
    // getDepthVector (:188-206) -> para_Feature ; setDepth (:151-168) <- para_Feature ; clearDepth (:170-186)
    void depth_vector(double* inv_depth) const { int i = 0; for (const FeatureTrack& t : tracks_) if (in_problem(t)) inv_depth[i++] = 1.0 / (t.estimated_depth > 0 ? t.estimated_depth : init_depth_); }
    void set_depth(const double* inv_depth) { int i = 0; for (FeatureTrack& t : tracks_) if (in_problem(t)) { t.estimated_depth = 1.0 / inv_depth[i++]; t.solve_flag = t.estimated_depth < 0 ? 2 : 1; } }
    void clear_depth(const double* inv_depth) { int i = 0; for (FeatureTrack& t : tracks_) if (in_problem(t)) { t.estimated_depth = 1.0 / inv_depth[i++]; t.lidar_depth_flag = false; } }
    void remove_failures() { erase_if([](const FeatureTrack& t) { return t.solve_flag == 2; }); }                  // :170-179

    // triangulate (feature_manager.cpp:214-273): a landmark of the problem without a depth yet gets one from the window's poses -- the null vector of the
    // stacked projection constraints [f_x P_2 - f_z P_0 ; f_y P_2 - f_z P_1] of its observations (f = point.normalized(), P = [R^T | -R^T t] relative to
    // the anchor camera), depth = V_2 / V_3; a negative result becomes INIT_DEPTH.  pose: K x [p q(xyzw)] body poses, ex: [tic qic(xyzw)].  The reference
    // takes the last right singular vector of a JacobiSVD of the stacked rows; here: a one-sided (Hestenes) Jacobi SVD of the same 2 m x 4 matrix -- column rotations
    // until the columns are orthogonal, the column of smallest norm names the singular vector.  Working on A itself (never A^T A, whose condition number is the
    // square) keeps the null direction of a low-parallax track (sigma_min / sigma_max ~ 1e-8) resolved to the accuracy the reference has; the quotient
    // V_2 / V_3 does not see the vector's sign.
    void triangulate(const double* pose, const double* ex) {
        double Ric[9]; quat_R(ex + 3, Ric);
        for (FeatureTrack& t : tracks_) {
            if (!in_problem(t) || t.estimated_depth > 0) continue;
            std::vector<double>& A = svd_rows_; A.clear();
            double R0[9], t0[3];
            cam_pose(pose + 7 * t.start_frame, ex, Ric, R0, t0);
            for (size_t m = 0; m < t.obs.size(); ++m) {
                double R1[9], t1[3], P[12];
                cam_pose(pose + 7 * (t.start_frame + (int)m), ex, Ric, R1, t1);
                // R = R0^T R1, t = R0^T (t1 - t0); P = [R^T | -R^T t] = [R1^T R0 | -R1^T (t1 - t0)]
                const double d[3] = {t1[0] - t0[0], t1[1] - t0[1], t1[2] - t0[2]};
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) P[4 * r + c] = R1[r] * R0[c] + R1[3 + r] * R0[3 + c] + R1[6 + r] * R0[6 + c];
                    P[4 * r + 3] = -(R1[r] * d[0] + R1[3 + r] * d[1] + R1[6 + r] * d[2]);
                }
                const FeatureObs& o = t.obs[m];
                const double n = std::sqrt(o.point[0] * o.point[0] + o.point[1] * o.point[1] + o.point[2] * o.point[2]);
                const double f[3] = {o.point[0] / n, o.point[1] / n, o.point[2] / n};
                for (int row = 0; row < 2; ++row) {
                    double a[4];
                    for (int c = 0; c < 4; ++c) a[c] = f[row] * P[8 + c] - f[2] * P[4 * row + c];
                    A.insert(A.end(), a, a + 4);
                }
            }
            double V[16], nrm[4]; svd4_onesided(A.data(), (int)(A.size() / 4), V, nrm);
            int kmin = 0; for (int k = 1; k < 4; ++k) if (nrm[k] < nrm[kmin]) kmin = k;
            const double depth = V[4 * 2 + kmin] / V[4 * 3 + kmin];
            t.estimated_depth = depth < 0 ? init_depth_ : depth;
        }
    }

    // slideWindowOld with shift_depth (estimator.cpp:1798-1813, feature_manager.cpp:286-346): the oldest frame leaves; tracks
    // anchored in it move their depth into the next frame.  R / P: camera-to-world of the leaving (0) and the new first (1) frame, row-major.
    void remove_back_shift_depth(const double R0[9], const double P0[3], const double R1[9], const double P1[3]) {
        for (FeatureTrack& t : tracks_) {
            if (t.start_frame != 0) { --t.start_frame; continue; }
            const FeatureObs first = t.obs.front();
            double depth = -1.0;
            if (first.depth > 0) depth = first.depth; else if (t.estimated_depth > 0) depth = t.estimated_depth;
            t.obs.erase(t.obs.begin());
            if (t.obs.size() < 2) { t.solve_flag = -1; continue; }                  // dropped below
            const double pi[3] = {first.point[0] * depth, first.point[1] * depth, first.point[2] * depth};
            double w[3], d[3];
            for (int r = 0; r < 3; ++r) w[r] = R0[3 * r] * pi[0] + R0[3 * r + 1] * pi[1] + R0[3 * r + 2] * pi[2] + P0[r];
            for (int r = 0; r < 3; ++r) d[r] = w[r] - P1[r];
            const double dep_j = R1[2] * d[0] + R1[5] * d[1] + R1[8] * d[2];           // third row of R1^T
            if (t.obs.front().depth > 0) { t.estimated_depth = t.obs.front().depth; t.lidar_depth_flag = true; }
            else if (dep_j > 0) { t.estimated_depth = dep_j; t.lidar_depth_flag = false; }
            else { t.estimated_depth = init_depth_; t.lidar_depth_flag = false; }
        }
        erase_if([](const FeatureTrack& t) { return t.solve_flag == -1; });
    }
    void remove_back() {                                                           // feature_manager.cpp:348-363 (before initialisation)
        for (FeatureTrack& t : tracks_) { if (t.start_frame != 0) --t.start_frame; else t.obs.erase(t.obs.begin()); }
        erase_if([](const FeatureTrack& t) { return t.obs.empty(); });
    }
    void remove_front(int frame_count) {                                           // feature_manager.cpp:365-384: the second-newest frame leaves
        for (FeatureTrack& t : tracks_) {
            if (t.start_frame == frame_count) { --t.start_frame; continue; }
            if (t.end_frame() < frame_count - 1) continue;
            t.obs.erase(t.obs.begin() + (W_ - 1 - t.start_frame));
        }
        erase_if([](const FeatureTrack& t) { return t.obs.empty(); });
    }
    // the visual loop of Estimator::optimization() (estimator.cpp:1189-1242): one factor per observation after the first of
    // every landmark of the problem, landmarks numbered in table order; row_half = ROW / 2
    void pack(WindowPacker& pk, double row_half) const {
        int feature_index = -1;
        for (const FeatureTrack& t : tracks_) {
            if (!in_problem(t)) continue;
            ++feature_index;
            const FeatureObs& f0 = t.obs.front();
            for (size_t m = 1; m < t.obs.size(); ++m) {
                const FeatureObs& fj = t.obs[m];
                pk.add_visual(t.start_frame, t.start_frame + (int)m, feature_index, f0.point, fj.point, f0.velocity, fj.velocity, f0.cur_td, fj.cur_td,
                              f0.uv[1] - row_half, fj.uv[1] - row_half, t.lidar_depth_flag);
            }
        }
    }
    const std::vector<FeatureTrack>& tracks() const { return tracks_; }
    int last_track_num = 0;

    // ---- the fully resident window (vilsolve.h: vil_win_*) ------------------------------------------------------------------------
    // What vil_win_push_frame takes for window frame `frame`: every track observed in it, by track slot, as [x y z vx vy cur_td row 0]
    // (row = v - ROW / 2, projection_td_factor.cpp:12-19).  `slots`: feature id -> slot (TrackSlots below).
    template <class Slots> void win_frame_obs(int frame, double row_half, Slots& slots, std::vector<int32_t>& obs_track, std::vector<double>& obs) const {
        obs_track.clear(); obs.clear();
        for (const FeatureTrack& t : tracks_) {
            const int q = frame - t.start_frame;
            if (q < 0 || q >= (int)t.obs.size()) continue;
            const FeatureObs& o = t.obs[q];
            obs_track.push_back(slots.slot(t.feature_id));
            const double v[8] = {o.point[0], o.point[1], o.point[2], o.velocity[0], o.velocity[1], o.cur_td, o.uv[1] - row_half, 0.0};
            obs.insert(obs.end(), v, v + 8);
        }
    }
    // The landmark table of vil_win_problem: the tracks of the problem in table order (= feature_index, estimator.cpp:1192-1194)
    template <class Slots> void win_landmarks(Slots& slots, std::vector<int32_t>& lm_track, std::vector<int32_t>& lm_start, std::vector<int32_t>& lm_nobs, std::vector<uint8_t>& lm_const) const {
        lm_track.clear(); lm_start.clear(); lm_nobs.clear(); lm_const.clear();
        for (const FeatureTrack& t : tracks_) if (in_problem(t)) {
            lm_track.push_back(slots.slot(t.feature_id)); lm_start.push_back(t.start_frame); lm_nobs.push_back((int32_t)t.obs.size()); lm_const.push_back(t.lidar_depth_flag ? 1 : 0);
        }
    }

private:
    static void quat_R(const double* q, double* R) {            // [x y z w] -> row-major rotation
        const double x = q[0], y = q[1], z = q[2], w = q[3];
        R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
        R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
        R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
    }
    static void cam_pose(const double* pose7, const double* ex, const double* Ric, double* Rc, double* tc) {      // R_wc = R_wb R_ic, t_wc = p + R_wb t_ic
        double Rb[9]; quat_R(pose7 + 3, Rb);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) Rc[3 * r + c] = Rb[3 * r] * Ric[c] + Rb[3 * r + 1] * Ric[3 + c] + Rb[3 * r + 2] * Ric[6 + c];
            tc[r] = pose7[r] + Rb[3 * r] * ex[0] + Rb[3 * r + 1] * ex[1] + Rb[3 * r + 2] * ex[2];
        }
    }
    // one-sided Jacobi SVD of A (rows x 4, row-major, overwritten by U Sigma): V = right singular vectors in columns, nrm = singular values (unsorted)
    static void svd4_onesided(double* A, int rows, double* V, double* nrm) {
        for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
        for (int sweep = 0; sweep < 60; ++sweep) {
            bool rotated = false;
            for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) {
                double app = 0.0, aqq = 0.0, apq = 0.0;
                for (int r = 0; r < rows; ++r) { const double x = A[4 * r + p], y = A[4 * r + q]; app += x * x; aqq += y * y; apq += x * y; }
                if (apq == 0.0 || std::fabs(apq) <= 1e-16 * std::sqrt(app * aqq)) continue;      // the pair is orthogonal to working precision
                rotated = true;
                const double th = (aqq - app) / (2.0 * apq);
                const double tt = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0)), cs = 1.0 / std::sqrt(tt * tt + 1.0), sn = tt * cs;
                for (int r = 0; r < rows; ++r) { const double x = A[4 * r + p], y = A[4 * r + q]; A[4 * r + p] = cs * x - sn * y; A[4 * r + q] = sn * x + cs * y; }
                for (int k = 0; k < 4; ++k) { const double x = V[4 * k + p], y = V[4 * k + q]; V[4 * k + p] = cs * x - sn * y; V[4 * k + q] = sn * x + cs * y; }
            }
            if (!rotated) break;
        }
        for (int k = 0; k < 4; ++k) { double s2 = 0.0; for (int r = 0; r < rows; ++r) s2 += A[4 * r + k] * A[4 * r + k]; nrm[k] = std::sqrt(s2); }
    }
    std::vector<double> svd_rows_;
    FeatureTrack* find(int id) { for (FeatureTrack& t : tracks_) if (t.feature_id == id) return &t; return nullptr; }
    template <class Pred> void erase_if(Pred p) { tracks_.erase(std::remove_if(tracks_.begin(), tracks_.end(), p), tracks_.end()); }
    // compensatedParallax2 (:386-417): image-plane distance between the second- and third-newest frames
    static double parallax(const FeatureTrack& t, int frame_count) {
        const FeatureObs& fi = t.obs[frame_count - 2 - t.start_frame]; const FeatureObs& fj = t.obs[frame_count - 1 - t.start_frame];
        const double du = fi.point[0] / fi.point[2] - fj.point[0], dv = fi.point[1] / fi.point[2] - fj.point[1];
        return std::sqrt(du * du + dv * dv);
    }
    int W_; double init_depth_, min_parallax_;
    std::vector<FeatureTrack> tracks_;
};

// Track slots of the device-resident observation store: one per live feature track, handed out on first use, reusable once the track
// has left the table (call retain() after every slide).
class TrackSlots {
public:
    explicit TrackSlots(int max_tracks) : id_of_(max_tracks, -1) { for (int q = max_tracks - 1; q >= 0; --q) free_.push_back(q); }
    int slot(int feature_id) {                        // -1: no slot left (more live tracks than vil_win_cfg.max_tracks)
        for (size_t q = 0; q < id_of_.size(); ++q) if (id_of_[q] == feature_id) return (int)q;
        if (free_.empty()) return -1;
        const int q = free_.back(); free_.pop_back(); id_of_[q] = feature_id;
        return q;
    }
    void retain(const std::vector<FeatureTrack>& live) {
        for (size_t q = 0; q < id_of_.size(); ++q) {
            if (id_of_[q] < 0) continue;
            bool found = false;
            for (const FeatureTrack& t : live) if (t.feature_id == id_of_[q]) { found = true; break; }
            if (!found) { id_of_[q] = -1; free_.push_back((int)q); }
        }
    }
private:
    std::vector<int> id_of_, free_;
};

// ---- the window's per-frame state and IMU sample buffers (estimator.h:96-130), and its shift -------------------------------
// Frame k owns: stamp, pose [p q(xyzw)], speed-bias [v ba bg], and the samples integrated between frame k-1 and k with the
// measurement (acc0, gyr0) and biases (lin_ba, lin_bg) its IntegrationBase was created with; record[k] = their 287-number
// pre-integration (vilsolve.h), produced by vpre_integrate for the frames marked dirty.
class WindowFrames {
public:
    explicit WindowFrames(int K) : stamp(K, 0.0), pose(7 * (size_t)K, 0.0), speedbias(9 * (size_t)K, 0.0), dt(K), acc(K), gyr(K),
                                   acc0(3 * (size_t)K, 0.0), gyr0(3 * (size_t)K, 0.0), lin_ba(3 * (size_t)K, 0.0), lin_bg(3 * (size_t)K, 0.0),
                                   record((size_t)VIL_IMU_CONST * K, 0.0), dirty(K, 1), K_(K) { for (int k = 0; k < K; ++k) pose[7 * k + 6] = 1.0; }
    int K() const { return K_; }
    // processIMU (estimator.cpp:121-146): a sample arrives for the newest frame `k`
    void push_sample(int k, double dt_, const double a[3], const double g[3]) { dt[k].push_back(dt_); acc[k].insert(acc[k].end(), a, a + 3); gyr[k].insert(gyr[k].end(), g, g + 3); dirty[k] = 1; }
    // new IntegrationBase{acc_0, gyr_0, Bas[k], Bgs[k]} for frame k (estimator.cpp:125-128, :1725, :1777)
    void reset_interval(int k, const double a0[3], const double g0[3]) {
        dt[k].clear(); acc[k].clear(); gyr[k].clear(); dirty[k] = 1;
        std::memcpy(&acc0[3 * k], a0, 24); std::memcpy(&gyr0[3 * k], g0, 24);
        std::memcpy(&lin_ba[3 * k], &speedbias[9 * k + 3], 24); std::memcpy(&lin_bg[3 * k], &speedbias[9 * k + 6], 24);
    }
    // MARGIN_OLD (estimator.cpp:1693-1751): every frame moves one slot down, the newest is duplicated and gets a fresh interval
    void slide_old(const double a0[3], const double g0[3]) {
        for (int k = 0; k + 1 < K_; ++k) move_frame(k + 1, k);
        copy_state(K_ - 2, K_ - 1);
        reset_interval(K_ - 1, a0, g0);
    }
    // MARGIN_SECOND_NEW (:1754-1786): the newest frame replaces the second-newest, whose interval absorbs the newest one's samples
    void slide_new(const double a0[3], const double g0[3]) {
        const int n = K_ - 1, m = K_ - 2;
        dt[m].insert(dt[m].end(), dt[n].begin(), dt[n].end()); acc[m].insert(acc[m].end(), acc[n].begin(), acc[n].end()); gyr[m].insert(gyr[m].end(), gyr[n].begin(), gyr[n].end());
        dirty[m] = 1;                                      // push_back continues the integration with the interval's own linearisation point
        copy_state(n, m);
        reset_interval(n, a0, g0);
    }
    // (re-)integrates the dirty intervals in one batch on the device; returns the vpre_integrate status
    int integrate(vpre_ctx* ctx, const double noise4[4]) {
        std::vector<int32_t> idx, start(1, 0);
        std::vector<double> d, a, g, a0, g0, ba, bg;
        for (int k = 0; k < K_; ++k) if (dirty[k]) {
            idx.push_back(k);
            d.insert(d.end(), dt[k].begin(), dt[k].end()); a.insert(a.end(), acc[k].begin(), acc[k].end()); g.insert(g.end(), gyr[k].begin(), gyr[k].end());
            start.push_back((int32_t)d.size());
            a0.insert(a0.end(), &acc0[3 * k], &acc0[3 * k] + 3); g0.insert(g0.end(), &gyr0[3 * k], &gyr0[3 * k] + 3);
            ba.insert(ba.end(), &lin_ba[3 * k], &lin_ba[3 * k] + 3); bg.insert(bg.end(), &lin_bg[3 * k], &lin_bg[3 * k] + 3);
        }
        if (idx.empty()) return 0;
        std::vector<double> out((size_t)VIL_IMU_CONST * idx.size());
        const int st = vpre_integrate(ctx, (int32_t)idx.size(), start.data(), d.data(), a.data(), g.data(), a0.data(), g0.data(), ba.data(), bg.data(), noise4, out.data(), nullptr);
        if (st != 0) return st;
        for (size_t q = 0; q < idx.size(); ++q) { std::memcpy(&record[(size_t)VIL_IMU_CONST * idx[q]], &out[(size_t)VIL_IMU_CONST * q], sizeof(double) * VIL_IMU_CONST); dirty[idx[q]] = 0; }
        return 0;
    }
    // the IMU loop of Estimator::optimization() (estimator.cpp:1179-1186)
    void pack(WindowPacker& pk) const { for (int k = 1; k < K_; ++k) pk.add_imu(k - 1, k, &record[(size_t)VIL_IMU_CONST * k]); }

    std::vector<double> stamp, pose, speedbias;
    std::vector<std::vector<double>> dt, acc, gyr;
    std::vector<double> acc0, gyr0, lin_ba, lin_bg, record;
    std::vector<uint8_t> dirty;

private:
    void copy_state(int from, int to) { stamp[to] = stamp[from]; std::memcpy(&pose[7 * to], &pose[7 * from], 56); std::memcpy(&speedbias[9 * to], &speedbias[9 * from], 72); }
    void move_frame(int from, int to) {
        copy_state(from, to);
        dt[to].swap(dt[from]); acc[to].swap(acc[from]); gyr[to].swap(gyr[from]);
        std::memcpy(&acc0[3 * to], &acc0[3 * from], 24); std::memcpy(&gyr0[3 * to], &gyr0[3 * from], 24); std::memcpy(&lin_ba[3 * to], &lin_ba[3 * from], 24); std::memcpy(&lin_bg[3 * to], &lin_bg[3 * from], 24);
        std::memcpy(&record[(size_t)VIL_IMU_CONST * to], &record[(size_t)VIL_IMU_CONST * from], sizeof(double) * VIL_IMU_CONST);
        dirty[to] = dirty[from];
    }
    int K_;
};

}  // namespace vil
#endif
