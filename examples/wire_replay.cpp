// wire_replay.cpp -- SURVEY 8(f) row 4, end to end: the bag-replay harness minus the bag.  A recorded sequence of the messages either side of the
// estimator node is driven through the fully resident window on the GPU and the reference's trajectory log comes out:
//
//   /feature_tracker_/feature bytes (points + 6 float32 channels)   estimator_node.cpp:485-503      -> vil::decode_feature_cloud   (vilformat.hpp)
//   IMU samples of the interval                                     estimator.cpp:86-120 processIMU -> state propagation + vil_win_frame samples
//   addFeatureCheckParallax / triangulate / setDepth / slide        feature_manager.cpp:45-384      -> vil::FeatureTable, vil::TrackSlots, vil::WindowFrames
//   optimization() + slideWindow()                                  estimator.cpp:1124-1814         -> vil_win_push_frame / solve / marginalize / drop_frame
//   "Frontend.txt"                                                  visualization.cpp:199-212       -> vil::format_trajectory_line
//
// The window's first K frames stand in for the initial alignment (initialStructure, estimator.cpp:640-860: SfM + visual-inertial alignment, not part of
// this path): their poses / velocities / biases come with the sequence.  The sequence file is written by tests/wire_chain.py (length-prefixed arrays).
//   g++ -std=c++17 -Iinclude examples/wire_replay.cpp mvil-fusion_amd/csrc/libvilsolve.so -o wire_replay ; ./wire_replay sequence.bin Frontend.txt
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vilformat.hpp"
#include "vilsolve.h"
#include "vilwindow_shim.hpp"

namespace {

template <class T> bool rd(FILE* f, std::vector<T>& v) {
    long long n = 0;
    if (std::fread(&n, 8, 1, f) != 1 || n < 0) return false;
    v.resize((size_t)n);
    return n == 0 || std::fread(v.data(), sizeof(T), (size_t)n, f) == (size_t)n;
}

struct Record {                     // everything that arrives for one image
    std::vector<double> stamp, dt, acc, gyr, first /* acc0 gyr0 */, plane, edge, init /* pose 7 | speed-bias 9: bootstrap frames only */;
    std::vector<float> points, ch[6];
    bool load(FILE* f) {
        if (!(rd(f, stamp) && stamp.size() == 1 && rd(f, dt) && rd(f, acc) && rd(f, gyr) && rd(f, first) && first.size() == 6 && rd(f, points))) return false;
        for (int c = 0; c < 6; ++c) if (!rd(f, ch[c]) || ch[c].size() * 3 != points.size()) return false;
        return rd(f, plane) && rd(f, edge) && rd(f, init);
    }
};

void quat_R(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
// q <- q (x) deltaQ(theta) = q (x) [1, theta / 2], normalised (utility.h:12-24; Rs[j] *= deltaQ(un_gyr * dt).toRotationMatrix(), estimator.cpp:112)
void quat_mul_delta(double* q, const double th[3]) {
    const double dx = 0.5 * th[0], dy = 0.5 * th[1], dz = 0.5 * th[2], dn = std::sqrt(1.0 + dx * dx + dy * dy + dz * dz);
    const double bw = 1.0 / dn, bx = dx / dn, by = dy / dn, bz = dz / dn;      // (toRotationMatrix of the un-normalised deltaQ = rotation of its normalisation)
    const double ax = q[0], ay = q[1], az = q[2], aw = q[3];
    double r[4] = {aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz};
    const double n = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    for (int k = 0; k < 4; ++k) q[k] = r[k] / n;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s sequence.bin Frontend.txt\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::vector<int32_t> hi; std::vector<double> hd;
    if (!rd(f, hi) || !rd(f, hd) || hi.size() != 7 || hd.size() != 27) { std::fprintf(stderr, "bad header\n"); return 2; }
    const int K = hi[0], n_images = hi[4], num_of_cam = hi[6];
    const double row_half = hi[5];
    vil_win_cfg cfg; std::memset(&cfg, 0, sizeof cfg);
    cfg.K = K; cfg.max_tracks = hi[1]; cfg.max_samples = hi[2]; cfg.use_td = hi[3];
    for (int k = 0; k < 4; ++k) { cfg.noise[k] = hd[k]; cfg.q_lb[k] = hd[9 + k]; }
    for (int k = 0; k < 3; ++k) { cfg.G[k] = hd[4 + k]; cfg.t_lb[k] = hd[13 + k]; }
    cfg.sqrt_info_px = hd[7]; cfg.tr_over_row = hd[8];
    const double init_depth = hd[16], min_parallax = hd[17];
    double td = hd[18], ex[7]; for (int k = 0; k < 7; ++k) ex[k] = hd[19 + k];
    const int max_iterations = (int)hd[26];

    vil_device_cfg dc = {0, 0, 1, 0};
    vil_ctx* vil_ = nullptr;
    int rc = vil_create(&dc, &vil_);
    if (rc != VIL_OK) { std::fprintf(stderr, "vil_create: %s (no CPU path)\n", vil_strerror(rc)); return 3; }
    if ((rc = vil_win_open(vil_, &cfg)) != VIL_OK) { std::fprintf(stderr, "vil_win_open: %s\n", vil_strerror(rc)); return 4; }

    vil::FeatureTable ft(K - 1 /* WINDOW_SIZE */, init_depth, min_parallax);
    vil::TrackSlots slots(cfg.max_tracks);
    vil::WindowFrames wf(K);
    std::vector<int32_t> obs_track; std::vector<double> obs;
    bool keyframe = true;

    // one image arrives for window frame k: its IMU interval, its feature message, its LiDAR correspondences.  bootstrap: the state comes with the record
    // (initial alignment); afterwards it is propagated through the samples like processIMU does.
    auto take_frame = [&](int k, const Record& r, bool bootstrap) -> int {
        const int ns = (int)r.dt.size();
        if (bootstrap) {
            std::memcpy(&wf.pose[7 * k], &r.init[0], 56); std::memcpy(&wf.speedbias[9 * k], &r.init[7], 72);
            wf.reset_interval(k, &r.first[0], &r.first[3]);                   // new IntegrationBase{acc_0, gyr_0, Bas[k], Bgs[k]} (estimator.cpp:95-98)
        }
        wf.stamp[k] = r.stamp[0];
        double a0[3], g0[3]; std::memcpy(a0, &wf.acc0[3 * k], 24); std::memcpy(g0, &wf.gyr0[3 * k], 24);
        for (int s = 0; s < ns; ++s) {
            const double dt = r.dt[s]; const double* a1 = &r.acc[3 * s]; const double* g1 = &r.gyr[3 * s];
            wf.push_sample(k, dt, a1, g1);
            if (!bootstrap) {                                                 // estimator.cpp:109-116
                double* P = &wf.pose[7 * k]; double* V = &wf.speedbias[9 * k]; const double* ba = V + 3; const double* bg = V + 6;
                double R[9]; quat_R(P + 3, R);
                double ua0[3], ua1[3], th[3];
                for (int i = 0; i < 3; ++i) ua0[i] = R[3 * i] * (a0[0] - ba[0]) + R[3 * i + 1] * (a0[1] - ba[1]) + R[3 * i + 2] * (a0[2] - ba[2]) - cfg.G[i];
                for (int i = 0; i < 3; ++i) th[i] = (0.5 * (g0[i] + g1[i]) - bg[i]) * dt;
                quat_mul_delta(P + 3, th);
                quat_R(P + 3, R);
                for (int i = 0; i < 3; ++i) ua1[i] = R[3 * i] * (a1[0] - ba[0]) + R[3 * i + 1] * (a1[1] - ba[1]) + R[3 * i + 2] * (a1[2] - ba[2]) - cfg.G[i];
                for (int i = 0; i < 3; ++i) { const double ua = 0.5 * (ua0[i] + ua1[i]); P[i] += dt * V[i] + 0.5 * dt * dt * ua; V[i] += dt * ua; }
            }
            std::memcpy(a0, a1, 24); std::memcpy(g0, g1, 24);
        }
        // the feature message, as estimator_node.cpp:485-503 decodes it
        const float* chp[6]; for (int c = 0; c < 6; ++c) chp[c] = r.ch[c].data();
        vil::FeatureFrame ff;
        if (!vil::decode_feature_cloud((int)r.ch[0].size(), r.points.data(), chp, num_of_cam, true, ff)) { std::fprintf(stderr, "feature point with z != 1\n"); return -1; }
        keyframe = ft.add_frame(k, ff.ids.data(), ff.obs8.data(), (int)ff.ids.size(), td);      // addFeatureCheckParallax -> marginalization_flag
        ft.win_frame_obs(k, row_half, slots, obs_track, obs);
        for (int q : obs_track) if (q < 0) { std::fprintf(stderr, "more live tracks than max_tracks\n"); return -1; }
        vil_win_frame fr; std::memset(&fr, 0, sizeof fr);
        fr.n_samples = ns; fr.dt = wf.dt[k].data(); fr.acc = wf.acc[k].data(); fr.gyr = wf.gyr[k].data();
        std::memcpy(fr.acc0, &wf.acc0[3 * k], 24); std::memcpy(fr.gyr0, &wf.gyr0[3 * k], 24); std::memcpy(fr.lin_ba, &wf.lin_ba[3 * k], 24); std::memcpy(fr.lin_bg, &wf.lin_bg[3 * k], 24);
        fr.n_obs = (int32_t)obs_track.size(); fr.obs_track = obs_track.data(); fr.obs = obs.data();
        fr.n_plane = (int32_t)(r.plane.size() / 7); fr.plane_const = r.plane.data(); fr.n_edge = (int32_t)(r.edge.size() / 9); fr.edge_const = r.edge.data();
        return vil_win_push_frame(vil_, &fr);
    };

    Record rec;
    for (int k = 0; k < K; ++k) {
        if (!rec.load(f)) { std::fprintf(stderr, "short sequence\n"); return 2; }
        if ((rc = take_frame(k, rec, true)) != VIL_OK) { std::fprintf(stderr, "frame %d: %s\n", k, rc < 0 ? vil_strerror(rc) : "bad input"); return 4; }
    }
    FILE* log = std::fopen(argv[2], "w");
    if (!log) { std::fprintf(stderr, "cannot write %s\n", argv[2]); return 2; }
    vil_options opt; vil_default_options(&opt);
    opt.max_iterations = max_iterations; opt.max_time_s = 0.0;               // NUM_ITERATIONS; the time cap is off for reproducibility
    std::vector<int32_t> lm_track, lm_start, lm_nobs; std::vector<uint8_t> lm_const; std::vector<double> lam;
    for (int img = 0; img < n_images; ++img) {
        const int flag = keyframe ? VIL_MARGIN_OLD : VIL_MARGIN_SECOND_NEW;   // estimator.cpp:512-515
        // ---- solveOdometry(): triangulate, optimization()
        ft.triangulate(wf.pose.data(), ex);
        ft.win_landmarks(slots, lm_track, lm_start, lm_nobs, lm_const);
        lam.assign(lm_track.size() + 1, 0.0); ft.depth_vector(lam.data());
        vil_win_problem wp; std::memset(&wp, 0, sizeof wp);
        wp.L = (int32_t)lm_track.size(); wp.lm_track = lm_track.data(); wp.lm_start = lm_start.data(); wp.lm_nobs = lm_nobs.data(); wp.lm_const = lm_const.data();
        vil_state st{K, wp.L, wf.pose.data(), wf.speedbias.data(), ex, &td, lam.data()};
        vil_summary sum;
        if ((rc = vil_win_solve(vil_, &wp, &st, &opt, &sum)) != VIL_OK) { std::fprintf(stderr, "vil_win_solve: %s\n", vil_strerror(rc)); return 4; }      // comes back gauge-fixed (double2vector)
        ft.set_depth(lam.data());
        vil_marg_spec ms{flag, -1, -1, 4};
        vil_win_prior_info info;
        if ((rc = vil_win_marginalize(vil_, &opt, &ms, &info)) != VIL_OK) { std::fprintf(stderr, "vil_win_marginalize: %s\n", vil_strerror(rc)); return 4; }
        char line[256];
        vil::format_trajectory_line(wf.stamp[K - 1], &wf.pose[7 * (K - 1)], &wf.pose[7 * (K - 1) + 3], line, sizeof line);      // pubOdometry -> VINS_RESULT_PATH
        std::fputs(line, log);
        std::printf("IMG %d flag %d L %d iterations %d cost %.9g -> %.9g prior n %d\n", img, flag, wp.L, sum.iterations, sum.initial_cost, sum.final_cost, info.n);
        if (img + 1 == n_images) break;
        if (!rec.load(f)) { std::fprintf(stderr, "short sequence\n"); return 2; }
        // ---- slideWindow() + removeFailures()
        if (flag == VIL_MARGIN_OLD) {
            double Ric[9], Rb[9], R0[9], R1[9], P0[3], P1[3];
            quat_R(ex + 3, Ric);
            for (int w = 0; w < 2; ++w) {                                        // camera-to-world of the leaving frame and the new first one (estimator.cpp:1798-1811)
                double* Rc = w ? R1 : R0; double* Pc = w ? P1 : P0; const double* p = &wf.pose[7 * w];
                quat_R(p + 3, Rb);
                for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Rc[3 * r + c] = Rb[3 * r] * Ric[c] + Rb[3 * r + 1] * Ric[3 + c] + Rb[3 * r + 2] * Ric[6 + c]; Pc[r] = p[r] + Rb[3 * r] * ex[0] + Rb[3 * r + 1] * ex[1] + Rb[3 * r + 2] * ex[2]; }
            }
            ft.remove_back_shift_depth(R0, P0, R1, P1);
            wf.slide_old(&rec.first[0], &rec.first[3]);
        } else {
            ft.remove_front(K - 1);
            wf.slide_new(&rec.first[0], &rec.first[3]);
        }
        if ((rc = vil_win_drop_frame(vil_, flag)) != VIL_OK) { std::fprintf(stderr, "vil_win_drop_frame: %s\n", vil_strerror(rc)); return 4; }
        ft.remove_failures();
        slots.retain(ft.tracks());
        // ---- the next image
        if ((rc = take_frame(K - 1, rec, false)) != VIL_OK) { std::fprintf(stderr, "image %d: %s\n", img + 1, rc < 0 ? vil_strerror(rc) : "bad input"); return 4; }
    }
    std::fclose(log); std::fclose(f);
    vil_destroy(vil_);
    return 0;
}
