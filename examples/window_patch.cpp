// window_patch.cpp -- INTEGRATION.md section 5b, compiled: the per-image call sequence of the fully resident window (vil_win_*) from C++, on
// a sequence dumped by tests/test_example.py (the harness' synthetic replay).  Per image the dump holds what the estimator would hand over:
// the new frame (IMU samples of its interval, observations by track slot, LiDAR correspondences), the landmark list, the ICP / LPS lists and
// the state vector2double() produced; this program pushes / solves / marginalises / slides and prints what double2vector() would read back.
//   g++ -std=c++17 -Iinclude examples/window_patch.cpp mvil-fusion_amd/csrc/libvilsolve.so -o window_patch ; ./window_patch sequence.bin
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vilsolve.h"

namespace {

template <class T> bool rd(FILE* f, std::vector<T>& v) {
    long long n = 0;
    if (std::fread(&n, 8, 1, f) != 1 || n < 0) return false;
    v.resize((size_t)n);
    return n == 0 || std::fread(v.data(), sizeof(T), (size_t)n, f) == (size_t)n;
}

struct Frame {                      // vil_win_frame, owned
    std::vector<double> dt, acc, gyr, hdr /* acc0 gyr0 lin_ba lin_bg */, obs, plane, edge;
    std::vector<int32_t> obs_track;
    bool load(FILE* f) { return rd(f, dt) && rd(f, acc) && rd(f, gyr) && rd(f, hdr) && hdr.size() == 12 && rd(f, obs_track) && rd(f, obs) && rd(f, plane) && rd(f, edge); }
    vil_win_frame view() const {
        vil_win_frame fr; std::memset(&fr, 0, sizeof fr);
        fr.n_samples = (int32_t)dt.size(); fr.dt = dt.data(); fr.acc = acc.data(); fr.gyr = gyr.data();
        std::memcpy(fr.acc0, &hdr[0], 24); std::memcpy(fr.gyr0, &hdr[3], 24); std::memcpy(fr.lin_ba, &hdr[6], 24); std::memcpy(fr.lin_bg, &hdr[9], 24);
        fr.n_obs = (int32_t)obs_track.size(); fr.obs_track = obs_track.data(); fr.obs = obs.data();
        fr.n_plane = (int32_t)(plane.size() / 7); fr.plane_const = plane.data(); fr.n_edge = (int32_t)(edge.size() / 9); fr.edge_const = edge.data();
        return fr;
    }
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s sequence.bin\n", argv[0]); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::vector<int32_t> hi; std::vector<double> hd;
    if (!rd(f, hi) || !rd(f, hd) || hi.size() != 5 || hd.size() != 16) { std::fprintf(stderr, "bad header\n"); return 2; }
    const int K = hi[0], n_images = hi[4];
    vil_win_cfg cfg; std::memset(&cfg, 0, sizeof cfg);
    cfg.K = K; cfg.max_tracks = hi[1]; cfg.max_samples = hi[2]; cfg.use_td = hi[3];
    for (int k = 0; k < 4; ++k) { cfg.noise[k] = hd[k]; cfg.q_lb[k] = hd[9 + k]; }
    for (int k = 0; k < 3; ++k) { cfg.G[k] = hd[4 + k]; cfg.t_lb[k] = hd[13 + k]; }
    cfg.sqrt_info_px = hd[7]; cfg.tr_over_row = hd[8];

    vil_device_cfg dc = {0, 0, 1, 0};
    vil_ctx* vil_ = nullptr;
    int rc = vil_create(&dc, &vil_);                                         // once, in Estimator::Estimator()
    if (rc != VIL_OK) { std::fprintf(stderr, "vil_create: %s (no CPU path)\n", vil_strerror(rc)); return 3; }
    vil_set_gauge_fix(vil_, 1);                                               // double2vector()'s gauge fix runs on the device
    if ((rc = vil_win_open(vil_, &cfg)) != VIL_OK) { std::fprintf(stderr, "vil_win_open: %s\n", vil_strerror(rc)); return 4; }
    for (int k = 0; k < K; ++k) {                                             // the window that initialisation leaves: K frames
        Frame fr; if (!fr.load(f)) return 2;
        vil_win_frame v = fr.view();
        if ((rc = vil_win_push_frame(vil_, &v)) != VIL_OK) { std::fprintf(stderr, "vil_win_push_frame: %s\n", vil_strerror(rc)); return 4; }
    }
    vil_options opt; vil_default_options(&opt);
    for (int img = 0; img < n_images; ++img) {
        // ---- optimization(): the small tables of this window + the state of vector2double()
        std::vector<int32_t> meta, lm_track, lm_start, lm_nobs, icp_ids, lps_ids; std::vector<unsigned char> lm_const; std::vector<double> icp_c, lps_c, pose, sb, ex, td, lam;
        if (!(rd(f, meta) && meta.size() == 4 && rd(f, lm_track) && rd(f, lm_start) && rd(f, lm_nobs) && rd(f, lm_const) && rd(f, icp_ids) && rd(f, icp_c) && rd(f, lps_ids) && rd(f, lps_c) &&
              rd(f, pose) && rd(f, sb) && rd(f, ex) && rd(f, td) && rd(f, lam))) return 2;
        opt.max_iterations = meta[3]; opt.max_time_s = 0.0;                   // NUM_ITERATIONS; the time cap is off for reproducibility
        vil_win_problem wp; std::memset(&wp, 0, sizeof wp);
        wp.L = (int32_t)lm_track.size(); wp.lm_track = lm_track.data(); wp.lm_start = lm_start.data(); wp.lm_nobs = lm_nobs.data(); wp.lm_const = lm_const.data();
        wp.n_icp = (int32_t)(icp_ids.size() / 4); wp.icp_ids = icp_ids.data(); wp.icp_const = icp_c.data();
        wp.n_lps = (int32_t)(lps_ids.size() / 2); wp.lps_ids = lps_ids.data(); wp.lps_const = lps_c.data();
        if (lam.empty()) lam.push_back(0.0);
        vil_state st{K, wp.L, pose.data(), sb.data(), ex.data(), td.data(), lam.data()};      // aliases para_Pose / para_SpeedBias / para_Ex_Pose / para_Td / para_Feature
        vil_summary sum;
        if ((rc = vil_win_solve(vil_, &wp, &st, &opt, &sum)) != VIL_OK) { std::fprintf(stderr, "vil_win_solve: %s\n", vil_strerror(rc)); return 4; }   // state untouched on error
        vil_marg_spec ms{meta[0], meta[1], meta[2], 4};                      // marginalization_flag, the ICP / LPS constraint that touches frame 0
        vil_win_prior_info info;
        if ((rc = vil_win_marginalize(vil_, &opt, &ms, &info)) != VIL_OK) { std::fprintf(stderr, "vil_win_marginalize: %s\n", vil_strerror(rc)); return 4; }
        std::printf("IMG %d %d %d %.17g %.17g %d", img, sum.iterations, sum.termination, sum.initial_cost, sum.final_cost, info.n);
        for (int q = 0; q < 7; ++q) std::printf(" %.17g", pose[7 * (K - 1) + q]);
        std::printf("\n");
        // ---- slideWindow() + the next image's frame
        if ((rc = vil_win_drop_frame(vil_, ms.flag)) != VIL_OK) { std::fprintf(stderr, "vil_win_drop_frame: %s\n", vil_strerror(rc)); return 4; }
        Frame fr; if (!fr.load(f)) return 2;
        vil_win_frame v = fr.view();
        if ((rc = vil_win_push_frame(vil_, &v)) != VIL_OK) { std::fprintf(stderr, "vil_win_push_frame: %s\n", vil_strerror(rc)); return 4; }
    }
    std::fclose(f);
    vil_destroy(vil_);
    return 0;
}
