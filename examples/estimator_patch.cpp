// estimator_patch.cpp -- the patch of INTEGRATION.md, compiled: a stand-in `Estimator` that carries the reference's member
// names (estimator.h:96-147) and runs the replacement body of Estimator::optimization() (estimator.cpp:1124-1687) through
// include/vilsolve.h + include/vilsolve_shim.hpp.  The window comes from a binary dump written by tests/test_example.py
// (the same tables the Python harness passes), so the output can be compared with the harness' own solve of that window.
//   g++ -std=c++17 -Iinclude examples/estimator_patch.cpp mvil-fusion_amd/csrc/libvilsolve.so -o estimator_patch
//   ./estimator_patch window.bin
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vilsolve_shim.hpp"

namespace {

const int WINDOW_SIZE_MAX = 20;
enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };

struct Dump {                       // what the estimator holds when optimization() is entered
    int K = 0, L = 0, use_td = 1, ex_const = 0, td_const = 0;
    double sqrt_info_px = 230.0, tr_over_row = 0.0, G[3] = {0, 0, 9.8}, q_lb[4] = {0, 0, 0, 1}, t_lb[3] = {0, 0, 0};
    std::vector<double> pose, speedbias, ex_pose, td, inv_depth;
    std::vector<unsigned char> pose_const, sb_const, lm_const;
    std::vector<int> imu_i, imu_j, vis_i, vis_j, vis_l, icp_ids, lps_ids, edge_pose, plane_pose, prior_kind, prior_index, prior_col;
    std::vector<double> imu_const, vis_const, icp_const, lps_const, edge_const, plane_const, prior_x0, prior_J0, prior_r0;
    int prior_n = 0;
};

template <class T> bool rd(FILE* f, std::vector<T>& v) {
    long long n = 0;
    if (std::fread(&n, 8, 1, f) != 1 || n < 0) return false;
    v.resize((size_t)n);
    return n == 0 || std::fread(v.data(), sizeof(T), (size_t)n, f) == (size_t)n;
}
bool load(const char* path, Dump& d) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::vector<int> hi; std::vector<double> hd;
    bool ok = rd(f, hi) && rd(f, hd) && hi.size() == 6 && hd.size() == 12;
    if (ok) {
        d.K = hi[0]; d.L = hi[1]; d.use_td = hi[2]; d.ex_const = hi[3]; d.td_const = hi[4]; d.prior_n = hi[5];
        d.sqrt_info_px = hd[0]; d.tr_over_row = hd[1];
        for (int k = 0; k < 3; ++k) { d.G[k] = hd[2 + k]; d.t_lb[k] = hd[9 + k]; }
        for (int k = 0; k < 4; ++k) d.q_lb[k] = hd[5 + k];
    }
    ok = ok && rd(f, d.pose) && rd(f, d.speedbias) && rd(f, d.ex_pose) && rd(f, d.td) && rd(f, d.inv_depth) && rd(f, d.pose_const) && rd(f, d.sb_const) && rd(f, d.lm_const) &&
         rd(f, d.imu_i) && rd(f, d.imu_j) && rd(f, d.imu_const) && rd(f, d.vis_i) && rd(f, d.vis_j) && rd(f, d.vis_l) && rd(f, d.vis_const) && rd(f, d.icp_ids) && rd(f, d.icp_const) &&
         rd(f, d.lps_ids) && rd(f, d.lps_const) && rd(f, d.edge_pose) && rd(f, d.edge_const) && rd(f, d.plane_pose) && rd(f, d.plane_const) &&
         rd(f, d.prior_kind) && rd(f, d.prior_index) && rd(f, d.prior_col) && rd(f, d.prior_x0) && rd(f, d.prior_J0) && rd(f, d.prior_r0);
    std::fclose(f);
    return ok;
}

class Estimator {
public:
    explicit Estimator(const Dump& d) : dump_(d), WINDOW_SIZE(d.K - 1), vil_prior_(d.K) {
        vil_device_cfg cfg = {0, 0, 1, 0};
        status = vil_create(&cfg, &vil_);                                   // once, in Estimator::Estimator()
        for (int i = 0; i < d.K; ++i) { for (int q = 0; q < 7; ++q) para_Pose[i][q] = d.pose[7 * i + q]; for (int q = 0; q < 9; ++q) para_SpeedBias[i][q] = d.speedbias[9 * i + q]; }
        for (int q = 0; q < 7; ++q) para_Ex_Pose[0][q] = d.ex_pose[q];
        para_Td[0][0] = d.td[0];
        para_Feature.assign(d.inv_depth.begin(), d.inv_depth.end()); if (para_Feature.empty()) para_Feature.push_back(0.0);
    }
    ~Estimator() { vil_destroy(vil_); }

    // ---- the replacement body; names as in estimator.cpp --------------------------------------------
    int optimization(vil_summary& sum) {
        const Dump& d = dump_;
        const int K = WINDOW_SIZE + 1;
        vil::WindowPacker pk(K, d.L);                                        // f_manager.getFeatureCount()
        pk.set_constants(d.G, 2.0 * d.sqrt_info_px, d.tr_over_row, !d.ex_const, d.use_td != 0);   // G, FOCAL_LENGTH, TR / ROW, ESTIMATE_EXTRINSIC, ESTIMATE_TD
        pk.set_lidar_extrinsic(d.q_lb, d.t_lb);
        vil_prior last;                                                      // last_marginalization_info of the previous image
        if (d.prior_n > 0) {
            last.n = d.prior_n; last.nblk = (int)d.prior_kind.size();
            last.blk_kind = d.prior_kind.data(); last.blk_index = d.prior_index.data(); last.blk_col = d.prior_col.data();
            last.x0 = d.prior_x0.data(); last.J0 = d.prior_J0.data(); last.r0 = d.prior_r0.data();
            pk.set_prior(last);                                              // MarginalizationFactor (:1171-1177)
        }
        for (size_t f = 0; f < d.imu_i.size(); ++f) pk.add_imu(d.imu_i[f], d.imu_j[f], &d.imu_const[287 * f]);      // IMUFactor (:1179-1186)
        for (size_t f = 0; f < d.vis_i.size(); ++f) {                        // ProjectionTdFactor (:1189-1242), one call per observation after the first
            const double* c = &d.vis_const[14 * f];
            pk.add_visual(d.vis_i[f], d.vis_j[f], d.vis_l[f], c, c + 3, c + 6, c + 8, c[10], c[11], c[12], c[13], d.lm_const[d.vis_l[f]] != 0);
        }
        for (size_t f = 0; f < d.lps_ids.size() / 2; ++f) { const double* c = &d.lps_const[7 * f]; pk.add_lps(d.lps_ids[2 * f], d.lps_ids[2 * f + 1], c[0], c[1], c[2], c + 3); }
        for (size_t f = 0; f < d.icp_ids.size() / 4; ++f) { const double* c = &d.icp_const[10 * f]; pk.add_icp(d.icp_ids[4 * f], d.icp_ids[4 * f + 1], d.icp_ids[4 * f + 2], d.icp_ids[4 * f + 3], c[0], c[1], c[2], c[3], c[4], c[5], c + 6, c[9]); }
        for (size_t f = 0; f < d.edge_pose.size(); ++f) { const double* c = &d.edge_const[9 * f]; pk.add_edge(d.edge_pose[f], c, c + 3, c + 6); }
        for (size_t f = 0; f < d.plane_pose.size(); ++f) { const double* c = &d.plane_const[7 * f]; pk.add_plane(d.plane_pose[f], c, c + 3, c[6]); }
        for (int k = 0; k < K; ++k) if (d.pose_const[k] && d.sb_const[k]) pk.freeze_frame(k);

        vil_state st{K, d.L, &para_Pose[0][0], &para_SpeedBias[0][0], &para_Ex_Pose[0][0], &para_Td[0][0], para_Feature.data()};
        vil_options opt; vil_default_options(&opt);                          // NUM_ITERATIONS / SOLVER_TIME would be set here (:1404, :1411)
        double pose0_before[7]; for (int q = 0; q < 7; ++q) pose0_before[q] = para_Pose[0][q];
        int rc = vil_solve(vil_, pk.finish(), &st, &opt, &sum);              // ceres::Solve (:1414)
        if (rc != VIL_OK) return rc;
        rc = vil_gauge_fix(pose0_before, &st);                               // double2vector's yaw / position re-anchoring (:960-1011)
        if (rc != VIL_OK) return rc;
        vil_marg_spec ms{marginalization_flag == MARGIN_OLD ? VIL_MARGIN_OLD : VIL_MARGIN_SECOND_NEW, -1, -1, 4};
        rc = vil_marginalize(vil_, pk.finish(), &st, &opt, &ms, vil_prior_.out());   // MarginalizationInfo (:1486-1616 / :1624-1681)
        if (rc == VIL_OK) vil_prior_.commit();
        return rc;
    }

    double para_Pose[WINDOW_SIZE_MAX + 1][7], para_SpeedBias[WINDOW_SIZE_MAX + 1][9], para_Ex_Pose[1][7], para_Td[1][1];
    std::vector<double> para_Feature;
    MarginalizationFlag marginalization_flag = MARGIN_OLD;
    int status = 0;
    const vil_prior& prior() const { return vil_prior_.prior(); }

private:
    const Dump& dump_;
    const int WINDOW_SIZE;
    vil_ctx* vil_ = nullptr;
    vil::PriorStore vil_prior_;
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s window.bin\n", argv[0]); return 2; }
    Dump d;
    if (!load(argv[1], d)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    Estimator est(d);
    if (est.status != VIL_OK) { std::fprintf(stderr, "vil_create: %s\n", vil_strerror(est.status)); return 3; }      // no GPU: fail loudly, no CPU path
    vil_summary sum;
    const int rc = est.optimization(sum);
    if (rc != VIL_OK) { std::fprintf(stderr, "optimization: %s\n", vil_strerror(rc)); return 4; }
    std::printf("RESULT %d %d %.17g %.17g %d", sum.iterations, sum.termination, sum.initial_cost, sum.final_cost, est.prior().n);
    for (int q = 0; q < 7; ++q) std::printf(" %.17g", est.para_Pose[d.K - 1][q]);
    std::printf("\n");
    return 0;
}
