// vilsolve_shim.hpp -- header-only C++ packing shim between mVIL-Fusion's Estimator and the C-ABI.
//
// It reproduces the bookkeeping of Estimator::optimization() (estimator.cpp:1126-1398) -- which
// parameter blocks exist, which are constant, which residual blocks are added in which order --
// but instead of `new`-ing ceres cost functions it appends POD rows to the tables of a vil_problem.
// No Eigen / ceres / ROS types appear here: the caller passes the raw numbers it already has
// (para_* arrays, IntegrationBase members, feature observations); see INTEGRATION.md for the
// ~60-line patch of estimator.cpp that feeds it.
#ifndef VILSOLVE_SHIM_HPP
#define VILSOLVE_SHIM_HPP

#include <cstring>
#include <vector>

#include "vilsolve.h"

namespace vil {

class WindowPacker {
public:
    // K = WINDOW_SIZE + 1 frames (parameters.h:12), L = f_manager.getFeatureCount()
    WindowPacker(int K, int L) : K_(K), L_(L), pose_const_(K, 0), sb_const_(K, 0), lm_const_(L, 0) {
        std::memset(&prob_, 0, sizeof prob_);
        prob_.K = K; prob_.L = L;
        prob_.use_td = 1; prob_.sqrt_info_px = 460.0 / 2.0;      // FOCAL_LENGTH / 2 (estimator.cpp:18-19)
        prob_.q_lb[3] = 1.0;
    }
    // globals of parameters.h / the yaml
    void set_constants(const double G[3], double focal_length, double tr_over_row, bool estimate_extrinsic, bool estimate_td) {
        std::memcpy(prob_.G, G, sizeof prob_.G);
        prob_.sqrt_info_px = focal_length / 2.0;                  // estimator.cpp:18-19
        prob_.tr_over_row = tr_over_row;
        prob_.ex_const = estimate_extrinsic ? 0 : 1;              // estimator.cpp:1154-1158
        prob_.use_td = estimate_td ? 1 : 0;                       // estimator.cpp:1162-1166, 1204 vs 1233
        prob_.td_const = 0;
    }
    void set_lidar_extrinsic(const double q_lb_xyzw[4], const double t_lb[3]) { std::memcpy(prob_.q_lb, q_lb_xyzw, 32); std::memcpy(prob_.t_lb, t_lb, 24); }

    // estimator.cpp:1179-1186: one IMUFactor per consecutive frame pair.  `rec` = the 287 numbers of vilsolve.h
    // (delta_p/q/v, linearized_ba/bg, sum_dt, the five 3x3 jacobian blocks, covariance) read from IntegrationBase.
    void add_imu(int i, int j, const double rec[VIL_IMU_CONST]) { imu_i_.push_back(i); imu_j_.push_back(j); imu_.insert(imu_.end(), rec, rec + VIL_IMU_CONST); }

    // estimator.cpp:1189-1242: for every feature (in f_manager order, used_num >= 2 && start_frame < WINDOW_SIZE - 2),
    // for every observation after the first.  row_* already has ROW/2 subtracted (projection_td_factor.cpp:18-19).
    void add_visual(int imu_i, int imu_j, int feature_index, const double pts_i[3], const double pts_j[3], const double vel_i[2], const double vel_j[2],
                    double td_i, double td_j, double row_i, double row_j, bool lidar_depth_flag) {
        vis_i_.push_back(imu_i); vis_j_.push_back(imu_j); vis_l_.push_back(feature_index);
        const double rec[VIL_VIS_CONST] = {pts_i[0], pts_i[1], pts_i[2], pts_j[0], pts_j[1], pts_j[2], vel_i[0], vel_i[1], vel_j[0], vel_j[1], td_i, td_j, row_i, row_j};
        vis_.insert(vis_.end(), rec, rec + VIL_VIS_CONST);
        if (lidar_depth_flag) lm_const_[feature_index] = 1;       // estimator.cpp:1217-1221
    }
    // estimator.cpp:1371-1396 (constraint_mode == 3 after FindWindowsID) ; s = lidar_sqrt_info(0,0)
    void add_icp(int id_a, int id_b, int id_c, int id_d, double ta, double tb, double tc, double td, double ti, double tj, const double PIJ[3], double s) {
        const int ids[4] = {id_a, id_b, id_c, id_d}; icp_ids_.insert(icp_ids_.end(), ids, ids + 4);
        const double rec[VIL_ICP_CONST] = {ta, tb, tc, td, ti, tj, PIJ[0], PIJ[1], PIJ[2], s}; icp_.insert(icp_.end(), rec, rec + VIL_ICP_CONST);
    }
    // estimator.cpp:1354-1370 (constraint_mode == 4): zero velocity, freeze pose / speed-bias WINDOW_SIZE-1
    void freeze_frame(int k) { pose_const_[k] = 1; sb_const_[k] = 1; }
    // estimator.cpp:1298-1324 (bracket gap < 0.2 s) ; q = LPSq (x y z w) already moved lidar->body (:1289-1290)
    void add_lps(int id_l, int id_r, double tl, double tr, double tk, const double q_xyzw[4]) {
        lps_ids_.push_back(id_l); lps_ids_.push_back(id_r);
        const double rec[VIL_LPS_CONST] = {tl, tr, tk, q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3]}; lps_.insert(lps_.end(), rec, rec + VIL_LPS_CONST);
    }
    // extended mode: LiDAR point factors attached to window pose k, point in the LiDAR frame
    void add_edge(int k, const double cp[3], const double a[3], const double b[3]) { edge_pose_.push_back(k); edge_.insert(edge_.end(), cp, cp + 3); edge_.insert(edge_.end(), a, a + 3); edge_.insert(edge_.end(), b, b + 3); }
    void add_plane(int k, const double cp[3], const double n[3], double d) { plane_pose_.push_back(k); plane_.insert(plane_.end(), cp, cp + 3); plane_.insert(plane_.end(), n, n + 3); plane_.push_back(d); }
    // estimator.cpp:1171-1177: last_marginalization_info, stored as the vil_prior the previous vil_marginalize produced
    void set_prior(const vil_prior& pr) { prob_.prior = pr; }

    const vil_problem* finish() {
        prob_.pose_const = pose_const_.data(); prob_.sb_const = sb_const_.data(); prob_.lm_const = lm_const_.data();
        prob_.n_imu = (int)imu_i_.size(); prob_.imu_i = imu_i_.data(); prob_.imu_j = imu_j_.data(); prob_.imu_const = imu_.data();
        prob_.n_vis = (int)vis_i_.size(); prob_.vis_i = vis_i_.data(); prob_.vis_j = vis_j_.data(); prob_.vis_l = vis_l_.data(); prob_.vis_const = vis_.data();
        prob_.n_icp = (int)icp_ids_.size() / 4; prob_.icp_ids = icp_ids_.data(); prob_.icp_const = icp_.data();
        prob_.n_lps = (int)lps_ids_.size() / 2; prob_.lps_ids = lps_ids_.data(); prob_.lps_const = lps_.data();
        prob_.n_edge = (int)edge_pose_.size(); prob_.edge_pose = edge_pose_.data(); prob_.edge_const = edge_.data();
        prob_.n_plane = (int)plane_pose_.size(); prob_.plane_pose = plane_pose_.data(); prob_.plane_const = plane_.data();
        return &prob_;
    }

private:
    int K_, L_;
    vil_problem prob_;
    std::vector<uint8_t> pose_const_, sb_const_, lm_const_;
    std::vector<int32_t> imu_i_, imu_j_, vis_i_, vis_j_, vis_l_, icp_ids_, lps_ids_, edge_pose_, plane_pose_;
    std::vector<double> imu_, vis_, icp_, lps_, edge_, plane_;
};

// ---- timestamp -> window index mapping of the LiDAR constraints (lidar_backend.cpp:3-93) -------------
// Host logic, O(K).  `stamps[k]` = Headers[k].stamp.toSec() for the K = WINDOW_SIZE + 1 frames, ascending.

// lidar_backend.cpp:3-36: the two window frames bracketing time tl.  The reference sorts tl into the stamp list and
// takes the neighbours of its first occurrence, i.e. id_b = first frame with stamp >= tl, id_a = id_b - 1;
// rejected when tl lies before frame 0 or after the last frame.
inline bool find_nearest_2id(const double* stamps, int K, double tl, int& id_a, int& id_b) {
    int lb = 0;
    while (lb < K && stamps[lb] < tl) ++lb;
    id_a = lb - 1; id_b = lb;
    return id_b <= K - 1 && id_a >= 0;
}

// lidar_backend.cpp:38-93: window indices of the four frame stamps of an ICP constraint.  Stamps are matched with
// exact floating-point equality, as in the reference; an id whose stamp is not in the window KEEPS the value the caller
// passed in (the reference leaves it untouched), so initialise them to 0 the way estimator.cpp:1352 does.
inline bool find_windows_id(const double* stamps, int K, double ta, double tb, double tc, double td, int& id_a, int& id_b, int& id_c, int& id_d) {
    if (stamps[0] > ta || stamps[K - 1] < td || (tb - ta) > 0.5) return false;
    auto locate = [&](double t, int& id) { for (int k = 0; k < K; ++k) if (stamps[k] == t) { id = k; return; } };
    locate(ta, id_a); locate(tb, id_b); locate(tc, id_c); locate(td, id_d);
    if (id_b == id_c) { --id_a; --id_b; }       // scans i and j share a bracket frame: shift the first bracket down
    return id_b > id_a && id_d > id_c && id_a >= 0 && id_a != id_c;
}

// Owner of the caller-side storage of a prior across frames (what `last_marginalization_info` +
// `last_marginalization_parameter_blocks` are in estimator.h:146-147).
class PriorStore {
public:
    explicit PriorStore(int K) {
        vil_prior_capacity(K, &n_max_, &nblk_max_, &x0_max_);
        kind_.resize(nblk_max_); index_.resize(nblk_max_); col_.resize(nblk_max_);
        x0_.resize(x0_max_); J0_.resize((size_t)n_max_ * n_max_); r0_.resize(n_max_);
        std::memset(&out_, 0, sizeof out_);
        out_.blk_kind = kind_.data(); out_.blk_index = index_.data(); out_.blk_col = col_.data();
        out_.x0 = x0_.data(); out_.J0 = J0_.data(); out_.r0 = r0_.data();
        std::memset(&prior_, 0, sizeof prior_);
    }
    vil_prior_out* out() { return &out_; }
    // after a successful vil_marginalize: adopt the new prior (n == -1: keep the old one, estimator.cpp:1620)
    void commit() {
        if (out_.n < 0) return;
        kind_c_ = kind_; index_c_ = index_; col_c_ = col_; x0_c_ = x0_; J0_c_ = J0_; r0_c_ = r0_;
        prior_.n = out_.n; prior_.nblk = out_.nblk;
        prior_.blk_kind = kind_c_.data(); prior_.blk_index = index_c_.data(); prior_.blk_col = col_c_.data();
        prior_.x0 = x0_c_.data(); prior_.J0 = J0_c_.data(); prior_.r0 = r0_c_.data();
    }
    const vil_prior& prior() const { return prior_; }

private:
    int n_max_, nblk_max_, x0_max_;
    std::vector<int32_t> kind_, index_, col_, kind_c_, index_c_, col_c_;
    std::vector<double> x0_, J0_, r0_, x0_c_, J0_c_, r0_c_;
    vil_prior_out out_;
    vil_prior prior_;
};

}  // namespace vil
#endif
