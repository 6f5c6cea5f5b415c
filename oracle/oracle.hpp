// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle_math.hpp header).  PARITY UNPINNED.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "../include/vilsolve.h"
#include "oracle_math.hpp"

namespace orc {

struct Preint {
    double dp[3], dq[4], dv[3];
    double ba[3], bg[3];
    double acc0[3], gyr0[3];
    double noise[4];
    double sum_dt;
    double jac[225], cov[225];
};

// factors (oracle_factors.cpp)
bool cholesky_lower(int n, const double* A, double* Lo);
bool imu_sqrt_info(const double* cov, double* U);
void imu_evaluate(const double* c, const double* G3, const double* pi, const double* sbi, const double* pj, const double* sbj, double* r, double* J);
void visual_evaluate(const double* c, double sqrt_info, double tr_over_row, int use_td, const double* pi, const double* pj, const double* ex, double inv_dep, double td, double* r, double* J);
void prior_dx(const vil_prior& pr, const double* const* params, double* dx);
void prior_evaluate(const vil_prior& pr, const double* const* params, double* r, double* J);
void icp_evaluate(const double* c, const double* pa, const double* pb, const double* pc, const double* pd, double* r, double* J);
void lps_evaluate(const double* c, const double* pa, const double* pb, double* r, double* J);
void edge_evaluate(const double* c, const double* q_lb, const double* t_lb, const double* pose, double* r, double* J);
void plane_evaluate(const double* c, const double* q_lb, const double* t_lb, const double* pose, double* r, double* J);
void plane3_evaluate(const double* c12, const double* q_lb, const double* t_lb, const double* pose, double* r, double* J);      // LidarPlaneFactor   lidarFactor.hpp:57-104
void distance_evaluate(const double* c6, const double* q_lb, const double* t_lb, const double* pose, double* r, double* J);    // LidarDistanceFactor :141-172
void edge_residual_ref(const double* cp, const double* a3, const double* b3, const double* q_wl_xyzw, const double* t_wl, double* r);
void plane_residual_ref(const double* cp, const double* n3, double d, const double* q_wl_xyzw, const double* t_wl, double* r);
void loss_evaluate(int kind, double a, double s, double rho[3]);
double apply_corrector(int kind, double a, int nr, double* r, int nblocks, double* const* Jb, const int* ncols);
void preint_init(Preint& s, const double* acc0, const double* gyr0, const double* ba, const double* bg, const double noise4[4]);
void preint_push(Preint& s, double dt, const double* acc1, const double* gyr1);
void preint_pack(const Preint& s, double* c);

// canonical reduced ("camera") ordering used by oracle AND library:
//   pose k -> 6k ; ex -> 6K ; td -> 6K+6 ; speedbias k -> 6K+7+9k ;  D = 15K+7
struct Layout {
    int K, L, D;
    explicit Layout(int K_, int L_) : K(K_), L(L_), D(15 * K_ + 7) {}
    int pose(int k) const { return 6 * k; }
    int ex() const { return 6 * K; }
    int td() const { return 6 * K + 6; }
    int sb(int k) const { return 6 * K + 7 + 9 * k; }
};

}  // namespace orc
