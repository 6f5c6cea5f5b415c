/* vilpreint.h -- C-ABI of the IMU pre-integration of mVIL-Fusion's estimator.
 *
 * Replaces IntegrationBase::{push_back, propagate, midPointIntegration, repropagate} (factor/integration_base.h:30-158) for
 * a BATCH of intervals: interval k integrates the samples [start[k], start[k+1]) of the (dt, acc, gyr) streams from its
 * initial measurement (acc0, gyr0) with bias linearisation point (ba, bg) -- exactly what repropagate() recomputes for one
 * IntegrationBase, and what a sequence of push_back() calls accumulates.  All K-1 intervals of a window run concurrently.
 * Output per interval: the VIL_IMU_CONST (287) doubles vil_problem.imu_const takes (layout in vilsolve.h: delta_p, delta_q
 * [x y z w], delta_v, linearized_ba/bg, sum_dt, the five 3x3 Jacobian blocks, covariance 15x15) and, optionally, the full
 * 15x15 `jacobian` (row-major; order P R V BA BG as O_P..O_BG, parameters.h:80-87).  fp64 throughout, mid-point rule,
 * delta_q normalised after every sample (:153), F / V blocks as :91-120, noise = diag(ACC_N^2, GYR_N^2, ACC_N^2, GYR_N^2,
 * ACC_W^2, GYR_W^2) (x) I3 (:21-27).  Plain C, POD only, host pointers. */
#ifndef VILPREINT_H
#define VILPREINT_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vpre_ctx vpre_ctx;

int vpre_create(int32_t device, vpre_ctx** out);
void vpre_destroy(vpre_ctx* ctx);
/* noise4 = {ACC_N, GYR_N, ACC_W, GYR_W}; acc / gyr: 3 doubles per sample; acc0 / gyr0 / ba / bg: 3 doubles per interval;
 * imu_const: n x 287; jacobian: n x 225 or NULL.  An interval without samples yields the identity record (sum_dt = 0). */
int vpre_integrate(vpre_ctx* ctx, int32_t n, const int32_t* start, const double* dt, const double* acc, const double* gyr,
                   const double* acc0, const double* gyr0, const double* ba, const double* bg, const double* noise4,
                   double* imu_const, double* jacobian);
/* measurement hook (bench.py): HIP events on the library's stream around the kernel; returns launches and total ms, resets */
int vpre_profile_enable(vpre_ctx* ctx, int32_t enable);
int vpre_profile_read(vpre_ctx* ctx, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif
