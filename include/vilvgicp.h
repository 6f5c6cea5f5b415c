/* vilvgicp.h -- C-ABI of the scan-to-scan voxelised GICP registration that produces mVIL-Fusion's LiDAR ICP constraint
 * (the first "next" row after the sliding-window solve).
 *
 * Replaces, in vils_estimator's Estimator::processLidar (estimator.cpp:269-300), the object
 *     fast_gicp::FastVGICP<POINT, POINT> gicp;  gicp.setResolution(0.5); gicp.setInputSource(..); gicp.setInputTarget(..); gicp.align(.., guess)
 * i.e. the third-party fast_gicp sources vendored under vils_estimator/src/lidar_functions/fast_gicp:
 *   vgicp_set_target      GaussianVoxelMap::create_voxelmap            gicp/fast_vgicp_voxel.hpp:128-159 (ADDITIVE voxels :107-125)
 *   vgicp_linearize       FastVGICP::update_correspondences + linearize gicp/impl/fast_vgicp_impl.hpp:73-170
 *   vgicp_compute_error   FastVGICP::compute_error                      gicp/impl/fast_vgicp_impl.hpp:173-196
 *   vgicp_align           LsqRegistration::computeTransformation / step_lm / step_gn / is_converged   gicp/impl/lsq_registration_impl.hpp:48-165
 *   vgicp_covariances     FastGICP::calculate_covariances               gicp/impl/fast_gicp_impl.hpp:241-300 (k = 20, PLANE)
 * Per-point covariances are row-major 3x3 (the upper-left block of fast_gicp's 4x4 matrices); vgicp_set_source /
 * vgicp_set_target accept cov9 = NULL and then compute them with vgicp_covariances' kernel.
 * Plain C, POD only, host pointers; fp64 arithmetic on points given as float xyz (PCL points are float, cast to double
 * exactly like getVector4fMap().cast<double>()). */
#ifndef VILVGICP_H
#define VILVGICP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vgicp_ctx vgicp_ctx;

enum { VGICP_DIRECT1 = 1, VGICP_DIRECT7 = 7, VGICP_DIRECT27 = 27 };   /* NeighborSearchMethod, gicp_settings.hpp:8 */
enum { VGICP_LM = 0, VGICP_GN = 1 };                                   /* LSQ_OPTIMIZER_TYPE */

typedef struct vgicp_options {
    int32_t neighbor_mode;            /* DIRECT1 (FastVGICP ctor default, fast_vgicp_impl.hpp:23) */
    int32_t optimizer;                /* LevenbergMarquardt (lsq_registration_impl.hpp:17) */
    int32_t max_iterations;           /* 64 */
    int32_t lm_max_iterations;        /* 10 */
    double rotation_epsilon;          /* 2e-3 */
    double transformation_epsilon;    /* 5e-4 */
    double lm_init_lambda_factor;     /* 1e-9 */
} vgicp_options;

typedef struct vgicp_summary {
    int32_t iterations;               /* outer iterations executed (nr_iterations_ + 1) */
    int32_t converged;
    int32_t n_correspondences;        /* of the last linearisation */
    int32_t lm_failed;                /* step_lm ran out of lm_max_iterations ("lm not converged!!") */
    double final_error;               /* y0 of the last linearisation */
    double final_hessian[36];         /* LsqRegistration::final_hessian_ */
} vgicp_summary;

int vgicp_create(int32_t device, vgicp_ctx** out);
void vgicp_destroy(vgicp_ctx* ctx);
void vgicp_default_options(vgicp_options* o);
/* target cloud + covariances -> Gaussian voxel map of edge `resolution` (estimator.cpp:271 uses 0.5) */
int vgicp_set_target(vgicp_ctx* ctx, int32_t n, const float* xyz, const double* cov9, double resolution);
int vgicp_set_source(vgicp_ctx* ctx, int32_t n, const float* xyz, const double* cov9);
/* Covariance of the k nearest neighbours of every point (the point itself included, as pcl's nearestKSearch returns it),
 * regularised like RegularizationMethod::PLANE: singular values replaced by (1, 1, 1e-3).  The reference finds the
 * neighbours with a kd-tree in float; this is an EXACT search with the same float distances (ties may order differently).
 * out_cov9: n x 9 host buffer. */
int vgicp_covariances(vgicp_ctx* ctx, int32_t n, const float* xyz, int32_t k, double* out_cov9);
/* neighbour search of vgicp_covariances: clouds of at least min_points points (default 4096) use the uniform-grid search with an initial
 * cell edge `cell` (default 1.0 m, adapted to ~8 points per cell), smaller ones the tiled exhaustive search; both are exact. */
int vgicp_set_knn_grid(vgicp_ctx* ctx, int32_t min_points, double cell);
/* T: row-major 4x4 isometry (source -> target).  Recomputes the correspondences and their Mahalanobis matrices at T.
 * H (6x6 row-major, [rotation | translation] order), b (6) may both be NULL (error only). */
int vgicp_linearize(vgicp_ctx* ctx, const double* T, int32_t neighbor_mode, double* err, double* H, double* b, int32_t* n_corr);
/* error at T with the correspondences / Mahalanobis matrices of the LAST vgicp_linearize (as the reference) */
int vgicp_compute_error(vgicp_ctx* ctx, const double* T, double* err);
/* One launch per alignment: a persistent kernel whose workgroups wait for one another.  Its grid is clamped to what the device holds at
 * once, launches of such kernels are serialised process-wide (csrc/vil_coop.hpp), and a device that cannot hold it takes one launch per pass. */
int vgicp_align(vgicp_ctx* ctx, const double* guess, const vgicp_options* opts, double* T_out, vgicp_summary* out);
/* profiling aid (bench.py): with enable != 0 every vgicp_linearize brackets its main kernel with HIP events on the library's
 * own stream; vgicp_profile_read returns the number of timed launches and their total duration, and resets both. */
int vgicp_profile_enable(vgicp_ctx* ctx, int32_t enable);
int vgicp_profile_read(vgicp_ctx* ctx, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif
