/* vilmap.h -- C-ABI of the LiDAR scan-to-map registration of mVIL-Fusion's lidar_mapping node.
 *
 * Replaces, in lidar_mapping/src/localMapping.cpp, the body of the `iterCount < 2` loop (:594-791):
 *   vmap_set_map     kdtreeCornerFromMap->setInputCloud / kdtreeSurfFromMap->setInputCloud                     :590-591
 *   vmap_associate   per corner point: pointAssociateToMap (:170-179), 5-NN, line test by PCA (:613-660) -> LidarEdgeFactor(cp, a, b, 1.0)
 *                    per surf point: 10-NN re-ranked by |intensity difference| (:688-703), plane fit (:705-741) -> LidarPlaneNormFactor(cp, n, d)
 *   vmap_align       two rounds of {vmap_associate, 7-parameter solve with HuberLoss(0.1), DOGLEG, max 4 iterations (:596-600,:766-777)},
 *                    submitted to the GPU as ONE batch: per round a search, a fit, a compaction and a one-launch 6-dof dogleg
 *                    solve (the trust-region algorithm of vil_solve restricted to one pose block; csrc/vil_pose1.hpp); counts,
 *                    factor tables and the pose between the rounds stay on the device, one read-back ends the call.  The
 *                    factors are lidarFactor.hpp:12-55,106-138 in window-pose form with identity extrinsic.  `solver` is the
 *                    fallback for scans above 2^20 points (and the cross-check of the tests: environment variable
 *                    VIL_MAP_FUSED_MAX=<points>, read by vmap_create, 0 = always): vil_solve of include/vilsolve.h on a one-pose
 *                    window holding these factors -- same result, three launches per iteration.  opts->max_time_s is not
 *                    consulted by the one-launch solve (four iterations take tens of microseconds).  NOTE: the library's pose
 *                    update is right-multiplicative (q (x) dq, pose_local_parameterization.cpp) where the reference uses
 *                    ceres::EigenQuaternionParameterization here; both minimise the same cost, the four capped trust-region
 *                    iterations may land on slightly different iterates.
 * The neighbour searches are EXACT (same float distances as pcl's kd-tree; ties may order differently).
 * Points are float [x y z intensity] (PointXYZI).  Plain C, POD only, host pointers. */
#ifndef VILMAP_H
#define VILMAP_H
#include <stdint.h>
#include "vilsolve.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vmap_ctx vmap_ctx;

typedef struct vmap_summary {
    int32_t rounds;                   /* association + solve rounds executed (2) */
    int32_t n_edge, n_plane;          /* factors of the last round (corner_num, surf_num) */
    int32_t iterations;               /* trust-region iterations of the last solve */
    double initial_cost, final_cost;  /* of the last solve */
    double t_associate_ms, t_prepare_ms, t_solve_ms;  /* wall time over both rounds: "mapping data assosiation time" / "mapping solver time"
                                                         (localMapping.cpp:764,778); prepare = factor upload of vil_solve.  The single-
                                                         submission path cannot split them on the host clock: the whole call is in
                                                         t_solve_ms, the other two are 0 (per-kernel times: vmap_profile_read) */
} vmap_summary;

int vmap_create(int32_t device, vmap_ctx** out);
void vmap_destroy(vmap_ctx* ctx);
int vmap_set_map(vmap_ctx* ctx, int32_t n_corner, const float* corner_xyzi, int32_t n_surf, const float* surf_xyzi);
/* correspondences at pose (q = [x y z w], t): edge9 = n_edge x [cp a b], plane7 = n_plane x [cp n d], both in scan order;
 * capacities n_corner x 9 and n_surf x 7 doubles */
int vmap_associate(vmap_ctx* ctx, int32_t n_corner, const float* corner_xyzi, int32_t n_surf, const float* surf_xyzi,
                   const double* q_xyzw, const double* t, int32_t* n_edge, double* edge9, int32_t* n_plane, double* plane7);
/* localMapping.cpp:594-791: q_xyzw / t hold the initial guess (transformAssociateToMap) and receive the result */
int vmap_align(vmap_ctx* ctx, vil_ctx* solver, int32_t n_corner, const float* corner_xyzi, int32_t n_surf, const float* surf_xyzi,
               double* q_xyzw, double* t, const vil_options* opts, vmap_summary* out);
/* Scans of up to max_points points (default 2^20) take the one-launch 6-dof solve (k_pose_solve); larger ones -- and every scan when
 * max_points = 0 -- go through vil_solve on a one-pose window (the cross-check of the tests).  The one-launch solve is a persistent
 * kernel whose workgroups wait for one another: its grid is clamped to what the device holds at once and launches of such kernels are
 * serialised process-wide (csrc/vil_coop.hpp). */
int vmap_set_fused_max(vmap_ctx* ctx, int32_t max_points);
/* measurement hook (bench.py): HIP events on the library's stream around the two association kernels; read returns
 * {k_map_search, k_map_fit} launch counts and total durations and resets them */
int vmap_profile_enable(vmap_ctx* ctx, int32_t enable);
int vmap_profile_read(vmap_ctx* ctx, int64_t* launches2, double* total_ms2);

#ifdef __cplusplus
}
#endif
#endif
