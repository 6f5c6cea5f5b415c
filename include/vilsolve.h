/*
 * vilsolve.h -- C-ABI of the MI355X-native sliding-window factor-graph backend.
 *
 * This is the drop-in boundary for ONE hot path of mVIL-Fusion:
 *     void Estimator::optimization()        vils_estimator/src/estimator.cpp:1124-1687
 * and for the per-factor surface it is built on,
 *     bool ceres::CostFunction::Evaluate(double const* const* parameters,
 *                                        double* residuals, double** jacobians) const
 *     (imu_factor.h:19, projection_td_factor.cpp:34, projection_factor.cpp:21,
 *      marginalization_factor.cpp:352, lidar_backend.h:45/107, lidar_mapping/src/lidarFactor.hpp:12-138).
 *
 * Everything is plain-old-data, caller-owned, fp64 HOST pointers.  No pointer-identity semantics:
 * the reference's "address -> parameter block" maps (marginalization_factor.cpp:100,106) are
 * replaced by explicit (kind, index) block ids.
 *
 * Conventions (all from the reference):
 *   pose block      = [px py pz qx qy qz qw]           estimator.cpp:920-927 (Hamilton, body->world)
 *   speed-bias block= [vx vy vz bax bay baz bgx bgy bgz]  estimator.cpp:929-939
 *   tangent update  = p += d[0:3]; q = normalize(q (x) [1, d[3:6]/2])   pose_local_parameterization.cpp:3-18
 *   residual order of the IMU factor = [dp dtheta dv dba dbg]  parameters.h:80-87
 *   all Jacobian blocks are ROW-MAJOR, num_residuals x GLOBAL block size (7 for poses: the 7th
 *   column is 0 for analytic factors, raw d/dqw for the two AutoDiff factors).
 */
#ifndef VILSOLVE_H
#define VILSOLVE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIL_ABI_VERSION 1

/* ---- sizes (parameters.h:73-78) ------------------------------------------------------------ */
#define VIL_SIZE_POSE      7
#define VIL_SIZE_SPEEDBIAS 9
#define VIL_IMU_CONST      287   /* doubles per IMU factor, layout below */
#define VIL_VIS_CONST      14    /* doubles per visual factor, layout below */
#define VIL_EDGE_CONST     9     /* cp(3) a(3) b(3)            lidarFactor.hpp:12-55  */
#define VIL_PLANE_CONST    7     /* cp(3) n(3) d               lidarFactor.hpp:106-138 */
#define VIL_ICP_CONST      10    /* ta tb tc td ti tj PIJ(3) s lidar_backend.h:97-184 */
#define VIL_LPS_CONST      7     /* tl tr tk q(x y z w)        lidar_backend.h:35-95  */
#define VIL_MAX_TRACE      64

/* IMU constants, per factor (integration_base.h members consumed by imu_factor.h:19-181):
 *   [0:3)   delta_p        [3:7) delta_q (x y z w)   [7:10) delta_v
 *   [10:13) linearized_ba  [13:16) linearized_bg     [16]   sum_dt
 *   [17:26) dp_dba  [26:35) dp_dbg  [35:44) dq_dbg  [44:53) dv_dba  [53:62) dv_dbg   (3x3 row-major)
 *   [62:287) covariance 15x15 row-major (symmetric)
 * Visual constants, per factor (projection_td_factor.cpp:6-19; row already has ROW/2 subtracted):
 *   [0:3) pts_i (z=1)  [3:6) pts_j  [6:8) vel_i  [8:10) vel_j  [10] td_i  [11] td_j  [12] row_i  [13] row_j
 */

typedef enum {
    VIL_OK = 0,
    VIL_ERR_INVALID_ARGUMENT = -1,
    VIL_ERR_DEVICE = -2,           /* HIP runtime error / no device / extension missing */
    VIL_ERR_NON_FINITE = -3,       /* NaN/Inf in cost or step; state left unchanged      */
    VIL_ERR_NOT_POSITIVE_DEFINITE = -4, /* reduced system not PD after max mu             */
    VIL_ERR_COMM = -5,             /* RCCL failure                                        */
    VIL_ERR_UNSUPPORTED = -6
} vil_status;

typedef enum {
    VIL_FACTOR_IMU = 0,     /* A4  imu_factor.h:19                   r 15, J 15x7|15x9|15x7|15x9 */
    VIL_FACTOR_VISUAL = 1,  /* A6/A7 projection(_td)_factor.cpp      r 2,  J 2x7|2x7|2x7|2x1|2x1 */
    VIL_FACTOR_PRIOR = 2,   /* A8  marginalization_factor.cpp:352    r n,  J n x sum(global)     */
    VIL_FACTOR_ICP = 3,     /* A11 lidar_backend.h:107               r 3,  J 3x7 x4              */
    VIL_FACTOR_LPS = 4,     /* A12 lidar_backend.h:45                r 3,  J 3x7 x2              */
    VIL_FACTOR_EDGE = 5,    /* A14 lidarFactor.hpp:12  (window form) r 3,  J 3x7                 */
    VIL_FACTOR_PLANE = 6,   /* A14 lidarFactor.hpp:106 (window form) r 1,  J 1x7                 */
    VIL_FACTOR_NCLASS = 7
} vil_factor_class;

/* doubles of residual / Jacobian written per factor by vil_eval_factors (prior: n, n*sum_global) */
#define VIL_IMU_NR 15
#define VIL_IMU_NJ (15 * (7 + 9 + 7 + 9))
#define VIL_VIS_NR 2
#define VIL_VIS_NJ (2 * (7 + 7 + 7 + 1 + 1))
#define VIL_ICP_NR 3
#define VIL_ICP_NJ (3 * 7 * 4)
#define VIL_LPS_NR 3
#define VIL_LPS_NJ (3 * 7 * 2)
#define VIL_EDGE_NR 3
#define VIL_EDGE_NJ (3 * 7)
#define VIL_PLANE_NR 1
#define VIL_PLANE_NJ 7

typedef enum { VIL_BLK_POSE = 0, VIL_BLK_SPEEDBIAS = 1, VIL_BLK_EX = 2, VIL_BLK_TD = 3 } vil_block_kind;
typedef enum { VIL_LOSS_NONE = 0, VIL_LOSS_CAUCHY = 1, VIL_LOSS_HUBER = 2 } vil_loss_kind;
typedef enum { VIL_MARGIN_OLD = 0, VIL_MARGIN_SECOND_NEW = 1 } vil_marg_flag;

typedef enum {
    VIL_TERM_NONE = 0,
    VIL_TERM_FUNCTION_TOLERANCE = 1,
    VIL_TERM_GRADIENT_TOLERANCE = 2,
    VIL_TERM_PARAMETER_TOLERANCE = 3,
    VIL_TERM_MAX_ITERATIONS = 4,
    VIL_TERM_MAX_TIME = 5,
    VIL_TERM_FAILURE = 6
} vil_termination;

/* ---- state: the para_* arrays of estimator.h:110-116 ---------------------------------------- */
typedef struct vil_state {
    int32_t K;            /* frames in the window (reference: WINDOW_SIZE+1 = 7)  */
    int32_t L;            /* landmarks (reference: f_manager.getFeatureCount())   */
    double* pose;         /* K x 7   para_Pose                                    */
    double* speedbias;    /* K x 9   para_SpeedBias                               */
    double* ex_pose;      /* 7       para_Ex_Pose[0]  (tic, qic)                  */
    double* td;           /* 1       para_Td[0]                                   */
    double* inv_depth;    /* L       para_Feature                                 */
} vil_state;

/* ---- marginalisation prior: MarginalizationInfo's linearised output ------------------------- */
typedef struct vil_prior {
    int32_t n;                 /* residual count = sum of LOCAL sizes of kept blocks (marginalization_factor.cpp:199) */
    int32_t nblk;              /* kept parameter blocks (keep_block_size.size())                */
    const int32_t* blk_kind;   /* nblk  vil_block_kind                                          */
    const int32_t* blk_index;  /* nblk  frame index for POSE / SPEEDBIAS (already shifted i->i-1)*/
    const int32_t* blk_col;    /* nblk  first column of the block in J0 (keep_block_idx - m)    */
    const double* x0;          /* concatenated GLOBAL-size linearisation points (keep_block_data)*/
    const double* J0;          /* n x n COLUMN-major  linearized_jacobians                      */
    const double* r0;          /* n                  linearized_residuals                       */
} vil_prior;

/* ---- one window's factor graph (everything estimator.cpp:1126-1398 hands to ceres::Problem) -- */
typedef struct vil_problem {
    int32_t K, L;
    /* constancy (ceres SetParameterBlockConstant; estimator.cpp:1157,1220,1238,1369-1370) */
    const uint8_t* pose_const;      /* K or NULL */
    const uint8_t* sb_const;        /* K or NULL */
    const uint8_t* lm_const;        /* L or NULL   (lidar_depth_flag) */
    int32_t ex_const;               /* !ESTIMATE_EXTRINSIC */
    int32_t td_const;               /* 1 => td not optimised */
    int32_t use_td;                 /* 1: ProjectionTdFactor (A6), 0: ProjectionFactor (A7) */

    /* IMU factors (estimator.cpp:1179-1186); factors with sum_dt > 10 are skipped by the library */
    int32_t n_imu;
    const int32_t* imu_i;           /* n_imu  frame i (j = imu_j) */
    const int32_t* imu_j;           /* n_imu */
    const double* imu_const;        /* n_imu x VIL_IMU_CONST */

    /* visual factors, grouped by landmark in landmark order (estimator.cpp:1189-1242) */
    int32_t n_vis;
    const int32_t* vis_i;           /* n_vis  anchor frame (start_frame)   */
    const int32_t* vis_j;           /* n_vis  observing frame              */
    const int32_t* vis_l;           /* n_vis  landmark (feature_index), non-decreasing */
    const double* vis_const;        /* n_vis x VIL_VIS_CONST */

    /* prior (estimator.cpp:1171-1177); prior.n == 0 => none */
    vil_prior prior;

    /* LiDAR scan-to-scan relative constraints, constraint_mode==3 only (estimator.cpp:1371-1396) */
    int32_t n_icp;
    const int32_t* icp_ids;         /* n_icp x 4  (a,b,c,d) */
    const double* icp_const;        /* n_icp x VIL_ICP_CONST */
    /* LiDAR local-map rotation priors (estimator.cpp:1298-1324) */
    int32_t n_lps;
    const int32_t* lps_ids;         /* n_lps x 2  (l,r) */
    const double* lps_const;        /* n_lps x VIL_LPS_CONST */

    /* extended mode: point-level LiDAR factors attached to window poses */
    int32_t n_edge;
    const int32_t* edge_pose;       /* n_edge */
    const double* edge_const;       /* n_edge x VIL_EDGE_CONST, cp in the LiDAR frame */
    int32_t n_plane;
    const int32_t* plane_pose;      /* n_plane */
    const double* plane_const;      /* n_plane x VIL_PLANE_CONST */
    double q_lb[4];                 /* RLB as quaternion (x y z w): p_l = RLB p_b + TLB (estimator.cpp:1289-1290) */
    double t_lb[3];                 /* TLB */

    /* constants from parameters.h / the yaml */
    double G[3];                    /* gravity (0,0,g)   parameters.cpp:32,103            */
    double sqrt_info_px;            /* FOCAL_LENGTH/2 = 230   estimator.cpp:18-19         */
    double tr_over_row;             /* TR/ROW (0 when not rolling shutter)                */
} vil_problem;

/* ---- solver options: ceres::Solver::Options as used at estimator.cpp:1400-1411 --------------- */
typedef struct vil_options {
    int32_t max_iterations;         /* NUM_ITERATIONS (30)                                 */
    double max_time_s;              /* SOLVER_TIME (0.05); <= 0 disables (parity runs); ignored by sharded (multi-rank) solves.  The host reads the clock between two
                                     * chunks of enqueued iterations (3 - 15 of them): the cap takes effect at a chunk boundary, never in the middle of one -- ceres reads
                                     * it after every iteration.  At ~75 us per iteration the reference's 0.05 s is 600 iterations away; its 30-iteration cap always comes first */
    double function_tolerance;      /* 1e-6  */
    double gradient_tolerance;      /* 1e-10 */
    double parameter_tolerance;     /* 1e-8  */
    double initial_radius;          /* 1e4   */
    double max_radius;              /* 1e16  */
    double min_relative_decrease;   /* 1e-3  */
    double min_mu, max_mu;          /* 1e-8, 1 */
    int32_t jacobi_scaling;         /* 1     */
    int32_t visual_loss;  double visual_loss_scale;   /* Cauchy 1.0  estimator.cpp:1129 */
    int32_t lidar_loss;   double lidar_loss_scale;    /* Huber 0.1   localMapping.cpp:597 */
    int32_t rel_loss;     double rel_loss_scale;      /* ICP/LPS: Cauchy 1.0              */
    int32_t autodiff_quirk;         /* 1 (faithful): ICP/LPS pose Jacobian = raw d/d(qx,qy,qz) */
    int32_t precision;              /* 0: fp64 everywhere; 1: fp32 factor evaluation, fp64 accumulation */
} vil_options;

typedef struct vil_summary {
    int32_t iterations;             /* trust-region iterations executed (successful + unsuccessful) */
    int32_t successful_steps;
    int32_t termination;            /* vil_termination */
    double initial_cost, final_cost;
    double t_prepare_ms, t_solve_ms, t_readback_ms;   /* phase timings (estimator.cpp:1168,1412-1417) */
    double cost_trace[VIL_MAX_TRACE];                 /* cost after each iteration   */
    double radius_trace[VIL_MAX_TRACE];
} vil_summary;

/* ---- marginalisation request (estimator.cpp:1484-1683) --------------------------------------- */
typedef struct vil_marg_spec {
    int32_t flag;                   /* vil_marg_flag */
    /* MARGIN_OLD only: the ICP / LPS constraint touching frame 0 (-1 = none; estimator.cpp:1311-1317,1381-1389) */
    int32_t icp_marg;               /* index into problem.icp_* */
    int32_t lps_marg;               /* index into problem.lps_* */
    int32_t threads;                /* CPU oracle only: NUM_THREADS (4) */
} vil_marg_spec;

/* caller-provided storage for the new prior; sizes from vil_prior_capacity(K) */
typedef struct vil_prior_out {
    int32_t n, nblk, m;             /* m = marginalised dimension (diagnostic) */
    int32_t* blk_kind;              /* capacity nblk_max */
    int32_t* blk_index;
    int32_t* blk_col;
    double* x0;                     /* capacity 7*K + 9 + 7 + 1 ... see vil_prior_capacity */
    double* J0;                     /* n_max x n_max, written n x n column-major.  A square root of the marginal information:
                                     * J0^T J0 = A, J0^T r0 = b (marginalization_factor.cpp:313-314).  The reference takes
                                     * sqrt(S) V^T of an eigen-decomposition (basis implementation-defined); the library
                                     * returns the transposed pivoted-Cholesky factor -- the same prior cost, gradient and
                                     * Gauss-Newton matrix */
    double* r0;
    double* A;                      /* optional n_max x n_max: reduced information matrix (J0^T J0 up to eps-truncation) */
    double* b;                      /* optional n_max */
} vil_prior_out;

typedef struct vil_device_cfg {
    int32_t device;                 /* HIP device ordinal                         */
    int32_t rank, world;            /* data-parallel shard of the factor set      */
    int32_t reserved;
} vil_device_cfg;

typedef struct vil_ctx vil_ctx;

/* ---- entry points ----------------------------------------------------------------------------- */
int vil_abi_version(void);
const char* vil_strerror(int status);

/* context: device buffers, stream, (optional) RCCL communicator. Non-reentrant per ctx
 * (the reference never re-enters optimization(): estimator_node.cpp:388,352-355).
 * Window size: K <= 20 frames (reduced dimension 15 K + 7 <= 307: the step kernel's LDS work space; the reference runs K = 7,
 * parameters.h:12); larger windows return VIL_ERR_UNSUPPORTED.  Per landmark at most 128 observations, at most 12 ICP + LPS
 * constraints (the reference trims to 5 + 7), prior dimension <= 136 for vil_marginalize. */
int vil_create(const vil_device_cfg* cfg, vil_ctx** out);
void vil_destroy(vil_ctx* ctx);

/* multi-GPU: rank 0 calls vil_comm_unique_id, the 128 bytes are broadcast by the host launcher
 * (torch.distributed / MPI / anything), every rank calls vil_comm_init.  Afterwards vil_upload keeps this rank's shard of the
 * factor set (vil_shard_ranges) and every trust-region iteration performs ONE all-reduce of the linear-system set; all ranks
 * return the same state bit for bit.  Every rank must make the same sequence of calls; a failing upload is reported on all. */
int vil_comm_unique_id(void* id128);
int vil_comm_init(vil_ctx* ctx, const void* id128, int rank, int world);
/* Communicator for contexts that live in ONE process (one host thread per context; <= 8): the same sharding and the
 * same per-iteration reductions as vil_comm_init, summed through device memory instead of RCCL.  Also lets a sharded
 * solve be exercised on a single device (tests). */
int vil_comm_init_local(vil_ctx** ctxs, int n);
/* Multi-process exchange WITHOUT RCCL (one process per GPU, or -- tests -- several processes on one GPU): the per-iteration message is
 * latency-bound, and on the fully connected xGMI mesh "everybody writes its message into everybody's inbox, then sums locally in rank
 * order" is one hop instead of a ring.  vil_comm_ipc_export allocates this rank's inbox (world x 2 x max_doubles doubles: size it for the
 * largest window, vil_reduced_dim(K)^2 + 3 D + 4 + 17 L + 6 F doubles) and returns its 64-byte IPC handle; the launcher gathers the
 * `world` handles in rank order (any transport) and every rank calls vil_comm_ipc_init.  Same sharding, same call sequence and the same
 * bit-identical results on every rank as with vil_comm_init; the collective is three small launches in stream order. */
/* What travels per trust-region iteration through these exchanges (and vil_comm_init_local): of the set [S' | g | cost | landmark arrays] the lower
 * triangle of S', the vectors and THIS RANK's slice of the landmark arrays (a landmark's entries are non-zero on its owner only: the sum over ranks is
 * the owner's value) -- 103 + 380 / world kB per peer at K = 10 / 1000 landmarks instead of 484 kB.  vil_comm_message_bytes reports it for the
 * uploaded (sharded) window. */
int vil_comm_message_bytes(vil_ctx* ctx, int64_t* bytes_per_peer, int64_t* bytes_full_set);
/* What the context's communicator is: its rank, the number of ranks it spans and the transport of the per-iteration collective (0: none, 1: RCCL,
 * 2: in-process, 3: peer buffers; -3: peer buffers exported but not yet initialised).  A launcher asserts world == the ranks it started. */
int vil_comm_info(vil_ctx* ctx, int32_t* rank, int32_t* world, int32_t* transport);
/* RCCL (vil_comm_init, world <= 8) moves the same content as ONE ncclAllReduce of the packed camera part [lower(S') | g | b_c | diag | cost] plus ONE ncclAllGather
 * of the owners' landmark slices (padded to the largest): 103 + 48 kB per rank at K = 10 / 1000 landmarks / world 8; vil_comm_message_bytes reports that sum.
 * test hook: the pack / unpack kernels of that path over the in-process communicator (two small kernels stand in for the RCCL calls), so that they run on
 * 2 / 3 / 8 ranks of one device. */
int vil_debug_set_slim_emul(vil_ctx* ctx, int32_t on);
int vil_comm_ipc_export(vil_ctx* ctx, int rank, int world, size_t max_doubles, void* handle64);
int vil_comm_ipc_init(vil_ctx* ctx, const void* handles /* world x 64 bytes */);
/* test hook: run the multi-GPU plumbing (partial system in set 0, the collective sums it into set 1, step kernel on set 1) on a
 * single rank, with or without a 1-rank communicator.  Invalidates the resident window. */
int vil_debug_set_split(vil_ctx* ctx, int32_t on);
/* test hook: which launch structure a single-GPU solve takes.  0 (default): the library's choice -- the whole iteration (sweep, gather, chain elimination, step)
 * in ONE launch whenever the device holds its waiting workgroups and the roles share one dynamic-LDS size (every BASELINE size on an MI355X), else the next one
 * down this list; 3: sweep launch + gather / step launch (round 4's structure); 1: the fallback for devices / windows where it does not: separate gather launch, the speed-bias chain
 * eliminated by a workgroup of the SWEEP launch; 2: no chain workgroup at all (the step kernel eliminates the chain itself, the round-2 structure).
 * 4: one launch per ITERATION (k_iter) also where the whole solve could run as one resident launch (k_solve: the default for windows whose every role fits the
 * device at once -- configs[1]; same bits as mode 4).
 * Same results to rounding in every mode.  Invalidates the resident window. */
int vil_debug_set_launch_mode(vil_ctx* ctx, int32_t mode);
/* test hook: the next n hipGraph captures of this context's solves are treated as failed (as a driver that cannot capture or instantiate the chunk would make them).
 * A failed capture is not an error: nothing has run yet, the solve at hand and every later solve of the context launch directly.  n = 0 re-arms graph replay. */
int vil_debug_fail_graph_capture(vil_ctx* ctx, int32_t n);
/* n resident windows (one context each, all on one device, each uploaded by vil_upload / vil_solve) solved CONCURRENTLY: a stream and a host thread per context, one
 * launch per trust-region iteration each.  One window keeps a single master workgroup busy for two thirds of an iteration; n of them interleave on the device -- what a
 * server that tracks several sessions on one GPU runs, and the single-GPU form of the `replicas` leg of bench.py --gpus N.  statuses[i] / summaries[i]: as
 * vil_solve_resident of context i; every window's result is bit-equal to its solo solve under vil_debug_set_launch_mode(4).  More windows than the device holds
 * waiting workgroups for are solved in groups: the waiting workgroups of a group fit the device per XCD and take at most half of it, counting the launches the runtime
 * runs together (its hardware queues: four, or GPU_MAX_HW_QUEUES).  Returns the first non-zero status. */
int vil_solve_batch(vil_ctx** ctxs, int32_t n, const vil_options* options, vil_summary* summaries, int32_t* statuses);
/* Recovery of a one-launch solve whose workgroups could not all run together.  The one-launch iteration's roles wait for one another inside the launch; every such
 * wait is bounded by TIME (50 ms of the device's 100 MHz wall clock).  A wait that gives up ends the launches at once; vil_solve_resident / vil_solve / vil_win_solve
 * then put the resident state back to what the solve started from and run the SAME solve again with two launches per iteration (sweep, then gather + step: mode 3 of
 * vil_debug_set_launch_mode -- same results to rounding), and return its result.  Only if that attempt gives up too is VIL_ERR_DEVICE returned -- with the resident
 * state (and the caller's vil_state) as the solve found them.  The next solve takes the one-launch structure again.
 * vil_recovery_counts: solves of this context that were re-run that way / that failed on both structures.
 * test hook vil_debug_drop_flag: in launch `launch` (0-based) of the NEXT solve, sweep role `role` (workgroup index in the sweep's order [imu | prior | rel | visual |
 * plane | edge]; -2 - g: gather workgroup g) does not post its completion flag -- what a workgroup that never became resident looks like to the ones waiting for it.
 * launch | 0x10000: a gather workgroup's flag is lost in the retry as well (a solve that fails on both structures). */
int vil_debug_drop_flag(vil_ctx* ctx, int32_t role, int32_t launch);
int vil_recovery_counts(vil_ctx* ctx, int64_t* recovered, int64_t* failed);
/* profiling (with vil_profile_enable(ctx, 1), one-launch iterations): times == NULL arms it -- from now on every workgroup of launch `launch` (0-based) of a solve
 * leaves its entry and exit time (100 MHz device clock) --; with times != NULL the pairs {entry, exit} of the first max_workgroups (<= 4096) workgroups of the last
 * recorded launch are copied out, in block-index order = the launch's grid order [imu | prior | rel][chain][visual | plane | edge][master | helpers | tiles][gather]
 * (zeros: a workgroup that did not run).  tools/probe_workgroups.py prints them by role. */
int vil_profile_workgroups(vil_ctx* ctx, int32_t launch, uint64_t* times, int32_t max_workgroups);
/* test hook: the step's dense solve on a matrix of the caller's -- A is (D + 1) x (D + 1) row major, its lower triangle the SPD matrix, its last row the right-hand
 * side (D <= 159).  L receives the Cholesky factor (lower, row major, last row = L^-1 rhs), x the solution, *ok 0 when a pivot was not positive.  variant 1: what the
 * one-launch iteration runs (16-wide panels factored a matrix row per lane, back substitution a column per lane: vil_step.hpp chol_rowwave / back_subst_cols);
 * variant 0: the look-ahead factorisation (4-wide panels) and the back substitution through inverted diagonal tiles that the other launch structures run. */
int vil_debug_dense_solve(vil_ctx* ctx, int32_t D, const double* A, double* L, double* x, int32_t* ok, int32_t variant);
/* what the uploaded window's solves launch per trust-region iteration: 0 (nothing: the whole solve is ONE resident launch, k_solve -- windows whose roles all fit the
 * device at once; *one_launch = 1 as well), 1 (the one-launch iteration), 2 (sweep + gather / step) or 3 (sweep, gather, step) */
int vil_debug_get_launch_structure(vil_ctx* ctx, int32_t* launches_per_iteration, int32_t* one_launch);

/* replaces estimator.cpp:1126-1419 (build ceres::Problem ... ceres::Solve): state is updated in
 * place on success and left UNCHANGED on any error. */
int vil_solve(vil_ctx* ctx, const vil_problem* problem, vil_state* state_inout,
              const vil_options* options, vil_summary* summary);

/* resident variant: upload once, solve many times without host<->device traffic of the factor tables.
 * vil_solve_resident returns when the summary is known: the launch that finishes the solve leaves its trust-region record in pinned host
 * memory and the host polls it, so the tail of the stream (launches that find the solve finished, the copy of the accepted state,
 * the gauge fix) may still be draining.  Everything that reads the window afterwards (vil_download_state, vil_marginalize_resident,
 * the next vil_solve_resident, vil_upload) is ordered behind it on the context's stream; VIL_NO_POLL=1 restores copy + synchronise. */
int vil_upload(vil_ctx* ctx, const vil_problem* problem, const vil_state* state);
int vil_solve_resident(vil_ctx* ctx, const vil_options* options, vil_summary* summary);
int vil_reset_state(vil_ctx* ctx);                       /* restore the uploaded state on device */
int vil_download_state(vil_ctx* ctx, vil_state* state_out);

/* kernel timing with HIP events recorded on the library's own stream around every sweep launch
 * (bench.py's roofline leg).  Off by default. */
typedef struct vil_profile {
    int64_t sweep_launches;        /* live (not early-exited) sweep launches timed              */
    double sweep_ms;               /* sum of their durations                                    */
    int64_t step_launches;
    double step_ms;                /* sweep end -> next sweep start (reduce + step kernels + boundaries) */
    double reduce_ms;              /* of which: sweep end -> reduce end                         */
    double collective_ms;          /* of which: reduce end -> end of the iteration's collective (sharded solves; ~0 otherwise) */
} vil_profile;
int vil_profile_enable(vil_ctx* ctx, int on);
int vil_profile_read(vil_ctx* ctx, vil_profile* out, int reset);
/* One-launch iterations (the whole trust-region iteration as one kernel launch: sweep, gather, chain elimination, step): the launch stamps its phases with the
 * device's 100 MHz wall clock while profiling is on -- vil_profile.sweep_ms is then the sweep PHASE (first workgroup started -> last sweep role's record out),
 * reduce_ms the gather's tail behind it, step_ms the rest of the launch.  vil_profile_phases returns the average position in microseconds of 16 phase stamps
 * after the launch's first workgroup started (24 slots; csrc/vilsolve.hip lists them) and the number of launches averaged. */
int vil_profile_phases(vil_ctx* ctx, double* avg_us32, int64_t* launches, int reset);
/* the raw stamps (100 MHz device clock; 32 per launch / iteration, slot 0 stored inverted; 0: not stamped) of the last profiled solve: returns the launches copied */
int vil_debug_read_stamps(vil_ctx* ctx, uint64_t* out, int32_t max_launches);
/* the 16 wall-clock stamps (100 MHz) the kernels of the context's LAST marginalisation left: k_marg [0] entered, [1] dropped block gathered, [2] its Cholesky inverse
 * done, [3] kept x dropped blocks staged, [4] T = A_kd A_dd^-1, [5] A = A_kk - T A_dk and b, [6] symmetrised copies out; k_marg_fast [7] entered, [8] tiles loaded,
 * [9] n x n factorisation done, [10] J0 / r0 out; k_marg (pivoted square root, only when the un-pivoted one was refused) [11] entered, [12] done */
int vil_debug_marg_stamps(vil_ctx* ctx, uint64_t* out16);

/* replaces ceres::CostFunction::Evaluate for a whole factor class at once: raw (no loss) residuals
 * and row-major global-size Jacobian blocks, factor-major, in the caller's factor order. */
int vil_eval_factors(vil_ctx* ctx, const vil_problem* problem, const vil_state* state,
                     int factor_class, double* residuals, double* jacobians);

/* The four functors of lidar_mapping/src/lidarFactor.hpp in window-pose form (point in the LiDAR frame, LiDAR->body extrinsic
 * q_lb / t_lb, one window pose [p q(xyzw)]), Evaluate()-compatible: residuals n x nr, Jacobians n x nr x 7 row-major (pose tangent
 * in the first six columns).  kind / constants per point / nr:
 *   VIL_LIDAR_EDGE       [cp a b]      9   3   LidarEdgeFactor       :12-54  (s = 1)
 *   VIL_LIDAR_PLANE3     [cp j l m]   12   1   LidarPlaneFactor      :57-104 (s = 1; unit normal of (j-l) x (j-m), as the functor's ctor)
 *   VIL_LIDAR_PLANE_NORM [cp n d]      7   1   LidarPlaneNormFactor  :107-138
 *   VIL_LIDAR_DISTANCE   [cp closed]   6   3   LidarDistanceFactor   :141-172
 * The window solve uses EDGE and PLANE_NORM (problem.edge_* / plane_*); PLANE3 and DISTANCE are never instantiated by the
 * reference (localMapping.cpp:668-685, 748-765 are commented out) and are provided for completeness of the per-factor surface. */
enum { VIL_LIDAR_EDGE = 0, VIL_LIDAR_PLANE3 = 1, VIL_LIDAR_PLANE_NORM = 2, VIL_LIDAR_DISTANCE = 3 };
int vil_eval_lidar_functors(vil_ctx* ctx, int32_t kind, int32_t n, const double* consts, const double* q_lb, const double* t_lb,
                            const double* pose7, double* residuals, double* jacobians);

/* Plan of the sweep's visual role for a window (host logic, no device needed): the landmarks are sorted by (first frame, last frame), the factor tables
 * stored in that order and cut into chunks, one per workgroup; a chunk's record is the upper 16 x 16 tiles of ITS frame window.  Diagnostic surface:
 * chunk count, widest window in column tiles, bytes of all records of one sweep (and what the packed (6K + 7)^2 triangles of rounds 1 - 3 would have been),
 * dynamic LDS of the sweep; optionally per chunk {first frame, frames, factors, landmarks} and the sorted position of every factor. */
typedef struct vil_visual_plan_info {
    int32_t n_chunks, max_tiles, tiles_per_wave, cost_cap;
    int64_t record_bytes, dense_record_bytes, lds_bytes;
} vil_visual_plan_info;
int vil_visual_plan(const vil_problem* problem, vil_visual_plan_info* info, int32_t max_chunks, int32_t* chunk_first_frame, int32_t* chunk_frames,
                    int32_t* chunk_factors, int32_t* chunk_landmarks, int32_t* factor_position);

/* one linearisation of the whole window: robustified cost and the Schur-reduced normal equations
 * S (D x D row-major, D = vil_reduced_dim(K)), g (D), at `state`, with trust-region damping mu = 0
 * and no Jacobi scaling.  Diagnostic / parity surface of the hot loop's sweep + reduction. */
int vil_linearize(vil_ctx* ctx, const vil_problem* problem, const vil_state* state,
                  const vil_options* options, double* cost, double* S, double* g);

/* replaces estimator.cpp:1486-1616 / 1624-1681 (MarginalizationInfo: preMarginalize + marginalize
 * + getParameterBlocks with the i->i-1 address shift done as an index remap).
 * Collected factors: the prior, IMU (0,1), every visual factor anchored in frame 0, the remembered ICP / LPS constraint and --
 * extended mode -- the LiDAR edge / plane points attached to frame 0 (marginalization_factor.cpp:176-316 folds every factor
 * that touches a dropped block; they touch pose 0 only).  Rank rule: directions of A_mm at or below eps = 1e-8 are zeroed like
 * the reference's eigenvalue threshold (:277) -- a landmark without parallax (h_ll <= eps) keeps its factors' information on
 * the poses; the 15 x 15 pose / speed-bias block is inverted by Cholesky when certified full rank, by thresholded
 * eigen-decomposition otherwise.
 * NOTE: vil_marginalize, vil_eval_factors and vil_linearize upload a derived problem and thereby REPLACE the window left
 * resident by vil_upload / vil_solve (vil_solve_resident then returns VIL_ERR_INVALID_ARGUMENT until the next upload). */
int vil_marginalize(vil_ctx* ctx, const vil_problem* problem, const vil_state* state,
                    const vil_options* options, const vil_marg_spec* spec, vil_prior_out* out);

/* ---- window residency across frames (slideWindow, estimator.cpp:1689-1814) ------------------------------------
 * What does not change between two images stays in HBM:
 *  - LiDAR point factors live on the device as one slab per window frame.  vil_lidar_push appends the newest frame's points
 *    (the only LiDAR bytes that cross PCIe per image), vil_lidar_drop removes a frame (0 for MARGIN_OLD, count-2 for
 *    MARGIN_SECOND_NEW; later slabs move down -- an index remap, no data moves).  A problem with
 *    n_plane = n_edge = VIL_LIDAR_RESIDENT takes its point factors from the slabs: slab i belongs to window pose
 *    K - count + i (the newest slab is the newest frame); plane_* / edge_* of the problem are ignored.
 *    vil_lidar_push / drop / reset change which slab belongs to which pose (and recycle physical slabs): a window that was uploaded
 *    with resident point factors is INVALIDATED by them -- vil_solve_resident / vil_marginalize_resident return
 *    VIL_ERR_INVALID_ARGUMENT until the next vil_upload / vil_solve.  Per image: solve, marginalise, THEN drop and push.
 *  - vil_set_gauge_fix(ctx, 1): vil_solve applies double2vector()'s yaw / translation gauge fix (estimator.cpp:960-1011) on
 *    the device before the state is read back (vil_gauge_fix on the returned state is then the identity).
 *  - vil_marginalize_resident marginalises the window that vil_solve / vil_upload left on the device, at its solved (and
 *    gauge-fixed) state: the factors MarginalizationInfo collects are selected by masks inside the sweep, nothing is packed
 *    or uploaded again.  The linearisation point x0 of the new prior is read back from the DEVICE state the factors were
 *    linearised at (`solved` only supplies K and L; with the device gauge fix on, that is the state vil_solve returned).
 *    Same results as vil_marginalize on the same window and state. */
#define VIL_LIDAR_RESIDENT (-1)
int vil_lidar_reset(vil_ctx* ctx);
int vil_lidar_push(vil_ctx* ctx, int32_t n_plane, const double* plane_const, int32_t n_edge, const double* edge_const);
int vil_lidar_drop(vil_ctx* ctx, int32_t slab);
int vil_lidar_count(vil_ctx* ctx, int32_t* n_slabs, int32_t* n_plane, int32_t* n_edge);
int vil_set_gauge_fix(vil_ctx* ctx, int32_t on);
int vil_marginalize_resident(vil_ctx* ctx, const vil_state* solved, const vil_options* options,
                             const vil_marg_spec* spec, vil_prior_out* out);

/* ---- the fully resident window -----------------------------------------------------------------------------------
 * Everything slideWindow() leaves unchanged stays in HBM across images -- observations, IMU samples and pre-integration records,
 * LiDAR points, the marginalisation prior -- in SLOTS the library hands out and recycles: sliding the window is an index remap on
 * the host, nothing moves on the device.  Per image the host sends
 *    vil_win_push_frame   the new frame: the IMU samples since the previous frame (pre-integrated ON the device into the frame's IMU
 *                         slot, integration_base.h:30-158), its feature observations (by track slot), its LiDAR points     -- one DMA
 *    vil_win_solve        the window's small tables: landmark list (track slot, anchor frame, observation count, constancy, inverse
 *                         depth), the camera state (the host predicts the newest frame, estimator.cpp:170-200, and owns the depth
 *                         bookkeeping of FeatureManager, so the state is host-authoritative: 16 K + 8 + L doubles), ICP / LPS      -- one DMA
 *    vil_win_marginalize  nothing: the factors MarginalizationInfo collects are selected by masks on the resident tables, the new
 *                         prior (J0, r0, x0 and the contractions the solve uses) is written device-to-device into the prior slot
 *                         and read there by the next vil_win_solve.  The call returns when the work is enqueued; it reports the
 *                         block structure of the new prior (a function of the window's structure alone), a failure surfaces as the
 *                         status of the next vil_win_solve / vil_win_prior_download.
 *    vil_win_drop_frame   nothing: MARGIN_OLD -- frame 0 leaves; MARGIN_SECOND_NEW -- frame count-2 leaves and the newest frame's IMU
 *                         samples continue its interval (estimator.cpp:1763-1772, re-integrated on the device).
 * and receives the solved, gauge-fixed state (vil_state) in pinned memory the finishing kernel wrote (no copy, no synchronisation).
 * Track slots are the caller's: one per live feature track (FeaturePerId), < max_tracks, reusable once the track is gone.
 * Observation layout (VIL_WIN_OBS = 8 doubles): [x y z vx vy cur_td row 0], row = v - ROW/2 (projection_td_factor.cpp:12-19).
 * Under a communicator (vil_comm_init / _init_local / _ipc_init) every rank makes the SAME calls with the SAME arguments: each is handed
 * every frame and every landmark list (the observation store and the IMU slots are whole on every rank -- a few kB per image), keeps ITS slice of each
 * frame's LiDAR points and the visual factors of ITS landmark range (the rule of vil_shard_ranges applied to the landmark list), rank 0 alone the IMU /
 * prior / ICP / LPS factors; the new prior is formed by every rank from the all-reduced marginalisation system (identical bits) into its own slot.
 * One collective per trust-region iteration, one per marginalisation, as for a window handed over through vil_upload. */
#define VIL_WIN_OBS 8
typedef struct vil_win_cfg {
    int32_t K;                      /* frames in the window */
    int32_t max_tracks;             /* track slots */
    int32_t max_samples;            /* IMU samples per interval kept on the device (a merged interval counts its parts) */
    int32_t use_td;                 /* ProjectionTdFactor (1) / ProjectionFactor (0) */
    double noise[4];                /* ACC_N GYR_N ACC_W GYR_W (integration_base.h:21-27) */
    double G[3], sqrt_info_px, tr_over_row, q_lb[4], t_lb[3];      /* as in vil_problem */
} vil_win_cfg;
typedef struct vil_win_frame {
    int32_t n_samples; const double* dt; const double* acc; const double* gyr;       /* n, n x 3, n x 3: the interval that ENDS in this frame (0 for the first frame) */
    double acc0[3], gyr0[3], lin_ba[3], lin_bg[3];     /* IntegrationBase{acc_0, gyr_0, Bas, Bgs} (estimator.cpp:125-128) */
    int32_t n_obs; const int32_t* obs_track; const double* obs;                       /* n_obs, n_obs x VIL_WIN_OBS */
    int32_t n_plane; const double* plane_const; int32_t n_edge; const double* edge_const;
} vil_win_frame;
typedef struct vil_win_problem {
    int32_t L; const int32_t* lm_track; const int32_t* lm_start; const int32_t* lm_nobs;   /* landmark l: track slot, anchor frame, observations in consecutive frames (>= 2) */
    const uint8_t* lm_const;        /* L or NULL */
    const uint8_t* pose_const; const uint8_t* sb_const; int32_t ex_const, td_const;
    int32_t n_icp; const int32_t* icp_ids; const double* icp_const;
    int32_t n_lps; const int32_t* lps_ids; const double* lps_const;
} vil_win_problem;
#define VIL_WIN_MAXBLK 24
typedef struct vil_win_prior_info { int32_t n, nblk, m; int32_t blk_kind[VIL_WIN_MAXBLK], blk_index[VIL_WIN_MAXBLK], blk_col[VIL_WIN_MAXBLK]; } vil_win_prior_info;
int vil_win_open(vil_ctx* ctx, const vil_win_cfg* cfg);                      /* (re)opens an empty window; the prior slot is cleared */
int vil_win_push_frame(vil_ctx* ctx, const vil_win_frame* frame);
int vil_win_drop_frame(vil_ctx* ctx, int32_t marg_flag);
int vil_win_solve(vil_ctx* ctx, const vil_win_problem* problem, vil_state* state_inout, const vil_options* options, vil_summary* summary);
int vil_win_marginalize(vil_ctx* ctx, const vil_options* options, const vil_marg_spec* spec, vil_win_prior_info* info /* may be NULL */);
int vil_win_prior_download(vil_ctx* ctx, vil_prior_out* out);              /* on request (tests, logging): the resident prior; out->n = 0 when there is none */
int vil_win_prior_set(vil_ctx* ctx, const vil_prior* prior);               /* seed the prior slot from the host (NULL or n == 0: none) */

/* host-side helpers of the boundary */
int vil_reduced_dim(int K);                               /* 15K + 7 */
void vil_prior_capacity(int K, int* n_max, int* nblk_max, int* x0_max);
void vil_default_options(vil_options* o);
/* estimator.cpp:960-1011: yaw/translation gauge fix of double2vector(), applied to a solved state
 * given the pre-solve pose of frame 0 (Rs[0], Ps[0]). */
int vil_gauge_fix(const double* pose0_before, vil_state* solved);
/* partition of the factor set for rank r of w: visual by landmark owner, LiDAR points in
 * contiguous pose-sorted chunks; fills [begin,end) ranges.  Pure host logic. */
int vil_shard_ranges(const vil_problem* problem, int rank, int world,
                     int32_t* lm_begin, int32_t* lm_end, int32_t* edge_begin, int32_t* edge_end,
                     int32_t* plane_begin, int32_t* plane_end);

#ifdef __cplusplus
}
#endif
#endif /* VILSOLVE_H */
