// Device-visible problem description shared by all kernels of libvilsolve (passed by value).
// Reduced ("camera") ordering, identical to the oracle's:  pose k -> 6k ; ex -> 6K ; td -> 6K+6 ;
// speed-bias k -> 6K+7+9k ;  D = 15K+7 ;  NV = 6K+7 is the sub-space visual factors touch.
#pragma once
#include <stdint.h>

#define VIL_THREADS 256       // eval / reduce kernels, LiDAR chunk size
#define VIL_SWEEP_THREADS 512 // sweep workgroups: 8 waves = 2 per SIMD for LDS-latency hiding
#define VIL_VCHUNK_FBAL 32    // factors below which a visual chunk is not closed for balance (vilsolve.hip: visual_chunks; its bounds: VIS_MF, VIS_LM, VIS_GM in vil_sweep.hpp)
#define VIL_STEP_THREADS 512

struct SysBuf {       // one linearisation of the window (double-buffered: current / candidate)
    double* S;        // D x D  Schur-reduced H (upper triangle accumulated by the sweep, mirrored by the step kernel)
    double* gred;     // D      Schur-reduced gradient
    double* bc;       // D      un-reduced gradient J_c^T r
    double* diag;     // D      diagonal of the un-reduced H_cc
    double* hll;      // L      J_l^T J_l
    double* bl;       // L      J_l^T r
    double* invp;     // L      1 / (hll + mu dl^2 / Sl^2), 0 for constant landmarks
    double* sl;       // L      Jacobi scale of the landmark (fixed at the first linearisation; carried with the set so that it travels with the all-reduce)
    double* eA;       // L x 13 e_l on [anchor pose(6) | ex(6) | td(1)]
    double* eO;       // F x 6  e_l on the observing pose of each factor
    double* cost;     // 1
    double* ar;       // start of the set: ONE contiguous block [S | gred | bc | diag | cost | 2 spare | hll | bl | invp | sl | eA | eO] --
                      // multi-GPU: every rank fills the landmark arrays of the landmarks it owns (zeros elsewhere) and the whole block is all-reduced once
};

struct Ctl {          // trust-region state, lives in device memory, owned by the step kernel
    int gen;          // solve generation (host): with n_sweeps it forms the epoch of the helper-workgroup flags
    int cur, iter, done, term, first, resweep, reuse, nsucc, invalid_run, status, lin_mode, n_sweeps;
    int swe;                      // advanced by every live step-kernel launch: epoch of the sweep's workgroup flags (swflag)
    int outd, pad_;               // outd: the finished solve has been written out (solve_finish: accepted state -> x[0], gauge fix, host mirror)
    double radius, mu, cost_cur, model_change, alpha, dogleg_norm, initial_cost, cand_cost;
    double mu_used, gn2, g2, gg;   // dogleg scalars of the current linearisation (reused after a rejected step)
    double cg, cn;                 // dogleg coefficients of the candidate: the sweep forms lambda_cand = lambda_cur + cg la + cn lb
    double saa, sab, sbb, xnl;     // landmark sums |la|^2, la.lb, |lb|^2, |lambda|^2 of the current linearisation (step norm without a landmark pass)
    double cost_trace[64], radius_trace[64];
};

struct SolveOpts {
    int max_iterations, jacobi_scaling, visual_loss, lidar_loss, rel_loss, autodiff_quirk;
    int precision;      // 1: visual / LiDAR point factors evaluated in fp32 (accumulation stays fp64)
    double function_tolerance, gradient_tolerance, parameter_tolerance, max_radius, min_relative_decrease, min_mu, max_mu;
    double visual_loss_scale, lidar_loss_scale, rel_loss_scale;
};

// The 405 of an IMU record's 931 doubles [30x30 H | 30 g | cost] that the chain workgroup gathers (speed-bias blocks of both frames, lower triangles; the block between them;
// the pose rows against the speed-bias columns; the speed-bias gradient), as one compact record: the chain workgroup is ONE compute unit pulling these across the
// device, bound by the lines it has in flight -- 405 contiguous doubles instead of 9-double runs in 240-byte rows.  -1: not gathered.  Host (table) and device (IMU role).
#define VIL_CHAIN_REC 405
__host__ __device__ inline int chain_rec_index(const int e) {
    if (e >= 930) return -1;
    if (e >= 900) { const int c = e - 900; return (c >= 6 && c < 15) ? 387 + c - 6 : c >= 21 ? 396 + c - 21 : -1; }
    const int r = e / 30, c = e - 30 * r;
    const bool ci = c >= 6 && c < 15, cj = c >= 21;
    if (!ci && !cj) return -1;
    if (r >= 6 && r < 15) return (ci && c <= r) ? (r - 6) * (r - 5) / 2 + c - 6 : -1;
    if (r >= 21) return ci ? 90 + (r - 21) * 9 + c - 6 : c <= r ? 45 + (r - 21) * (r - 20) / 2 + c - 21 : -1;
    const int pr = r < 6 ? r : r - 9;
    return ci ? 171 + pr * 9 + c - 6 : 279 + pr * 9 + c - 21;
}
struct DevP {
    int K, L, D, NV, NS;
    // constancy
    const uint8_t* pose_const; const uint8_t* sb_const; const uint8_t* lm_const;
    int ex_const, td_free, use_td;
    // state double buffer: [pose 7K | sb 9K | ex 7 | td 1 | lam L]
    double* x[2];
    // visual (SoA: component k of the factor at sorted position p at vis_c[k*vis_stride + p]; sorted = by landmark (first frame, last frame): vfinv / vfac)
    int n_vis, vis_stride;
    const double* vis_c; const int* vis_i; const int* vis_j; const int* vis_l;
    const int* lm_start;      // L+1
    const int* lm_acol;       // L   reduced column of the anchor pose (6 * start_frame), -1 if the landmark has no factor
    const int* fcol;          // F   reduced column of the observing pose of each factor (6 * vis_j)
    // the same three tables over the WHOLE window and the offset of this rank's first visual factor in it: with the factor set
    // sharded over ranks the sweep writes e_O at global factor positions and the step kernel walks every landmark (the local
    // tables only know this rank's factors).  Un-sharded: aliases of the local tables, vis_f0 = 0.
    const int* glm_start; const int* glm_acol; const int* gfcol; int vis_f0;
    // LiDAR points, sorted by pose; chunk = (start, count, pose)
    int n_plane, pl_stride, n_pchunk; const double* pl_c; const int* pchunk;
    int n_edge, ed_stride, n_echunk; const double* ed_c; const int* echunk;
    int lidar_rep;                // passes of two 256-point chunks a LiDAR workgroup of the sweep makes (vil_sweep.hpp: sweep_body)
    double Rbl[9], tbl[3];
    // IMU
    int n_imu; const double* imu_c; const double* imu_U; const int* imu_i; const int* imu_j;
    // prior
    int pn, pnblk; const int* pblk_kind; const int* pblk_index; const int* pblk_col; const int* pblk_xoff; const int* pmap;
    const double* px0; const double* pJ0; const double* pr0; double* pH; double* pg0; double* pc0;
    // ICP / LPS
    int n_icp, n_lps; const int* icp_ids; const double* icp_c; const int* lps_ids; const double* lps_c;
    double G[3], sqrt_info, k_tr;
    // linear system + solver work space
    SysBuf sys[2];
    // per-workgroup partial results of the sweep (no global atomics); gathered by k_reduce
    // visual workgroups = chunks of the landmark list sorted by (first frame, last frame) (vil_sweep.hpp: sweep_visual; vilsolve.hip: visual_chunks)
    int n_vwg, vis_ts;            // chunks; accumulator tiles per wave the widest chunk needs (k_sweep<vis_ts>: 2 or 5)
    const int* vwg;               // n_vwg x 8: {first sorted landmark, landmarks, first sorted factor, factors} {first frame, frames, column tiles T, record offset / 16}
    const int* vlm;               // sorted landmarks x 4: {landmark, chunk-local first factor, factors, anchor frame}
    const int* vfac;              // sorted factors x 2: {factor (the caller's index), chunk-local landmark}; vis_c / vis_i / vis_j / vis_l are stored in SORTED order
    const int* vfinv;             // caller's factor index -> sorted position (vil_eval_factors, k_win_pack)
    const int* vrec;              // n_vwg x 4: {record offset / 16, first frame, frames, T} (what the gather needs of vwg)
    const int* vwend;             // K: chunks whose first frame is <= f (the chunks are sorted by first frame)
    double* vpart;                // the records: per chunk T (T + 1) / 2 upper 16 x 16 tiles of its window | bc 16 T | diag 16 T | cost (+ padding to 16)
    double* lpart;                // (n_pchunk + n_echunk) x 28  [21 upper 6x6 | 6 g | cost]
    const int* lchunk_pose;       // 2 x (K+1): chunk ranges per pose (plane, edge)
    double* ipart;                // n_imu x 931  [30x30 H | 30 g | cost]
    double* chc;                  // the prior's constant share (J0^T J0 entries) of the chain workgroup's gather, in table order: written by the first iteration of a solve, read by the later ones
    const int* imu_perm;          // the order an IMU role of a one-launch iteration forms its 931 record entries in: the VIL_CHAIN_REC the chain workgroup gathers first (in the compact record's order)
    int wg_launch;                // profiling: the launch (0-based, of the solve) whose workgroups leave their entry / exit times behind the phase stamps in P.prof (vil_profile_workgroups); -1: none
    int* sall;                    // one-launch iteration: [16] / [32] = launch epoch once every non-visual / visual sweep role has posted -- published by the master workgroup, which polls the roles' flags ONCE for all gather workgroups
    int* cflag;                   // n_imu: an IMU role's compact record is complete (launch epoch; P.sflag[role] follows when the whole record is)
    double* irec;                 // n_imu x VIL_CHAIN_REC: one-launch iteration -- the part of the IMU records the chain workgroup gathers, compact (chain_rec_index)
    double* mpart;                // prior: [pn g | cost] then n_rel x 601 [24x24 H | 24 g | cost]
    const int* pinv;              // D: reduced column -> prior column or -1
    double* Sl; double* Sc; double* dc; double* dl; double* gradc; double* gradl; double* gnc; double* gnl;
    double* M; double* stepc; double* stepl; double* tmpc; double* tmpl;
    Ctl* ctl;
    int rank, world;              // data-parallel shard of the factor set
    long long* dbg;               // 64 cycle stamps (debug/profiling aid)
    // marginalisation of the RESIDENT window (vil_marginalize_resident): 1 = MARGIN_OLD -- only the factors touching frame 0 are live
    // (IMU (0,1), landmarks anchored in frame 0, LiDAR points of pose 0, the chosen ICP / LPS constraint, the prior), every block
    // free; 2 = MARGIN_SECOND_NEW -- only the prior.  0 = a solve.
    int marg, marg_icp, marg_lps;
    int skip_mask;                // debug: bit0 visual, 1 imu, 2 plane, 3 edge, 4 misc roles skipped in the sweep
    // helper workgroups of the single-GPU step kernel (landmark pre-pass on extra CUs): n_help of them, each publishes
    // {q, g2, gm} in hpart[4 * k ..] and then stores the launch epoch in hflag[k]; only the master workgroup ever waits
    // pinned host mirror of Ctl (device pointers into mapped host memory; null: off): the step kernel that finishes the solve copies
    // its Ctl there and then stores the solve generation in hseq -- the host polls that word instead of synchronising the stream
    Ctl* hctl; int* hseq;
    double* hstate;               // same mirror: the NS doubles of the final state (solve_finish)
    const double* xorig;          // the state the solve started from (the gauge fix re-anchors on its frame 0)
    int gauge_on;                 // double2vector()'s yaw / translation gauge fix as part of solve_finish
    const int* setup_stat;        // != 0: k_setup found an IMU covariance that is not positive definite -- the first step kernel ends the solve with it
    int n_help; double* hpart; int* hflag;
    // second landmark pass of the helpers (k_step): the master leaves Sc x_p in stepc as 64-bit words {half of a value, launch epoch} -- or
    // the epoch in xstat (no step this launch); every helper then leaves its six sums the same way in hpart2[16 * slot ..] (xflag / hflag2: unused)
    double* hpart2; int* hflag2; int* xflag; int* xstat;
    double* la; double* lb;        // L each: step directions of the inverse depths (Cauchy, Gauss-Newton), written by the step kernel's landmark pass
    // structure-exploiting solve (vil_chain.hpp): 0 dense, 1 chain with W^T in LDS, 2 chain with W^T in global memory (P.M)
    int chain, chain_rs;
    // ---- speed-bias chain eliminated BESIDE the gather (vil_prechain.hpp): prechain = 1 (single GPU, every IMU factor couples
    // frames (k, k+1), chain step kernel).  One extra workgroup of the merged gather + step launch gathers
    // the chain part of S' from the IMU / prior records through a host-built table (chtab: n_chtab x {dst, src a, src b, src c}), eliminates the
    // chain and leaves: W^T with unscaled pose rows (chW), the factored blocks (chLdg, chLsb), the chain columns' scales (chSc, chDc),
    // pieces of u^T S' u (chZ, chQ), status (chOk).  Further workgroups then form W W^T tile by tile (chWW, tiled lower layout).
    int prechain, n_chtab; const int* chtab; const int* chpq;      // prechain 1: beside the gather in the merged launch; 2: in k_sweep behind the IMU / prior flags (swflag), tiles in k_reduce
    int* swflag;                   // n_imu + 1 flags: the IMU / prior workgroups of the current sweep have written their records (prechain 2)
    // gather + step in ONE launch (rs_merged): grid = [master | helpers | chain | W W^T tiles (n_ww) | gather (n_gather)]; a workgroup that is
    // done posts the launch epoch in its flag -- gflag[n_gather], chflag, wwflag[n_ww] -- and the master / helpers / tile workgroups wait on them
    int rs_merged, n_ww, n_gather; int* gflag; int* chflag; int* wwflag;
    // ---- the whole iteration in ONE launch (k_iter, vil_iter.hpp): grid = [sweep roles (n_sw workgroups, the order of k_sweep) | chain | gather | master | helpers | W W^T tiles];
    // every sweep workgroup posts the launch epoch in sflag[its index] once its record is out (agent-scope stores); the gather workgroups wait for all of them,
    // the chain workgroup for the IMU / prior ones
    int n_sw; int* sflag;
    long long vmirror;             // doubles from a visual record to its mirror (timing experiment VIL_SKIP=1024: every tile written twice, gathered twice)
    int persist;                   // 1: this launch is the persistent solve (k_solve): Ctl leaves through its tail, the helpers post hflag2 behind their la / lb stores
    unsigned long long* xtag;      // persistent solve: the candidate's camera part as 2 (16 K + 8) tagged words {half of a value, epoch}, written by the master, polled by the sweep roles
    unsigned long long* ihdr;      // persistent solve (k_solve): the 64-byte hand-over line between two iterations (vil_iter.hpp)
    int drop_role, drop_launch;    // test hook (vil_debug_drop_flag): sweep role `drop_role` (its workgroup index in k_sweep's order; -2 - g: gather workgroup g) does not post its flag in launch `drop_launch` (0-based) of the solve; -1: off
    int* abortf;                   // one-launch iteration: a wait on another workgroup's flag that lasts 50 ms gives up and says so here; every later wait returns at once (vil_math.hpp: spin_until_eq)
    long long* prof;               // != null: wall-clock stamps (s_memrealtime, 100 MHz) of the roles of a one-launch iteration, 8 per launch slot (vil_profile)
    int gather_pose_only;          // the gather forms S' on the visual sub-space + the diagonal only (a solve on the prechain path: vil_sweep.hpp, reduce_gather)
    double* chW; double* chLraw; double* chLdg; double* chLsb; double* chSc; double* chDc; double* chZ; double* chQ; int* chOk; double* chWW;
};

// Phase stamps of a one-launch iteration (vil_profile_phases): slot k of launch `launch` keeps the LATEST (or, want_min, the EARLIEST: stored inverted) wall-clock
// reading any workgroup posted for it -- s_memrealtime, the 100 MHz counter every XCD shares; one atomic per workgroup and slot, nothing when P.prof is null
#define VIL_PROF_SLOTS 32
#define VIL_PROF_WGS 4096      // workgroups of one launch whose entry / exit times fit behind the phase stamps (vil_profile_workgroups)
#if defined(__HIPCC__)
__device__ __forceinline__ void prof_stamp(const DevP& P, int launch, int k, bool want_min = false) {
    if (P.prof) { const unsigned long long t = wall_clock64(); atomicMax((unsigned long long*)P.prof + VIL_PROF_SLOTS * (launch & 63) + k, want_min ? ~t : t); }
}
#endif
__host__ __device__ inline int xo_pose(const DevP& P, int k) { return 7 * k; }
__host__ __device__ inline int xo_sb(const DevP& P, int k) { return 7 * P.K + 9 * k; }
__host__ __device__ inline int xo_ex(const DevP& P) { return 16 * P.K; }
__host__ __device__ inline int xo_td(const DevP& P) { return 16 * P.K + 7; }
__host__ __device__ inline int xo_lam(const DevP& P) { return 16 * P.K + 8; }
__host__ __device__ inline int col_pose(const DevP& P, int k) { return 6 * k; }
__host__ __device__ inline int col_ex(const DevP& P) { return 6 * P.K; }
__host__ __device__ inline int col_td(const DevP& P) { return 6 * P.K + 6; }
__host__ __device__ inline int col_sb(const DevP& P, int k) { return 6 * P.K + 7 + 9 * k; }
