// The factor sweep: ONE launch evaluates every residual block of the window at the candidate state
// and contracts it towards the Schur-complement normal equations of the candidate linearisation
// (what ceres' Evaluate + SchurEliminator do per trust-region iteration behind estimator.cpp:1414).
// No global atomics: every workgroup accumulates in registers / LDS and writes ONE partial record;
// k_reduce then gathers the partials into the dense reduced system S', g (deterministic order).
//
// Work-group roles by blockIdx (workgroups of VIL_SWEEP_THREADS = 512 threads):
//   [imu]     one WG per IMU factor: lane 0 forms the raw 15x30 block, the WG whitens with the
//             pre-factored sqrt-information and contracts to a 30x30 H block
//   [visual]  one WG per chunk of (frame-sorted) landmarks, <= VIS_LM landmarks / VIS_MF factors: thread-per-factor evaluation
//             staged in LDS, 16 lanes per landmark for the Schur pivots, then sum_f Jc^T Jc - sum_l invp e e^T on the fp64 matrix
//             cores in the chunk's frame-window-local columns; the record is the upper 16 x 16 tiles of that window
//   [plane]/[edge] 256 threads per <=256 pose-uniform LiDAR points (two chunks per WG): thread-per-point evaluation,
//             wave64 butterfly reduction of the 6x6 + 6 + cost
//   [prior]   n x n gemv on the pre-contracted J0^T J0
//   [rel]     ICP and LPS AutoDiff factors (scalar forward-mode duals, thread = (factor, block, coordinate))
#pragma once
#ifdef VIL_PERSIST_TU
#define VIL_X_IN_LDS true       // this translation unit's one-launch roles are k_solve's: the camera part of the state comes as the workgroup's LDS copy
#else
#define VIL_X_IN_LDS false
#endif
#include "vil_dev.hpp"
#include "vil_factors.hpp"
#include "vil_finish.hpp"
#include "vil_prechain.hpp"

namespace vd {

__device__ __forceinline__ double block_sum(double v, double* red /*>= 4 doubles*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((vil_tid() & 63) == 0) red[vil_tid() >> 6] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sweep_imu(const DevP& P, const SolveOpts& O, int f, const double* x, double* sm, const int plaunch = -1, const bool chain_rec = false /* one-launch iteration: what the chain workgroup gathers also leaves as a compact record (chain_rec_index, vil_dev.hpp) */, const int chain_epoch = 0,
                                          const bool x_lds = false /* persistent solve: x is the workgroup's copy in LDS */, const bool resident = false /* ... and the factor's constants, sqrt-information and constancy flags are in LDS since the solve's first iteration */) {
    double* const outc = chain_rec ? P.irec + (size_t)f * VIL_CHAIN_REC : nullptr;
    #define IPROF(k) do { if (plaunch >= 0 && f == 0 && vil_tid() == 0) prof_stamp(P, plaunch, k); } while (0)
    IPROF(16);
    double* Jraw = sm;            // 450
    double* rr = sm + 450;        // 15
    double* UJ = sm + 480;        // 450
    double* Ur = sm + 930;        // 15
    double* cs = sm + 960;        // 287: the factor's constants (pre-integrated deltas, their bias Jacobians, linearisation biases, sum_dt)
    double* Us = sm + 1248;       // 225: sqrt-information (upper triangular)
    double* xs = sm + 1480;       // 32: pose i | speed-bias i | pose j | speed-bias j
    const double* c = P.imu_c + (size_t)f * 287;
    double* out = P.ipart + (size_t)f * 931;
    const int t = vil_tid();
    const int i = P.imu_i[f], j = P.imu_j[f];
    // (one-launch iteration: the order in which the record's entries are formed, P.imu_perm, is asked for HERE -- behind the whitening's barrier it was a dependent
    //  round trip to memory on the path the chain workgroup waits for)
    const bool perm_order = chain_rec && blockDim.x >= 512;
    int eA = 0, eB = 0;
    if (perm_order) { eA = P.imu_perm[t]; eB = P.imu_perm[min(t + 512, 930)]; }
    // everything the role reads from memory in ONE round trip, by different threads: constants, sqrt-information and the four state blocks go to LDS (before: sum_dt, then
    // the constants of every lane, then U inside the whitening loop -- three dependent round trips on the path the chain workgroup waits for)
    {
        double v = 0.0;
        if (!resident) { if (t < 287) v = c[t]; else if (t < 512) v = P.imu_U[(size_t)f * 225 + (t - 287)]; }
        double xv = 0.0;
        if (t < 32) { const int q = t < 7 ? xo_pose(P, i) + t : (t < 16 ? xo_sb(P, i) + (t - 7) : (t < 23 ? xo_pose(P, j) + (t - 16) : xo_sb(P, j) + (t - 23))); xv = (chain_rec && !x_lds) ? ld_ag(x + q) : x[q]; }      // (one-launch iteration: the candidate crosses from the master workgroup at agent scope)
        if (t >= 32 && t < 36 && !P.marg && !resident) {      // constancy of the four blocks (pose i, speed-bias i, pose j, speed-bias j): in the same round trip, not in front of the whitening
            const uint8_t* cp = (t & 1) ? P.sb_const : P.pose_const;
            xv = (cp && cp[t < 34 ? i : j]) ? 1.0 : 0.0;
        }
        for (int e = t; e < 450; e += blockDim.x) Jraw[e] = 0.0;
        if (!resident) { if (t < 287) cs[t] = v; else if (t < 512) Us[t - 287] = v; }
        if (t < (resident ? 32 : 36)) xs[t] = xv;      // (xs[32 .. 35]: constancy flags)
    }
    __syncthreads();
    IPROF(17);
    // (the record is stored at agent scope: the chain workgroup of the same launch may read it -- sweep_signal, prechain 2 / the one-launch iteration)
    if (cs[16] > 10.0 || (P.marg && !(P.marg == 1 && i == 0 && j == 1 && cs[16] < 10.0))) { for (int e = t; e < 931; e += blockDim.x) { st_ag(out + e, 0.0); if (outc) { const int ce = chain_rec_index(e); if (ce >= 0) st_ag(outc + ce, 0.0); } } return; }   // estimator.cpp:1182 / :1535
#ifdef VIL_STAMPS
    #define ISTAMP(k) do { if (t == 0 && f == 0) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); P.dbg[k] = tt_; } } while (0)
#else
    #define ISTAMP(k) do {} while (0)
#endif
    ISTAMP(17);
    if ((t & 63) < 3) {   // the 17 3 x 3 blocks + the residual, dealt to three lanes of each of the eight waves (common terms recomputed per lane): a wave runs the divergent
        // block cases of ITS lanes one after the other -- eighteen lanes of one wave walked all seventeen cases (2.5 us); two waves share a SIMD: ~5 cases per SIMD now
        const int item = (t >> 6) + 8 * (t & 63);      // 0 .. 23
        if (item <= IMU_NBLOCKS) {
            ImuCommon o;
            imu_common(cs, V3{P.G[0], P.G[1], P.G[2]}, xs, xs + 7, xs + 16, xs + 23, o);
            if (item == IMU_NBLOCKS) imu_resid(o, rr);
            else {
                int r0, c0; M3 m; double sc;
                imu_block(o, cs, item, r0, c0, m, sc);
                put33(Jraw, 30, r0, c0, m, sc);
                if (item == 16) put33(Jraw, 30, 12, 27, m, sc);
            }
        }
    }
    __syncthreads();
    ISTAMP(18);
    IPROF(18);
    const double* U = Us;
    const bool ci = xs[32] != 0.0, si = xs[33] != 0.0, cj = xs[34] != 0.0, sj = xs[35] != 0.0;      // (marginalisation: every block free -- the flags are zero)
    // (U is upper triangular WITH its zeros stored: a fixed fifteen-term sum -- every LDS read in flight at once -- instead of a loop from the row's diagonal; the same bits)
    for (int e = t; e < 465; e += blockDim.x) {
        if (e < 450) {
            const int row = e / 30, col = e % 30;
            const bool cst = col < 6 ? ci : (col < 15 ? si : (col < 21 ? cj : sj));
            double s = 0;
#pragma unroll
            for (int k = 0; k < 15; ++k) { const double u = U[row * 15 + k], jv = Jraw[k * 30 + col]; s += k >= row ? u * jv : 0.0; }
            UJ[e] = cst ? 0.0 : s;
        } else {
            const int row = e - 450;
            double s = 0;
#pragma unroll
            for (int k = 0; k < 15; ++k) { const double u = U[row * 15 + k], rv = rr[k]; s += k >= row ? u * rv : 0.0; }
            Ur[row] = s;
        }
    }
    __syncthreads();
    ISTAMP(19);
    IPROF(19);
    auto entry = [&](const int e) {
        double s = 0;
        if (e < 900) { const int a = e / 30, b = e % 30; for (int k = 0; k < 15; ++k) s += UJ[k * 30 + a] * UJ[k * 30 + b]; }
        else if (e < 930) { const int a = e - 900; for (int k = 0; k < 15; ++k) s += UJ[k * 30 + a] * Ur[k]; }
        else { for (int k = 0; k < 15; ++k) s += Ur[k] * Ur[k]; s *= 0.5; }
        return s;
    };
    if (perm_order) {
        // one-launch iteration: the 405 entries the chain workgroup gathers are formed and stored FIRST (P.imu_perm: they lead the order), and the role's chain flag
        // goes up behind THEIR stores -- the other 526 entries are formed while those stores travel and leave afterwards (the gather workgroups wait for the
        // role's ordinary flag, and have slack: DESIGN.md 0d).  The chain leg of the iteration starts ~0.8 us earlier.
        const double sA = entry(eA);
        st_ag(out + eA, sA);
        if (t < VIL_CHAIN_REC) st_ag(outc + t, sA);
        const double sB = entry(eB);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) { st_ag(P.cflag + f, chain_epoch); IPROF(20); }
        if (t + 512 < 931) st_ag(out + eB, sB);
        ISTAMP(28);
        return;
    }
    for (int e = t; e < 931; e += blockDim.x) {
        const double s = entry(e);
        st_ag(out + e, s);
        if (outc) { const int ce = chain_rec_index(e); if (ce >= 0) st_ag(outc + ce, s); }
    }
    ISTAMP(28);
    IPROF(20);
}

// ---------------------------------------------------------------------------------------------
// [visual] round 4: ONE chunk of landmarks per workgroup, block outer products on the fp64 matrix cores for every window size, columns LOCAL to the
// chunk's frame window, no LDS atomics anywhere (every sum has a fixed order: bit-reproducible).
//
// The host sorts the landmarks by (first frame, last frame) and cuts the sorted list into chunks (vilsolve.hip: visual_chunks): a chunk's factors
// touch the poses of frames [fa, fa + span) only, so its contribution to S' lives in a (6 span + 7)^2 corner structure -- local columns
//   pose of frame k -> 6 (k - fa) .. +5 | extrinsic -> 6 span .. +5 | td -> 6 span + 6 | r -> 6 span + 7     (T = ceil((6 span + 8) / 16) column tiles)
// and the record a workgroup leaves is T (T + 1) / 2 upper 16 x 16 tiles of THAT matrix instead of the packed triangle of the whole (6K + 7)^2 visual
// sub-space (K = 20: 65 kB per workgroup whatever it touched; 12.5 MB per sweep written and read back by the gather).  With G the dense rows of the
// corrected Jacobians (two per factor, the residual appended as column r) and E the rows e_l of the landmarks (b_l appended), the record is
//   sum_f G_f^T G_f - sum_l invp_l E_l^T E_l
// one v_mfma_f64_16x16x4 per tile and four rows, accumulators in registers from the first row to the store (every tile is stored TRANSPOSED: the
// matrix is symmetric, and the gather reads a column's rows with consecutive lanes).  Column r of that matrix is the
// Schur-reduced gradient; the un-reduced gradient and diagonal are read off the accumulators between the factor rows and the landmark rows.
// LDS per factor: [Ji 12 | Jj 12 | Jex 12 | Jt 2 | Jl 2 | r 2 | eO 6] = 48 doubles
#define VF_STRIDE 49   // odd stride: conflict-free column access
#define VIS_MF 64      // factors a chunk may hold (a thread and two operand rows each)
#define VIS_LM 16      // landmarks a chunk may hold (one operand row each: one MFMA batch)
#define VIS_GM 16384   // doubles of LDS the operand rows of a chunk may take (factor rows in batches of 16 + 16 landmark rows + 16 scales)
#define VIS_TMAX 8     // column tiles of the widest window (K = 20: 6 * 20 + 8 = 128 columns)
__host__ __device__ inline int vis_tiles(int span) { return (6 * span + 8 + 15) >> 4; }
__host__ __device__ inline int vis_rs(int T) { return 16 * ((T + 1) | 1); }          // row stride: >= 16 T + 16 and = 16 mod 32 (the four rows of an operand fragment on different banks)
__host__ __device__ inline int vis_ntile(int T) { return (T * (T + 1)) >> 1; }
__host__ __device__ inline int vis_rows(int nf) { return (2 * nf + 15) & ~15; }      // factor rows, whole batches
__host__ __device__ inline int vis_gm_doubles(int nf, int T) { return (vis_rows(nf) + 16) * vis_rs(T) + 16; }
__host__ __device__ inline int vis_rec_doubles(int T) { return vis_ntile(T) * 256 + 32 * T + 16; }      // [tiles | bc 16 T | diag 16 T | cost + padding]: whole 128-byte lines
__host__ __device__ inline int vis_slots(int T) { return (vis_ntile(T) + 7) >> 3; }  // tiles per wave (8 waves)
// fixed part of the role's LDS (doubles): red 8 | lmr | Jf | int tables (fj, fl, fa: VIS_MF each; lms VIS_LM + 1; lanc, lid: VIS_LM each)
#define VIS_LDS_FIXED (8 + VIS_LM * 16 + VIS_MF * VF_STRIDE + (3 * VIS_MF + 3 * VIS_LM + 2 + 1) / 2 + 2)
// The candidate inverse depth is formed HERE: lambda_cand = lambda_cur + cg la + cn lb (la, lb: the step directions the step
// kernel's landmark pass left, cg / cn: the dogleg coefficients in Ctl; first sweep and re-sweeps: cg = cn = 0), and written
// into the candidate state by the landmark's lane group.
// AG: what the workgroup leaves for other workgroups of the SAME launch (the one-launch iteration, vil_iter.hpp: record, landmark arrays) is stored at agent scope
template <int TS, bool AG = false>      // accumulator tiles per wave: 2 (windows whose widest chunk has T <= 5 column tiles: K <= 12) or 5 (T <= 8: K <= 20); a kernel per value
__device__ __forceinline__ void sweep_visual(const DevP& P, const SolveOpts& O, const Ctl& ctl, int wg, const double* x, SysBuf& sb, double* sm) {
    const int t = vil_tid();
    const int4 d0 = ((const int4*)P.vwg)[2 * wg], d1 = ((const int4*)P.vwg)[2 * wg + 1];      // {first sorted landmark, landmarks, first sorted factor, factors}, {fa, span, T, record offset / 16}
    const int p0 = d0.x, nl = d0.y, fp0 = d0.z, nf = d0.w, fa0 = d1.x, span = d1.y, T = d1.z;
    const int RS = vis_rs(T), ntile = vis_ntile(T), nrow = vis_rows(nf);
    const int cX = 6 * span, cT = cX + 6, cR = cX + 7;                 // local columns of the extrinsic, td and the residual
    double* red = sm;                                  // 8
    double* lmr = red + 8;                             // VIS_LM x 16: invp, eA[13], -, -
    double* Jf = lmr + VIS_LM * 16;                    // VIS_MF x VF_STRIDE
    int* fj = (int*)(Jf + VIS_MF * VF_STRIDE);         // observer frame of a factor
    int* fl = fj + VIS_MF;                             // factor -> chunk-local landmark
    int* fa = fl + VIS_MF;                             // anchor frame of a factor
    int* lms = fa + VIS_MF;                            // VIS_LM + 1 chunk-local factor offsets
    int* lanc = lms + VIS_LM + 1;                      // anchor frame of a landmark
    int* lid = lanc + VIS_LM;                          // landmark id
    double* Gm_ = sm + VIS_LDS_FIXED; double* const Gm = Gm_;          // nrow x RS rows of [Jc | r] (two per factor) | 16 x RS rows of [e_l | b_l] | 16 scales -invp_l
    double* const Em = Gm + nrow * RS;
    double* const sa = Em + 16 * RS;
#ifdef VIL_STAMPS
    #define VSTAMP(k) do { __syncthreads(); if (t == 0 && wg == 0) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); P.dbg[32 + k] = tt_; } } while (0)
    long long vt0 = 0; if (t == 0) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(vt0) :: "memory");
#else
    #define VSTAMP(k) do {} while (0)
#endif
    VSTAMP(0);
    const double* xcur = P.x[ctl.cur];
    double* xcand = P.x[1 - ctl.cur];
    const double cg = ctl.cg, cn = ctl.cn;
    // first sweep of a solve and re-sweeps: cg = cn = 0 and la / lb still hold the PREVIOUS solve's directions -- possibly inf / NaN after a
    // diverged solve, and 0 * inf is NaN: the terms are dropped, not multiplied by zero (same bits whenever la, lb are finite)
    const bool stepped = cg != 0.0 || cn != 0.0;
    for (int e = t; e < (nrow + 16) * RS + 16; e += blockDim.x) Gm[e] = 0.0;      // operand rows and scales
    const bool mfree = P.marg != 0;            // marginalisation of the resident window: every block free, factors masked
    const bool exc = !mfree && P.ex_const != 0, tdc = mfree ? !P.use_td : !P.td_free;
    double cost = 0.0;
    if (t >= 256 && t <= 256 + nl) {
        const int q = t - 256;
        if (q < nl) { const int4 lm = ((const int4*)P.vlm)[p0 + q]; lid[q] = lm.x; lms[q] = lm.y; lanc[q] = lm.w; }      // {landmark, chunk-local first factor, -, anchor frame}
        else lms[q] = nf;
    }
    if (t < nf) {
        const int2 fq = ((const int2*)P.vfac)[fp0 + t];                // {factor (caller's index: e_O is stored by it), chunk-local landmark}
        const int f = fq.x, fs = fp0 + t;                               // (the factor tables are stored in sorted order: consecutive rows for a chunk)
        double c[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) c[k] = P.vis_c[(size_t)k * P.vis_stride + fs];
        const int i = P.vis_i[fs], j = P.vis_j[fs], l = P.vis_l[fs];
        // (AG -- the one-launch iteration and the persistent solve: the camera part of the candidate was written by the master workgroup, la / lb by the helpers, the
        //  current inverse depths by a visual workgroup of an earlier iteration: all cross at agent scope; the other launch structures read them behind a launch boundary)
        double pi[7], pj[7], ex[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) { pi[k] = ldx<AG && !VIL_X_IN_LDS>(x + xo_pose(P, i) + k); pj[k] = ldx<AG && !VIL_X_IN_LDS>(x + xo_pose(P, j) + k); ex[k] = ldx<AG && !VIL_X_IN_LDS>(x + xo_ex(P) + k); }      // (VIL_X_IN_LDS: the persistent solve hands its roles the workgroup's copy of the camera part in LDS)
        const double tdv = ldx<AG && !VIL_X_IN_LDS>(x + xo_td(P));
        VisJ o;
        const double lam = stepped ? ldx<AG>(xcur + xo_lam(P) + l) + cg * ldx<AG>(P.la + l) + cn * ldx<AG>(P.lb + l) : ldx<AG>(xcur + xo_lam(P) + l);
        if (O.precision)
            visual_eval_f32(c, quatR(pi + 3), V3{pi[0], pi[1], pi[2]}, quatR(pj + 3), V3{pj[0], pj[1], pj[2]}, quatR(ex + 3), V3{ex[0], ex[1], ex[2]},
                            lam, tdv, P.sqrt_info, P.k_tr, P.use_td, o);
        else
            visual_eval(c, quatR(pi + 3), V3{pi[0], pi[1], pi[2]}, quatR(pj + 3), V3{pj[0], pj[1], pj[2]}, quatR(ex + 3), V3{ex[0], ex[1], ex[2]},
                        lam, tdv, P.sqrt_info, P.k_tr, P.use_td, o);
        double rho, rho1;
        loss_eval(O.visual_loss, O.visual_loss_scale, o.r[0] * o.r[0] + o.r[1] * o.r[1], rho, rho1);
        const bool live = !mfree || (P.marg == 1 && i == 0);      // estimator.cpp:1547-1589: landmarks anchored in frame 0
        if (!live) { rho = 0.0; rho1 = 0.0; }
        cost += 0.5 * rho;
        const double sr = sqrt(rho1);
        const bool ci = !mfree && P.pose_const && P.pose_const[i], cj = !mfree && P.pose_const && P.pose_const[j], cl = !mfree && P.lm_const && P.lm_const[l];
        double* w = Jf + t * VF_STRIDE;
        for (int k = 0; k < 12; ++k) { w[k] = ci ? 0.0 : sr * o.Ji[k]; w[12 + k] = cj ? 0.0 : sr * o.Jj[k]; w[24 + k] = exc ? 0.0 : sr * o.Jex[k]; }
        w[36] = tdc ? 0.0 : sr * o.Jt[0]; w[37] = tdc ? 0.0 : sr * o.Jt[1];
        w[38] = cl ? 0.0 : sr * o.Jl[0]; w[39] = cl ? 0.0 : sr * o.Jl[1];
        w[40] = sr * o.r[0]; w[41] = sr * o.r[1];
        fj[t] = j; fl[t] = fq.y; fa[t] = i;
        // e_l on the observing pose: one factor's product, no landmark-level sum
        for (int k = 0; k < 6; ++k) {
            const double j0 = cj ? 0.0 : sr * o.Jj[k], j1 = cj ? 0.0 : sr * o.Jj[6 + k];
            const double eo = j0 * w[38] + j1 * w[39];
            w[42 + k] = eo;
            stx<AG>(sb.eO + (size_t)(P.vis_f0 + f) * 6 + k, eo);
        }
    }
    __syncthreads();
    VSTAMP(1);
    // ---- per landmark: pivot, e on the shared groups; 16 lanes per landmark ---------------
    // lane component k: 0..5 anchor pose, 6..11 extrinsic, 12 td, 13 -> (h, b) pivot pieces
    if (t < 16 * nl) {
        const int tl = t >> 4, k = t & 15;
        const int l = lid[tl];
        const int fs = lms[tl], fe = lms[tl + 1];
        const int a = lanc[tl];
        double e = 0, h = 0, b = 0;
        const int off = k < 6 ? k : (k < 12 ? 24 + (k - 6) : 36);
        const int rs = k < 12 ? 6 : 1;      // row stride inside the 2 x n block
        for (int q = fs; q < fe; ++q) {
            const double* w = Jf + q * VF_STRIDE;
            const double l0_ = w[38], l1_ = w[39], r0 = w[40], r1 = w[41];
            if (k < 13) { const double j0 = w[off], j1 = w[off + rs]; e += j0 * l0_ + j1 * l1_; }
            else { h += l0_ * l0_ + l1_ * l1_; b += l0_ * r0 + l1_ * r1; }
        }
        // broadcast (h, b) of lane 13 to the 16-lane group
        h = __shfl(h, (t & ~15) + 13, 64); b = __shfl(b, (t & ~15) + 13, 64);
        const bool cl = !mfree && P.lm_const && P.lm_const[l];
        double Sl = 1.0;
        if (ctl.first) { Sl = (O.jacobi_scaling && !ctl.lin_mode) ? 1.0 / (1.0 + sqrt(h)) : 1.0; if (k == 13) P.Sl[l] = Sl; }
        else Sl = P.Sl[l];
        double dl2 = Sl * Sl * h; dl2 = fmin(fmax(dl2, 1e-6), 1e32);
        const double p = ctl.lin_mode ? h : h + ctl.mu * dl2 / (Sl * Sl);
        // lin_mode 2 (marginalisation): MarginalizationInfo's pseudo inverse zeroes every direction of A_mm whose eigenvalue is
        // <= eps = 1e-8 (marginalization_factor.cpp:277); for a landmark without parallax that direction IS the landmark
        // (eigenvalue = h_ll to first order), so its pivot is dropped and its factors enter the prior as if it were fixed
        const double invp = (cl || !(p > (ctl.lin_mode == 2 ? 1e-8 : 0.0))) ? 0.0 : 1.0 / p;
        double* lr = lmr + tl * 16;
        double* er = Em + tl * RS;
        // (a landmark of another rank's shard is in no chunk of this rank: its entries of the set stay zero here and the all-reduce takes them from the owner)
        if (k == 13) { stx<AG>(sb.hll + l, h); stx<AG>(sb.bl + l, b); stx<AG>(sb.invp + l, invp); stx<AG>(sb.sl + l, Sl); lr[0] = invp; sa[tl] = -invp; er[cR] = b; }
        if (k == 14) stx<AG>(xcand + xo_lam(P) + l, stepped ? ldx<AG>(xcur + xo_lam(P) + l) + cg * ldx<AG>(P.la + l) + cn * ldx<AG>(P.lb + l) : ldx<AG>(xcur + xo_lam(P) + l));      // the same expression the factor threads evaluated (read by the step roles once the candidate is accepted)
        if (k < 13) {
            lr[1 + k] = e; stx<AG>(sb.eA + (size_t)l * 13 + k, e);
            er[k < 6 ? 6 * (a - fa0) + k : cX + (k - 6)] = e;
        }
        // observer columns of e_l: lanes 0..5 of the group walk the factors
        if (k < 6) for (int q = fs; q < fe; ++q) er[6 * (fj[q] - fa0) + k] = Jf[q * VF_STRIDE + 42 + k];
    } else {
        // the threads the landmarks do not use spread the staged Jacobian blocks into the dense operand rows (two per factor, zeroed before): item =
        // (factor, residual row, one of the 19 columns or r); written by the evaluating thread itself the six row / group addresses cost it registers it
        // does not have (19 VGPRs spilled, 3 MB of scratch traffic per launch)
        const int t0 = 16 * nl, nt = blockDim.x - t0;
        int tq = t; asm volatile("" : "+v"(tq));
        for (int it = tq - t0; it < nf * 40; it += nt) {
            const int q = it / 40, e = it - 40 * q, rr = e >= 20 ? 1 : 0, m = e - 20 * rr;
            const double* w = Jf + q * VF_STRIDE;
            int col; double v;
            if (m < 6) { col = 6 * (fa[q] - fa0) + m; v = w[rr * 6 + m]; }
            else if (m < 12) { col = 6 * (fj[q] - fa0) + (m - 6); v = w[12 + rr * 6 + (m - 6)]; }
            else if (m < 18) { col = cX + (m - 12); v = w[24 + rr * 6 + (m - 12)]; }
            else if (m == 18) { col = cT; v = w[36 + rr]; }
            else { col = cR; v = w[40 + rr]; }
            Gm[(2 * q + rr) * RS + col] = v;
        }
    }
    __syncthreads();
    VSTAMP(2);
    // ---- record = sum_f G_f^T G_f - sum_l invp_l E_l^T E_l on the matrix cores: tile g = wave + 8 u of the upper tiles (row by row), accumulators in
    //      registers.  Four rows per MFMA; NK k-steps of operands in flight per batch.
    double* const rec = P.vpart + (size_t)d1.w * 16;
    {
        int tq = t; asm volatile("" : "+v"(tq));      // (opaque: what is derived from it is computed here, not at the top of the role)
        const int wave = __builtin_amdgcn_readfirstlane(tq >> 6), lane = tq & 63;
        constexpr int NK = TS > 2 ? 2 : 4;
        int tI[TS], tJ[TS]; bool on[TS]; d4 acc[TS];
#pragma unroll
        for (int u = 0; u < TS; ++u) {
            const int g = wave + 8 * u;
            on[u] = g < ntile;
            int I = 0, rem = on[u] ? g : 0; while (rem >= T - I) { rem -= T - I; ++I; }      // upper tiles row by row: (I, I + rem)
            tI[u] = I; tJ[u] = I + rem;
            acc[u] = d4{0.0, 0.0, 0.0, 0.0};
        }
        auto mma = [&](const double* G, const double* scl, auto scaled_c) {       // acc += G_A^T G over NK * 4 rows; G_A = G, or the rows of G scaled by scl[row]
            constexpr bool SCALED = decltype(scaled_c)::value;
            const double* p = G + (lane >> 4) * RS + (lane & 15);
            double av[TS][NK], bv[TS][NK];
#pragma unroll
            for (int u = 0; u < TS; ++u) if (on[u]) {
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) { av[u][ks] = p[ks * 4 * RS + (tJ[u] << 4)]; bv[u][ks] = p[ks * 4 * RS + (tI[u] << 4)]; }      // (A from column tile J: the accumulator is the tile's TRANSPOSE)
            }
#pragma unroll
            for (int ks = 0; ks < NK; ++ks) {
                const double sc = SCALED ? scl[4 * ks + (lane >> 4)] : 1.0;
#pragma unroll
                for (int u = 0; u < TS; ++u) if (on[u]) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(SCALED ? sc * av[u][ks] : av[u][ks], bv[u][ks], acc[u], 0, 0, 0);
            }
        };
        for (int r0 = 0; r0 < nrow; r0 += 4 * NK) mma(Gm + r0 * RS, sa, std::false_type{});
        // un-reduced gradient J_c^T r (column r of the factor rows' product) and diagonal of J_c^T J_c: straight from the accumulators into the record
        {
            double* const rbc = rec + ntile * 256; double* const rdg = rbc + 16 * T;
            const int n = lane & 15, m0 = lane >> 4;
#pragma unroll
            for (int u = 0; u < TS; ++u) if (on[u]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + 4 * q;
                    if (tI[u] == tJ[u] && m == n) stx<AG>(rdg + (tI[u] << 4) + m, acc[u][q]);
                    if (tJ[u] == T - 1 && m == (cR & 15)) stx<AG>(rbc + (tI[u] << 4) + n, acc[u][q]);      // (accumulator row m = column of tile J, column n = row of tile I)
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 4 / NK; ++b) mma(Em + b * 4 * NK * RS, sa + b * 4 * NK, std::true_type{});      // the 16 landmark rows, -invp_l on the A operand
        // tiles leave as they sit in the accumulators -- TRANSPOSED (element (r, c) of tile (I, J) at c * 16 + r: the gather's lanes walk the rows r of a
        // column) --, 128-byte lines whole
#pragma unroll
        for (int u = 0; u < TS; ++u) if (on[u]) {
            double* o = rec + (wave + 8 * u) * 256 + (lane >> 4) * 16 + (lane & 15);
#pragma unroll
            for (int q = 0; q < 4; ++q) stx<AG>(o + q * 64, acc[u][q]);
            // (timing experiment, tuning build, VIL_SKIP=1024: every tile is written a SECOND time, into the mirror, and the gather reads both -- twice the record traffic
            //  with the results intact; DESIGN.md "record traffic at K = 20")
#ifdef VIL_TUNING
            if (P.skip_mask & 1024) {
#pragma unroll
                for (int q = 0; q < 4; ++q) stx<AG>(o + P.vmirror + q * 64, acc[u][q]);
            }
#endif
        }
    }
    VSTAMP(3);
    cost = block_sum(cost, red);
    if (t == 0) stx<AG>(rec + ntile * 256 + 32 * T, cost);
    VSTAMP(4);
#ifdef VIL_STAMPS
    if (t == 0) { long long vt1; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(vt1) :: "memory"); atomicMax((unsigned long long*)(P.dbg + 60), (unsigned long long)(vt1 - vt0)); atomicAdd((unsigned long long*)(P.dbg + 61), (unsigned long long)(vt1 - vt0)); }
#endif
}

// ---------------------------------------------------------------------------------------------
template <int NR, bool AG = false>
__device__ __forceinline__ void sweep_lidar(const DevP& P, const SolveOpts& O, int wgc, const double* x, double* sm) {
    const int per = blockDim.x >> 8;                    // 256-point chunks per workgroup
    const int sub = vil_tid() >> 8;
    const int nchunk = NR == 1 ? P.n_pchunk : P.n_echunk;
    const int chunk = wgc * per + sub;
    const bool have = chunk < nchunk;
    const int* ch = (NR == 1 ? P.pchunk : P.echunk) + 3 * (have ? chunk : 0);
    const int start = ch[0], cnt = have ? ch[1] : 0, k = ch[2];
    const int t = vil_tid() & 255;
    sm += sub * 128;
    const double* pose = x + xo_pose(P, k);
    const bool cst = !P.marg && P.pose_const && P.pose_const[k];
    const bool masked = P.marg && !(P.marg == 1 && k == 0);       // marginalisation: only the points of the dropped pose
    double acc[28];
#pragma unroll
    for (int q = 0; q < 28; ++q) acc[q] = 0.0;
    if (t < cnt && !masked) {
        const int f = start + t;
        const M3 R = quatR(pose + 3), Rbl = loadM3(P.Rbl);
        const V3 Pk{pose[0], pose[1], pose[2]}, tbl{P.tbl[0], P.tbl[1], P.tbl[2]};
        double r[NR], J[NR * 6];
        if (NR == 1) {
            const double* c = P.pl_c; const int s = P.pl_stride;
            plane_eval(V3{c[f], c[s + f], c[2 * s + f]}, V3{c[3 * s + f], c[4 * s + f], c[5 * s + f]}, c[6 * s + f], Rbl, tbl, R, Pk, r[0], J, O.precision);
        } else {
            const double* c = P.ed_c; const int s = P.ed_stride;
            edge_eval(V3{c[f], c[s + f], c[2 * s + f]}, V3{c[3 * s + f], c[4 * s + f], c[5 * s + f]}, V3{c[6 * s + f], c[7 * s + f], c[8 * s + f]}, Rbl, tbl, R, Pk, r, J, O.precision);
        }
        double sq = 0;
#pragma unroll
        for (int q = 0; q < NR; ++q) sq += r[q] * r[q];
        double rho, rho1;
        loss_eval(O.lidar_loss, O.lidar_loss_scale, sq, rho, rho1);
        acc[27] = 0.5 * rho;
        if (!cst) {
            // rho1 multiplies J^T J and J^T r (sqrt(rho1) on each factor of the product)
            int idx = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
#pragma unroll
                for (int b = a; b < 6; ++b) {
                    double s = 0;
#pragma unroll
                    for (int q = 0; q < NR; ++q) s += J[q * 6 + a] * J[q * 6 + b];
                    acc[idx++] = rho1 * s;
                }
                double g = 0;
#pragma unroll
                for (int q = 0; q < NR; ++q) g += J[q * 6 + a] * r[q];
                acc[21 + a] = rho1 * g;
            }
        }
    }
    const int wave = t >> 6, lane = t & 63;
    double f0, f1;
    wave_fold<28>(acc, f0, f1);                           // lanes 0..15: totals of values rev4(lane) and 16 + rev4(lane)
    if (lane < 16) { const int q = fold_slot(lane); sm[wave * 28 + q] = f0; if (q + 16 < 28) sm[wave * 28 + q + 16] = f1; }
    __syncthreads();
    const int gchunk = (NR == 1 ? 0 : P.n_pchunk) + chunk;
    if (have && t < 28) stx<AG>(P.lpart + (size_t)gchunk * 28 + t, sm[t] + sm[28 + t] + sm[56 + t] + sm[84 + t]);
}

// ---------------------------------------------------------------------------------------------
__device__ inline const double* prior_block_ptr(const DevP& P, const double* x, int b) {
    const int kind = P.pblk_kind[b], idx = P.pblk_index[b];
    return kind == 0 ? x + xo_pose(P, idx) : (kind == 1 ? x + xo_sb(P, idx) : (kind == 2 ? x + xo_ex(P) : x + xo_td(P)));
}

// [prior] workgroup: r = r0 + J0 dx ;  J0^T J0 = pH, J0^T r0 = pg0, r0^T r0 = pc0 are pre-contracted, so the prior costs
// one n x n gemv (n <= 136: K <= 20).
__device__ __forceinline__ void sweep_prior(const DevP& P, const double* x, double* sm) {
    const int t = vil_tid();
    if (P.pn <= 0) return;
    const int n = P.pn;
    double* dx = sm;           // n
    double* red = sm + 512;
    if (t < P.pnblk) {
        const int kind = P.pblk_kind[t];
        const int gs = kind == 0 || kind == 2 ? 7 : (kind == 1 ? 9 : 1);
        double d[9];
        prior_block_dx(gs, prior_block_ptr(P, x, t), P.px0 + P.pblk_xoff[t], d);
        const int ls = gs == 7 ? 6 : gs;
        for (int k = 0; k < ls; ++k) dx[P.pblk_col[t] + k] = d[k];
    }
    __syncthreads();
    // g = pg0 + pH dx.  pH = J0^T J0 is symmetric: lane = column i (consecutive lanes read consecutive doubles of row k), the rows k are dealt to the
    // eight waves, every load of a thread independent of the others; the eight partial sums of a column meet in LDS in a fixed order.
    // (Before: eight lanes per column over rows k = r, r + 8, ..: 64 different cache lines per wave-load -- 16 us at n = 130, and the chain workgroup of
    //  the chain workgroup of a fallback-structure sweep waits for this record.)
    double* pw = sm + 520;     // 8 x 136
    {
        const int wave = t >> 6, lane = t & 63;
        double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll 6
        for (int k = wave; k < n; k += 8) {
            const double dk = dx[k];
            const double* row = P.pH + (size_t)k * n;
#pragma unroll
            for (int c = 0; c < 3; ++c) { const int i = lane + 64 * c; acc[c] += row[min(i, n - 1)] * dk; }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { const int i = lane + 64 * c; if (i < n) pw[wave * 136 + i] = acc[c]; }
    }
    __syncthreads();
    double part = 0;
    if (t < n) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += pw[w * 136 + t];
        const double g = P.pg0[t] + s;
        part = dx[t] * (P.pg0[t] + g);
        st_ag(P.mpart + t, g);                         // (agent scope: read by the chain workgroup of the same launch, prechain 2)
    }
    part = block_sum(part, red);
    if (t == 0) st_ag(P.mpart + n, 0.5 * (P.pc0[0] + part));
}

// [rel] workgroup: the scan-to-scan ICP and LPS AutoDiff factors
template <bool AG = false>
__device__ __forceinline__ void sweep_misc(const DevP& P, const SolveOpts& O, const double* x, double* sm) {
    const int t = vil_tid();
    // ---- ICP (4 pose blocks) and LPS (2 pose blocks): thread per (factor, block) ----------------------------
    double* Jb = sm;              // up to 12 factors x 4 blocks x 21
    double* rb = sm + 12 * 84;    // 12 x 3
    const int n_rel = P.n_icp + P.n_lps;
    if (n_rel == 0) return;
    for (int it = t; it < 28 * n_rel; it += blockDim.x) {
        const int f = it / 28, rem = it - 28 * f, b = rem / 7, k = rem - 7 * b;
        double r3[3], d3[3] = {0.0, 0.0, 0.0};
        bool live = true;
        const int* id;
        if (f < P.n_icp) {
            id = P.icp_ids + 4 * f;
            icp_eval1(P.icp_c + (size_t)f * 10, x + xo_pose(P, id[0]), x + xo_pose(P, id[1]), x + xo_pose(P, id[2]), x + xo_pose(P, id[3]), b, k, r3, d3);
        } else if (b < 2) {
            id = P.lps_ids + 2 * (f - P.n_icp);
            lps_eval1(P.lps_c + (size_t)(f - P.n_icp) * 7, x + xo_pose(P, id[0]), x + xo_pose(P, id[1]), b, k, r3, d3);
        } else { live = false; id = P.lps_ids; r3[0] = r3[1] = r3[2] = 0.0; }
        if (live && !P.marg && P.pose_const && P.pose_const[id[b]]) d3[0] = d3[1] = d3[2] = 0.0;
        for (int q = 0; q < 3; ++q) Jb[(f * 4 + b) * 21 + 7 * q + k] = d3[q];
        if (b == 0 && k == 0) for (int q = 0; q < 3; ++q) rb[f * 3 + q] = r3[q];
    }
    __syncthreads();
    if (!O.autodiff_quirk) {   // mathematically-correct tangent Jacobian instead of the raw d/d(qx,qy,qz) columns
        for (int it = t; it < 4 * n_rel; it += blockDim.x) {
            const int f = it >> 2, b = it & 3;
            if (f >= P.n_icp && b >= 2) continue;
            const int* id = f < P.n_icp ? P.icp_ids + 4 * f : P.lps_ids + 2 * (f - P.n_icp);
            tangent_fix(x + xo_pose(P, id[b]), Jb + (f * 4 + b) * 21);
        }
        __syncthreads();
    }
    // per factor: 24 x 24 block over the 4 x 6 local columns (LPS: blocks 2,3 are zero) + gradient + cost
    double* out0 = P.mpart + (P.pn > 0 ? P.pn + 1 : 0);
    for (int e = t; e < n_rel * 601; e += blockDim.x) {
        const int f = e / 601, q = e - f * 601;
        const double* r = rb + f * 3;
        double rho, rho1;
        loss_eval(O.rel_loss, O.rel_loss_scale, r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho, rho1);
        if (P.marg && !(P.marg == 1 && (f < P.n_icp ? f == P.marg_icp : f - P.n_icp == P.marg_lps))) { rho = 0.0; rho1 = 0.0; }    // estimator.cpp:1508-1533
        double v;
        if (q < 576) {
            const int a = q / 24, b = q % 24;
            const double* Ja = Jb + (f * 4 + a / 6) * 21 + a % 6;
            const double* Jc = Jb + (f * 4 + b / 6) * 21 + b % 6;
            v = rho1 * (Ja[0] * Jc[0] + Ja[7] * Jc[7] + Ja[14] * Jc[14]);
        } else if (q < 600) {
            const int a = q - 576;
            const double* Ja = Jb + (f * 4 + a / 6) * 21 + a % 6;
            v = rho1 * (Ja[0] * r[0] + Ja[7] * r[1] + Ja[14] * r[2]);
        } else v = 0.5 * rho;
        stx<AG>(out0 + e, v);
    }
}

// local index (0..29) of reduced column `col` inside IMU factor (i,j), or -1
__device__ __forceinline__ int imu_local(const DevP& P, int i, int j, int col) {
    int d = col - col_pose(P, i); if (d >= 0 && d < 6) return d;
    d = col - col_sb(P, i); if (d >= 0 && d < 9) return 6 + d;
    d = col - col_pose(P, j); if (d >= 0 && d < 6) return 15 + d;
    d = col - col_sb(P, j); if (d >= 0 && d < 9) return 21 + d;
    return -1;
}

}  // namespace vd

// Gather of the sweep's partial records into the dense reduced system of the candidate set:
//   S' (D x D, both triangles), gred, bc, diag, cost.   32 lower-triangle entries per workgroup, 8 threads per
//   entry splitting the sum over the visual records; fixed summation order everywhere (the records themselves are matrix-core
//   contractions in a fixed row order: two runs agree bit for bit).  A visual record holds the upper 16 x 16 tiles of its chunk's
//   frame window (sweep_visual): an entry (i, j) of S' reads the records whose window covers both columns.
//   The last workgroup handles the vectors and the cost.
#define RED_EPW 32
#define VIS_TAB 512      // visual record descriptors a gather workgroup stages in LDS (8 kB)
// AG: the results are stored at agent scope (gather + step in one launch: the master reads them in the same launch)
// EPW: entries per workgroup = blockDim / 8 (32 with 256 threads: the gather kernel; 64 with 512: the gather workgroups of the merged launch,
// whose register allocation admits one workgroup per compute unit whatever its thread count)
// workgroups of the gather: entries of the visual triangle (32 slices each: EPW / 4 per workgroup), the other entries of the lower triangle (8 slices:
// EPW per workgroup), the 2 D vector entries (32 slices), the cost
// entries of the visual triangle per gather workgroup: a quarter of EPW (32 slices per entry: one round of loads covers 256 records) -- but EPW (8 slices) for
// the large windows (K > 16): at K = 20 the quarter made 508 workgroups of ~6 us each for the ~210 compute units the waiting roles leave free -- the launch's
// dynamic LDS allows one workgroup per unit -- i.e. three rounds, and the last gather workgroup saw the visual flags 21 us after the last visual record
__host__ __device__ inline int gather_epv(int NV, int epw) { return NV > 100 ? epw : epw / 4; }
__host__ __device__ inline int gather_vblocks(int NV, int epw) { const int epv = gather_epv(NV, epw); return ((NV * (NV + 1)) / 2 + epv - 1) / epv; }
// pose_only: a solve whose speed-bias chain is eliminated from the IMU / prior records directly (vil_prechain.hpp) reads S' on the visual sub-space only;
// of the rest (39 k of the 47 k entries at K = 20) the gather then forms just the diagonal (the dogleg scaling)
__host__ __device__ inline int gather_sblocks(int D, int NV, int epw, bool pose_only) { return gather_vblocks(NV, epw) + ((pose_only ? D - NV : (D * (D + 1)) / 2 - (NV * (NV + 1)) / 2) + epw - 1) / epw; }
__host__ __device__ inline int gather_blocks(int D, int NV, int epw, bool pose_only) { return gather_sblocks(D, NV, epw, pose_only) + (2 * D + epw / 4 - 1) / (epw / 4) + 1; }
// FUSED: the sweep's workgroups are workgroups of the SAME launch (the one-launch iteration, vil_iter.hpp): once its own tables are staged the workgroup waits
// for their flags (P.sflag = epoch, n_sw of them), every record is read at agent scope, and the scratch arrays live in the workgroup's dynamic LDS behind vtab
// the candidate's cost from the sweep roles' partials: 8 EPW threads, a fixed order (per-thread strided sums, wave sums, the waves in order) -- the gather's cost item and
// the persistent solve's cost-first judgement (vil_step.hpp) call THIS, so that both see the same bits.  The total is valid in thread 0; red: >= EPW / 8 doubles of LDS.
template <int EPW, bool FUSED>
__device__ __forceinline__ double gather_cost(const DevP& P, double* const red) {
    using namespace vd;
    const int t = vil_tid();
    auto rd = [](const double* p) -> double { return ldx<FUSED>(p); };
    const int4* const vrec = (const int4*)P.vrec;
    const int n_rel = P.n_icp + P.n_lps;
    const double* rel0 = P.mpart + (P.pn > 0 ? P.pn + 1 : 0);
    double c = 0.0;
    for (int w = t; w < P.n_vwg; w += 8 * EPW) { const int4 ds = vrec[w]; c += rd(P.vpart + (size_t)ds.x * 16 + vis_ntile(ds.w) * 256 + 32 * ds.w); }
    for (int q = t; q < P.n_pchunk + P.n_echunk; q += 8 * EPW) c += rd(P.lpart + (size_t)q * 28 + 27);
    for (int f = t; f < P.n_imu; f += 8 * EPW) c += rd(P.ipart + (size_t)f * 931 + 930);
    for (int f = t; f < n_rel; f += 8 * EPW) c += rd(rel0 + (size_t)f * 601 + 600);
    if (t == 0 && P.pn > 0) c += rd(P.mpart + P.pn);
    c = wave_sum(c);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = c;
    __syncthreads();
    double tot = 0.0;
    if (t == 0) for (int w = 0; w < EPW / 8; ++w) tot += red[w];      // (EPW / 8 waves)
    return tot;
}
template <bool AG = false, int EPW = RED_EPW, bool FUSED = false>
__device__ __forceinline__ void reduce_gather(const DevP& P, const Ctl& ctl, const int blk /* gather workgroup index */, int4* const vtab /* LDS, VIS_TAB entries */, const int epoch = 0) {
    using namespace vd;
    auto put = [](double* p, double v) { if (AG) st_ag(p, v); else *p = v; };
    auto rd = [](const double* p) -> double { return ldx<FUSED>(p); };          // a sweep record
    // the visual workgroups are the sweep's longest (15 us against ~9 for everything else): a gather workgroup first waits for the OTHER roles' flags, sums their
    // records while the visual workgroups are still at work, and only then waits for the visual records -- one round of loads behind the last of them
    auto wait_sweep = [&](const bool visual) {
        if constexpr (FUSED) {
            // (ONE word per group of roles, published by the master workgroup -- vil_step.hpp: it polls the roles' flags once for everybody.  Every gather workgroup
            //  polling every role's flag was n_gather x n_sweep lanes on a few dozen cache lines: 166 x 1700 at configs[2], where the gather saw the visual flags
            //  4.3 us after the last visual record)
            if (vil_tid() == 0) spin_until_eq(P.sall + (visual ? 32 : 16), epoch, P.abortf);
            __syncthreads();
            if (visual && vil_tid() == 0) prof_stamp(P, epoch - 1, 6);
        }
    };
    if (vil_tid() >= 8 * EPW) return;
    const int cand = 1 - ctl.cur;
    SysBuf sb = P.sys[cand];
    const int D = P.D, NV = P.NV, K = P.K, t = vil_tid();
    const int n_rel = P.n_icp + P.n_lps;
    const double* rel0 = P.mpart + (P.pn > 0 ? P.pn + 1 : 0);
    const int NL = (D * (D + 1)) >> 1, NVT = (NV * (NV + 1)) >> 1;
    constexpr int EPV = EPW / 4;               // entries per workgroup where the visual records are summed: 32 slices per entry, one round of loads
    const bool pose_only = P.gather_pose_only != 0;
    const int nVblk = gather_vblocks(NV, EPW), nSblk = gather_sblocks(D, NV, EPW, pose_only);
    constexpr int NTAB = 64 + 48 + 2 * 66 + 24;
    double (*part)[8 * EPW]; int* tab; double* red;
    if constexpr (FUSED) { part = reinterpret_cast<double (*)[8 * EPW]>(vtab + VIS_TAB); tab = reinterpret_cast<int*>(part + 2); red = reinterpret_cast<double*>(tab + NTAB + (NTAB & 1)); }
    else { __shared__ double part_st[2][8 * EPW]; __shared__ int tab_st[NTAB]; __shared__ double red_st[8]; part = part_st; tab = tab_st; red = red_st; }
    // tab: imu (i, j) pairs | ICP/LPS pose ids (4 per factor) | LiDAR chunk ranges per pose | visual records whose window starts at or before frame f
    // vtab: the visual records' descriptors {offset / 16, first frame, frames, column tiles}, staged per workgroup (records beyond VIS_TAB: read from memory)
    int* t_imu = tab; int* t_rel = tab + 64; int* t_lch = tab + 112; int* t_wend = tab + 244;
    const int4* const vrec = (const int4*)P.vrec;
    // (the hot loops run on the staged descriptors only, a tail loop on those beyond VIS_TAB -- a per-lane choice between the two inside a round would be
    //  an exec-mask branch with its own wait around every descriptor, which serialises the loads of the round)
    // local column of reduced column c (< NV) in a record's window, or -1: pose columns of frames [fa, fa + sp), then extrinsic / td
    auto vlocal = [&](int c, int fa, int sp) -> int { const int d = c - 6 * fa; return c >= 6 * K ? 6 * sp + (c - 6 * K) : (((unsigned)d < (unsigned)(6 * sp)) ? d : -1); };
    auto stage = [&](bool visual) {            // the small index tables, once per workgroup
        if (t < 2 * P.n_imu && t < 64) t_imu[t] = (t & 1) ? P.imu_j[t >> 1] : P.imu_i[t >> 1];
        if (t >= 64 && t < 64 + 4 * n_rel) { const int q = t - 64, f = q >> 2, b = q & 3; t_rel[q] = f < P.n_icp ? P.icp_ids[4 * f + b] : (b < 2 ? P.lps_ids[2 * (f - P.n_icp) + b] : -1); }
        if (t >= 128 && t < 128 + 2 * (K + 1) && t < 128 + 132) t_lch[t - 128] = P.lchunk_pose[t - 128];
        if (visual) {
            for (int e = t; e < min(P.n_vwg, VIS_TAB); e += 8 * EPW) vtab[e] = vrec[e];
            if (t >= 8 * EPW - 32 && t < 8 * EPW - 32 + K) t_wend[t - (8 * EPW - 32)] = P.vwend[t - (8 * EPW - 32)];      // (K <= 20)
        }
        __syncthreads();
    };
    if (blk < nSblk) {
        if (P.skip_mask & 32) return;
        const bool vis = blk < nVblk;          // (workgroup-uniform)
        stage(vis);
        const int epvb = gather_epv(NV, EPW);
        const int epw = vis ? epvb : EPW, ns = (8 * EPW) / epw;
        const int el = t & (epw - 1), slice = t / epw;
        int idx = vis ? blk * epvb + el : NVT + (blk - nVblk) * EPW + el;
        int i = 0, j = 0;
        bool ok = idx < (vis ? NVT : NL);
        if (!vis && pose_only) {               // only the diagonal of the non-visual part
            const int q = (blk - nVblk) * EPW + el;
            ok = q < D - NV; i = j = NV + min(q, D - NV - 1);
        } else
        if (ok) {
            int a = (int)((sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);      // (the two loops below make it exact; the fp64 square root is a 1 k-cycle chain)
            while (((a + 1) * (a + 2)) / 2 <= idx) ++a;
            while ((a * (a + 1)) / 2 > idx) --a;
            // row a of the lower triangle, column b <= a: rows below NV are the visual triangle (idx < NVT).  Read as the upper entry (i, j) = (b, a):
            // consecutive lanes -> consecutive rows i of one column j, which are consecutive addresses in the (transposed) tiles of the visual records
            j = a; i = idx - (a * (a + 1)) / 2;
        }
        wait_sweep(false);
        // the non-visual contributions are spread over the slices of an entry
        double ms = 0.0;
        if (ok) {
            if (j < 6 * K && i / 6 == j / 6) {                 // LiDAR plane and edge points: pose-diagonal blocks.  The chunk records of a pose
                const int k = i / 6, a = i - 6 * k, b = j - 6 * k;   // (one per 256 points: 9 per pose at 24 k points, 37 at 96 k) are dealt to the slices
                const int li = a * 6 - ((a * (a - 1)) >> 1) + (b - a);   // of the entry -- one slice walking them all was the longest chain of this kernel
                for (int c = t_lch[k] + slice; c < t_lch[k + 1]; c += ns) ms += rd(P.lpart + (size_t)c * 28 + li);
                for (int c = t_lch[K + 1 + k] + slice; c < t_lch[K + 2 + k]; c += ns) ms += rd(P.lpart + (size_t)(P.n_pchunk + c) * 28 + li);
            }
            // ICP / LPS and IMU factors are DEALT to the slices of an entry (factor f of the <= 12 ICP / LPS factors on slice 8 + f, IMU factor f on slice 20 + f, modulo the
            // entry's slices): every slice issues at most a load or two -- one slice walking all ICP / LPS factors was a chain of up to six dependent round trips
            // (~1.5 us each at agent scope), the longest path of the gather
            if (j < 6 * K) {                                   // ICP / LPS blocks live on pose columns
                const int pi = i / 6, pj = j / 6, ri = i - 6 * pi, rj = j - 6 * pj;
                for (int f = (slice + ns - (8 % ns)) % ns; f < n_rel; f += ns)
                    for (int ba = 0; ba < 4; ++ba) if (t_rel[4 * f + ba] == pi) for (int bb = 0; bb < 4; ++bb) if (t_rel[4 * f + bb] == pj)
                        ms += rd(rel0 + (size_t)f * 601 + (ba * 6 + ri) * 24 + bb * 6 + rj);
            }
            for (int f = (slice + ns - (20 % ns)) % ns; f < P.n_imu; f += ns) {
                const int la = imu_local(P, t_imu[2 * f], t_imu[2 * f + 1], i), lb = imu_local(P, t_imu[2 * f], t_imu[2 * f + 1], j);
                const double v = rd(P.ipart + (size_t)f * 931 + max(la, 0) * 30 + max(lb, 0));      // (unconditional load + select)
                ms += (la >= 0 && lb >= 0) ? v : 0.0;
            }
            if (slice == 6 && P.pn > 0) { const int pi = P.pinv[i], pj = P.pinv[j]; if (pi >= 0 && pj >= 0) ms += P.pH[(size_t)pi * P.pn + pj]; }
        }
        if (vis) wait_sweep(true);      // (workgroup-uniform; the other entries have no visual part)
        double vs = 0.0, vdg = 0.0;
        if (vis && ok) {
            // eight records per round and thread, every load issued before the first add (clamped record / element + select: no predicated loads); with
            // 32 slices per entry one round covers 256 records.  The records are sorted by first frame: an entry whose row sits in frame f is covered
            // by the first vwend[f] of them at most.  A wave that holds a diagonal entry also fetches the records' un-reduced diagonals in the same round
            constexpr int U = 8;
            const int nw = i < 6 * K ? t_wend[i / 6] : P.n_vwg, nws = min(nw, VIS_TAB), wlast = max(nws - 1, 0);
            const bool wdiag = __ballot(i == j) != 0ull;
            auto fetch = [&](const int4 ds, const bool live, double& va_, double& vd_) {
                const int T = ds.w, il_ = vlocal(i, ds.y, ds.z), jl_ = vlocal(j, ds.y, ds.z);
                const bool in = il_ >= 0 && jl_ >= 0 && live;
                const int il = in ? il_ : 0, jl = in ? jl_ : 0, I = il >> 4, J = jl >> 4;
                const double* r = P.vpart + (size_t)ds.x * 16;
                // (only the records that cover the entry are loaded: agent-scope loads cost per lane, and at K = 20 two records in three do not)
                va_ = 0.0; vd_ = 0.0;
                if (in) va_ = rd(r + (I * T - ((I * (I - 1)) >> 1) + (J - I)) * 256 + (jl & 15) * 16 + (il & 15));      // tile (I, J) holds its transpose
#ifdef VIL_TUNING
                if (in && (P.skip_mask & 1024)) { const double m2 = rd(r + P.vmirror + (I * T - ((I * (I - 1)) >> 1) + (J - I)) * 256 + (jl & 15) * 16 + (il & 15)); va_ = 0.5 * va_ + 0.5 * m2; }      // (the mirror: the same value)
#endif
                if (wdiag && in && i == j) vd_ = rd(r + vis_ntile(T) * 256 + 16 * T + il);
            };
            for (int w = slice; w < nws; w += ns * U) {
                double a[U], d[U];
#pragma unroll
                for (int u = 0; u < U; ++u) fetch(vtab[min(w + ns * u, wlast)], w + ns * u < nws, a[u], d[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) { vs += a[u]; vdg += d[u]; }
            }
            for (int w = VIS_TAB + slice; w < nw; w += ns) { double a, d; fetch(vrec[w], true, a, d); vs += a; vdg += d; }      // (windows with more than VIS_TAB chunks)
        }
        part[0][slice * epw + el] = vs + ms; part[1][slice * epw + el] = vdg - vs;     // [1]: un-reduced minus reduced visual diagonal
        __syncthreads();
        if (slice != 0 || !ok) return;
        double s = 0.0, dd = 0.0;
        for (int q = 0; q < ns; ++q) { s += part[0][q * epw + el]; dd += part[1][q * epw + el]; }
        put(sb.S + (size_t)i * D + j, s);
        put(sb.S + (size_t)j * D + i, s);
        if (i == j) put(sb.diag + i, s + (i < NV ? dd : 0.0));   // un-reduced diagonal: add back the Schur term of the visual part
        return;
    }
    // ---- gradient vectors bc / gred: 2D entries, 32 slices each ----------------------------------------------------------------------
    const int nVblk2 = (2 * D + EPV - 1) / EPV;
    if (blk < nSblk + nVblk2) {
        if (P.skip_mask & 64) return;
        stage(true);
        const int el = t & (EPV - 1), slice = t / EPV;
        const int v = (blk - nSblk) * EPV + el;
        const bool ok = v < 2 * D;
        const int which = v >= D ? 1 : 0, i = which ? v - D : v;       // 0: bc, 1: gred
        double acc = 0.0, accv = 0.0;                  // the other roles' records first (their workgroups finish long before the visual ones), then the visual records
        wait_sweep(false);
        if (ok) {
            if (i < 6 * K) {
                const int k = i / 6, a = i - 6 * k;
                for (int c = t_lch[k] + slice; c < t_lch[k + 1]; c += 32) acc += rd(P.lpart + (size_t)c * 28 + 21 + a);      // (chunk records dealt to the slices, as above)
                for (int c = t_lch[K + 1 + k] + slice; c < t_lch[K + 2 + k]; c += 32) acc += rd(P.lpart + (size_t)(P.n_pchunk + c) * 28 + 21 + a);
                for (int f = (slice + 32 - 8) % 32; f < n_rel; f += 32) for (int ba = 0; ba < 4; ++ba) if (t_rel[4 * f + ba] == k) acc += rd(rel0 + (size_t)f * 601 + 576 + ba * 6 + a);      // (factors dealt to the slices, as above)
            }
            for (int f = (slice + 32 - 20) % 32; f < P.n_imu; f += 32) { const int la = imu_local(P, t_imu[2 * f], t_imu[2 * f + 1], i); const double v = rd(P.ipart + (size_t)f * 931 + 900 + max(la, 0)); acc += la >= 0 ? v : 0.0; }
            if (slice == 6 && P.pn > 0) { const int pi = P.pinv[i]; if (pi >= 0) acc += rd(P.mpart + pi); }
        }
        wait_sweep(true);
        if (ok) {
            if (i < NV) {                              // bc: the record's vector; gred: column r of its last tile column.  Eight records per round and thread, as above
                const int nw = i < 6 * K ? t_wend[i / 6] : P.n_vwg, nws = min(nw, VIS_TAB), wlast = max(nws - 1, 0);
                auto fetch = [&](const int4 ds, const bool live) -> double {
                    const int T = ds.w, il_ = vlocal(i, ds.y, ds.z), il = max(il_, 0), I = il >> 4;
                    const double* r = P.vpart + (size_t)ds.x * 16;
                    const double val = which ? rd(r + (I * T - ((I * (I - 1)) >> 1) + (T - 1 - I)) * 256 + ((6 * ds.z + 7) & 15) * 16 + (il & 15)) : rd(r + vis_ntile(T) * 256 + il);
                    return (il_ >= 0 && live) ? val : 0.0;
                };
                for (int w = slice; w < nws; w += 256) {
                    double a[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) a[u] = fetch(vtab[min(w + 32 * u, wlast)], w + 32 * u < nws);
#pragma unroll
                    for (int u = 0; u < 8; ++u) accv += a[u];
                }
                for (int w = VIS_TAB + slice; w < nw; w += 32) accv += fetch(vrec[w], true);
            }
        }
        part[0][slice * EPV + el] = accv + acc;
        __syncthreads();
        if (slice != 0 || !ok) return;
        double sum = 0.0;
        for (int q = 0; q < 32; ++q) sum += part[0][q * EPV + el];
        put((which ? sb.gred : sb.bc) + i, sum);
        return;
    }
    // ---- cost (one workgroup, tree reduction) ---------------------------------------------------------------------
    if (P.skip_mask & 64) return;
    wait_sweep(false); wait_sweep(true);
    const double tot = gather_cost<EPW, FUSED>(P, red);
    if (t == 0) put(sb.cost, tot);
}

#ifndef VIL_PERSIST_TU
// Gather of the sweep's partial records (reduce_gather above) as a launch of its own: vil_linearize / the marginalisation (no step kernel
// behind it), the multi-GPU path (the collective sits between gather and step) and windows too large for the merged launch -- for
// those the grid carries one more workgroup per 16 x 16 tile of W W^T (vil_prechain.hpp; n_gather = the gather workgroups).  A solve of a
// small window on one GPU gathers inside k_step (rs_merged).
__global__ __launch_bounds__(VIL_THREADS) void k_reduce(DevP P, int n_gather) {
    const Ctl ctl = *P.ctl;
    if (ctl.done) return;
    if ((int)blockIdx.x >= n_gather) { vd::prechain_ww_tile(P, (int)blockIdx.x - n_gather); return; }
    __shared__ int4 vtab[VIS_TAB];
    reduce_gather(P, ctl, (int)blockIdx.x, vtab);
}
#endif

// the IMU / prior workgroup `slot` of this launch has written its record (read by the chain workgroup of the same launch, prechain 2).
// Epoch of the flags: solve generation + Ctl::swe, which only the step kernel advances.
// The record went out with agent-scope stores (the level the XCDs share), so the flag only has to be ordered behind every thread's own stores:
// no release fence (which writes the whole XCD's L2 back while the visual workgroups are filling it), the reader polls relaxed and loads at agent scope.
__device__ __forceinline__ void sweep_signal(const DevP& P, const Ctl& ctl, int slot) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's stores of the record (__syncthreads alone does not wait for global stores)
    __syncthreads();
    if (vil_tid() == 0) vd::st_ag(P.swflag + slot, (int)((((unsigned)ctl.gen) << 12) + (unsigned)ctl.swe + 1u));
}

// The sweep: grid = n_imu + 2 (+ 1: prechain 2) + n_vwg + ceil(n_pchunk / 2) + ceil(n_echunk / 2) workgroups of VIL_SWEEP_THREADS threads.
// Workgroup order: [imu x n_imu | prior | rel | (chain) | visual x n_vwg | plane | edge] -- the short roles the chain workgroup waits for
// come first (roles: top of this file)
// FUSED: the roles as workgroups of the one-launch iteration (k_iter, vil_iter.hpp): everything another workgroup of the launch reads goes out at agent scope,
// and every workgroup ends by posting the launch epoch in P.sflag[b] (the gather workgroups wait for all of them, the chain workgroup for the IMU / prior ones)
#define VIL_XSTAGE 2048      // doubles into a sweep role's dynamic LDS where a one-launch role keeps its copy of the state's camera part (16 K + 8 <= 328 doubles)
template <int TS, bool FUSED>      // TS: accumulator tiles per wave of the visual role
__device__ __forceinline__ void sweep_body(const DevP& P, const SolveOpts& O, const Ctl& ctl, double* const sm, const int blk, const double* const xlds = nullptr, const bool imu_resident = false) {
    const int cand = 1 - ctl.cur;
    const double* x = (FUSED && VIL_X_IN_LDS) ? xlds : P.x[cand];
    SysBuf sb = P.sys[cand];
    int b = blk;
    const bool pre = !FUSED && P.prechain == 2 && ctl.lin_mode == 0;
    if (FUSED && vil_tid() == 0) prof_stamp(P, ctl.n_sweeps, 0, true);
    auto posted = [&]() {
        if constexpr (FUSED) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's stores of the record (__syncthreads alone does not wait for global stores)
            __syncthreads();
            if (vil_tid() == 0 && !(blk == P.drop_role && ctl.n_sweeps == P.drop_launch)) { const int ep = (int)((((unsigned)ctl.gen) << 12) + (unsigned)ctl.n_sweeps + 1u); vd::st_ag(P.sflag + blk, ep); if (blk < P.n_imu) vd::st_ag(P.cflag + blk, ep); prof_stamp(P, ctl.n_sweeps, blk < P.n_imu ? 2 : (blk == P.n_imu ? 15 : 1)); }      // (an IMU role's chain flag: up already unless the role left early)
        }
    };
    // FUSED: the prior, ICP / LPS and LiDAR roles hand pointers into the state to the factor code -- they read the camera part from a copy in LDS, fetched at agent scope
    // (sm + VIL_XSTAGE: past what any of these roles uses, inside what every one-launch workgroup owns)
    auto xstage = [&]() -> const double* {
        if constexpr (!FUSED || VIL_X_IN_LDS) return x;
        else {
            double* xl = sm + VIL_XSTAGE;
            for (int e = vil_tid(); e < 16 * P.K + 8; e += blockDim.x) xl[e] = vd::ld_ag(x + e);
            __syncthreads();
            return xl;
        }
    };
    if (b < P.n_imu) { if (!(P.skip_mask & 2)) vd::sweep_imu(P, O, b, x, sm, FUSED ? ctl.n_sweeps : -1, FUSED, (int)((((unsigned)ctl.gen) << 12) + (unsigned)ctl.n_sweeps + 1u), FUSED && VIL_X_IN_LDS, imu_resident); if (pre) sweep_signal(P, ctl, b); posted(); return; }
    b -= P.n_imu;
    if (b == 0) { if (!(P.skip_mask & 16)) vd::sweep_prior(P, xstage(), sm); if (pre) sweep_signal(P, ctl, P.n_imu); posted(); return; }
    if (b == 1) {
        if (!(P.skip_mask & 16)) vd::sweep_misc<FUSED>(P, O, xstage(), sm);
        if (P.world > 1) {                               // factor set sharded over ranks: the visual workgroups of this rank form the candidate inverse depth of
            const double* xcur = P.x[ctl.cur];           // the landmarks it owns; every rank holds la / lb of ALL landmarks (the step kernel runs on the all-reduced
            double* xcand = P.x[1 - ctl.cur];            // system), so the rest is filled in here and the states stay identical on all ranks
            const bool stepped = ctl.cg != 0.0 || ctl.cn != 0.0;      // (as in sweep_visual: stale la / lb are not multiplied by zero)
            for (int l = vil_tid(); l < P.L; l += blockDim.x) xcand[xo_lam(P) + l] = stepped ? xcur[xo_lam(P) + l] + ctl.cg * P.la[l] + ctl.cn * P.lb[l] : xcur[xo_lam(P) + l];
        }
        posted();
        return;
    }
    b -= 2;
    if (!FUSED && P.prechain == 2) { if (b == 0) { if (pre) vd::prechain_wg(P, ctl, O.jacobi_scaling, sm, 0, true); return; } b -= 1; }
    // visual workgroups next: the longest-running factor role
    if (b < P.n_vwg) { if (!(P.skip_mask & 1)) vd::sweep_visual<TS, FUSED>(P, O, ctl, b, x, sb, sm); posted(); return; }
    b -= P.n_vwg;
    // LiDAR roles: 2 chunks of 256 points per pass, P.lidar_rep passes per workgroup (1 unless the window has more sweep roles than the device has compute units:
    // configs[2]'s 120 k points are 235 two-chunk workgroups -- with the visual roles two dispatch rounds, and the gather workgroups of a one-launch iteration queue behind them)
    const int per = VIL_SWEEP_THREADS / 256, R = max(P.lidar_rep, 1), npw = (P.n_pchunk + per * R - 1) / (per * R);
    const double* const xl = (FUSED && (P.skip_mask & 12) != 12) ? xstage() : x;
    if (b < npw) { if (!(P.skip_mask & 4)) for (int r = 0; r < R; ++r) { if (r) __syncthreads(); vd::sweep_lidar<1, FUSED>(P, O, b * R + r, xl, sm); } posted(); return; }
    b -= npw;
    if (!(P.skip_mask & 8)) for (int r = 0; r < R; ++r) { if (r) __syncthreads(); vd::sweep_lidar<3, FUSED>(P, O, b * R + r, xl, sm); }
    posted();
}

#ifndef VIL_PERSIST_TU
template <int TS>
__global__ __launch_bounds__(VIL_SWEEP_THREADS) void k_sweep(DevP P, SolveOpts O) {
    extern __shared__ double sm[];
    const Ctl ctl = *P.ctl;
    if (ctl.done) return;                                // (the result of a finished solve is written out by k_finish, vil_finish.hpp)
    if (blockIdx.x == 0 && vil_tid() == 0) P.ctl->n_sweeps = ctl.n_sweeps + 1;   // live (not early-exited) launches, for the profiler
    sweep_body<TS, false>(P, O, ctl, sm, (int)blockIdx.x);
}
#endif
