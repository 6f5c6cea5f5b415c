// The factor sweep: ONE launch evaluates every residual block of the window at the candidate state
// and contracts it towards the Schur-complement normal equations of the candidate linearisation
// (what ceres' Evaluate + SchurEliminator do per trust-region iteration behind estimator.cpp:1414).
// No global atomics: every workgroup accumulates in registers / LDS and writes ONE partial record;
// k_reduce then gathers the partials into the dense reduced system S', g (deterministic order).
//
// Work-group roles by blockIdx (workgroups of VIL_SWEEP_THREADS = 512 threads):
//   [imu]     one WG per IMU factor: lane 0 forms the raw 15x30 block, the WG whitens with the
//             pre-factored sqrt-information and contracts to a 30x30 H block
//   [visual]  one WG per group of landmark sub-chunks (<= VIL_VCHUNK_LM landmarks / VIL_VCHUNK_F factors each):
//             thread-per-factor evaluation staged in LDS, 16 lanes per landmark for the Schur pivots, then three uniform
//             passes of block outer products (shared x shared, shared x observer, observer x observer) accumulated
//             with ds_add_f64 into an LDS-resident packed triangle of the (6K+7)^2 visual sub-space
//   [plane]/[edge] 256 threads per <=256 pose-uniform LiDAR points (two chunks per WG): thread-per-point evaluation,
//             wave64 butterfly reduction of the 6x6 + 6 + cost
//   [prior]   n x n gemv on the pre-contracted J0^T J0
//   [rel]     ICP and LPS AutoDiff factors (scalar forward-mode duals, thread = (factor, block, coordinate))
#pragma once
#include "vil_dev.hpp"
#include "vil_factors.hpp"
#include "vil_finish.hpp"
#include "vil_prechain.hpp"

namespace vd {

__device__ __forceinline__ double block_sum(double v, double* red /*>= 4 doubles*/) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
    return t;
}
__device__ __forceinline__ void lds_add(double* p, double v) { unsafeAtomicAdd(p, v); }   // ds_add_f64
// a (wave-uniform) pointer the compiler cannot see through: address arithmetic on it stays where it is written instead of being hoisted to the top of
// the role and kept in registers across the factor evaluation, the kernel's register peak
template <class T> __device__ __forceinline__ T* opaque(T* p) { asm volatile("" : "+s"(p)); return p; }
__device__ __forceinline__ int tri_idx(int NV, int i, int j) { if (i > j) { const int t = i; i = j; j = t; } return i * NV - ((i * (i - 1)) >> 1) + (j - i); }

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sweep_imu(const DevP& P, const SolveOpts& O, int f, const double* x, double* sm) {
    double* Jraw = sm;            // 450
    double* rr = sm + 450;        // 15
    double* UJ = sm + 480;        // 450
    double* Ur = sm + 930;        // 15
    const double* c = P.imu_c + (size_t)f * 287;
    double* out = P.ipart + (size_t)f * 931;
    const int t = threadIdx.x;
    const int i = P.imu_i[f], j = P.imu_j[f];
    if (c[16] > 10.0 || (P.marg && !(P.marg == 1 && i == 0 && j == 1 && c[16] < 10.0))) { for (int e = t; e < 931; e += blockDim.x) out[e] = 0.0; return; }   // estimator.cpp:1182 / :1535
    for (int e = t; e < 450; e += blockDim.x) Jraw[e] = 0.0;
    __syncthreads();
    if (t <= IMU_NBLOCKS) {   // lanes 0..16: one 3x3 block each (common terms recomputed per lane); lane 17: residual
        ImuCommon o;
        imu_common(c, V3{P.G[0], P.G[1], P.G[2]}, x + xo_pose(P, i), x + xo_sb(P, i), x + xo_pose(P, j), x + xo_sb(P, j), o);
        if (t == IMU_NBLOCKS) imu_resid(o, rr);
        else {
            int r0, c0; M3 m; double sc;
            imu_block(o, c, t, r0, c0, m, sc);
            put33(Jraw, 30, r0, c0, m, sc);
            if (t == 16) put33(Jraw, 30, 12, 27, m, sc);
        }
    }
    __syncthreads();
    const double* U = P.imu_U + (size_t)f * 225;
    const bool fr = P.marg != 0;             // marginalisation: every block free
    const bool ci = !fr && P.pose_const && P.pose_const[i], cj = !fr && P.pose_const && P.pose_const[j];
    const bool si = !fr && P.sb_const && P.sb_const[i], sj = !fr && P.sb_const && P.sb_const[j];
    for (int e = t; e < 465; e += blockDim.x) {
        if (e < 450) {
            const int row = e / 30, col = e % 30;
            const bool cst = col < 6 ? ci : (col < 15 ? si : (col < 21 ? cj : sj));
            double s = 0;
            if (!cst) for (int k = row; k < 15; ++k) s += U[row * 15 + k] * Jraw[k * 30 + col];
            UJ[e] = s;
        } else {
            const int row = e - 450;
            double s = 0;
            for (int k = row; k < 15; ++k) s += U[row * 15 + k] * rr[k];
            Ur[row] = s;
        }
    }
    __syncthreads();
    for (int e = t; e < 931; e += blockDim.x) {
        double s = 0;
        if (e < 900) { const int a = e / 30, b = e % 30; for (int k = 0; k < 15; ++k) s += UJ[k * 30 + a] * UJ[k * 30 + b]; }
        else if (e < 930) { const int a = e - 900; for (int k = 0; k < 15; ++k) s += UJ[k * 30 + a] * Ur[k]; }
        else { for (int k = 0; k < 15; ++k) s += Ur[k] * Ur[k]; s *= 0.5; }
        out[e] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// LDS per factor: [Ji 12 | Jj 12 | Jex 12 | Jt 2 | Jl 2 | r 2 | eO 6] = 48 doubles
#define VF_STRIDE 49   // odd stride: conflict-free column access
// Windows up to K = 12 (NV <= 80, chunks of <= VIS_MF factors): the block outer products run on the fp64 matrix cores from dense operand rows in LDS
#define VIS_MF 64      // factors a chunk may hold on that path (two operand rows each); chunks are closed at VIL_VCHUNK_FBAL unless the window has more of them than workgroups
#define VIS_LM 16      // landmarks a chunk may hold (one operand row each: one MFMA batch)
#define VIS_RS 80      // row stride of the operand rows: five 16-column tiles, = 16 mod 32 (the four rows of an MFMA operand fragment on different banks)
#define VIS_T_SLOTS 2  // 16 x 16 tiles per wave: 15 upper tiles of a 5 x 5 grid on 8 waves
__host__ __device__ inline bool vis_mfma(int NV) { return NV <= VIS_RS; }
__host__ __device__ inline int vis_ntile(int NV) { const int T = (NV + 15) >> 4; return (T * (T + 1)) >> 1; }
// The candidate inverse depth is formed HERE: lambda_cand = lambda_cur + cg la + cn lb (la, lb: the step directions the step
// kernel's landmark pass left, cg / cn: the dogleg coefficients in Ctl; first sweep and re-sweeps: cg = cn = 0), and written
// into the candidate state by the landmark's lane group.
template <bool MF>      // MF: the block outer products on the matrix cores (windows up to K = 12); a kernel of its own per value -- both paths in one kernel spill
__device__ __forceinline__ void sweep_visual(const DevP& P, const SolveOpts& O, const Ctl& ctl, int wg, const double* x, SysBuf& sb, double* sm) {
    const int NV = P.NV, NVT = P.NVT;
    const int t = threadIdx.x;
    constexpr bool mf = MF;                            // (DevP::vis_mf, a property of the window: the host launches k_sweep<vis_mf>)
    const int VT = (NV + 15) >> 4, nvtile = vis_ntile(NV);
    double* tri_ = sm; double* const tri = tri_;       // NVT packed upper triangle of the visual sub-space -- or, mf: its upper 16 x 16 tiles (I <= J), nvtile x 256
    double* vbc = tri + (mf ? nvtile * 256 : NVT);     // NV
    double* vgr = vbc + NV;                            // NV
    double* vdg = vgr + NV;                            // NV
    double* Jf = vdg + NV;                             // VIL_VCHUNK_F (mf: VIS_MF) x VF_STRIDE
    double* lmr = Jf + (mf ? VIS_MF : VIL_VCHUNK_F) * VF_STRIDE;       // VIS_LM x 16: invp, eA[13]
    double* red = lmr + VIS_LM * 16;
    double* Gm_ = red + 8; double* const Gm = Gm_;                              // mf: 2 VIS_MF x VIS_RS rows of Jc (two per factor) | 16 x VIS_RS rows of e_l | 16 scales -invp_l
    double* Em_ = Gm_ + 2 * VIS_MF * VIS_RS; double* const Em = Em_;
    double* sa_ = Em_ + 16 * VIS_RS; double* const sa = sa_;
    int* fj = (int*)(mf ? sa + 16 : red + 8);          // VIL_VCHUNK_F observer frames
    int* lms = fj + VIL_VCHUNK_F;                      // VIS_LM + 1 chunk-local factor offsets
    int* lanc = lms + VIS_LM + 1;                      // VIS_LM anchor frames
    int* fl = lanc + VIS_LM;                           // VIL_VCHUNK_F factor -> chunk-local landmark
    int* fa = fl + VIL_VCHUNK_F;                       // (mf) VIS_MF anchor frames
#ifdef VIL_STAMPS
    long long vacc[3] = {0, 0, 0}, vprev = 0;
    #define VSTAMP(k) do { __syncthreads(); if (t == 0 && wg == 0) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); P.dbg[32 + k] = tt_; if (k >= 2 && k <= 3) vacc[k - 2] += tt_ - vprev; vprev = tt_; } } while (0)
    #define VSTAMP_SC() do { __syncthreads(); if (t == 0 && wg == 0) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); vacc[2] += tt_ - vprev; vprev = tt_; } } while (0)
#else
    #define VSTAMP(k) do {} while (0)
    #define VSTAMP_SC() do {} while (0)
#endif
    VSTAMP(0);
#ifdef VIL_STAMPS
    long long vt0 = 0; if (t == 0) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(vt0) :: "memory");
#endif
    const double* xcur = P.x[ctl.cur];
    double* xcand = P.x[1 - ctl.cur];
    const double cg = ctl.cg, cn = ctl.cn;
    // first sweep of a solve and re-sweeps: cg = cn = 0 and la / lb still hold the PREVIOUS solve's directions -- possibly inf / NaN after a
    // diverged solve, and 0 * inf is NaN: the terms are dropped, not multiplied by zero (same bits whenever la, lb are finite)
    const bool stepped = cg != 0.0 || cn != 0.0;
    if (mf) { for (int e = t; e < 3 * NV; e += blockDim.x) vbc[e] = 0.0; }      // (the tiles are carried from chunk to chunk only if there is more than one)
    else for (int e = t; e < NVT + 3 * NV; e += blockDim.x) tri[e] = 0.0;
    if (mf) {                                          // operand rows: two per factor of the largest chunk, the landmark rows and their scales
        for (int e = t; e < ((2 * P.vis_fmax + 15) & ~15) * VIS_RS; e += blockDim.x) Gm[e] = 0.0;      // (whole batches of 16 rows are read)
        for (int e = t; e < 16 * VIS_RS + 16; e += blockDim.x) Em[e] = 0.0;
    }
    const bool mfree = P.marg != 0;            // marginalisation of the resident window: every block free, factors masked
    const bool exc = !mfree && P.ex_const != 0, tdc = mfree ? !P.use_td : !P.td_free;
    double cost = 0.0;
    const int4 wg0 = ((const int4*)P.vwg)[2 * wg], wg1 = ((const int4*)P.vwg)[2 * wg + 1];      // {first chunk, end, -, -}, the first chunk's {l0, l1, f0, f1}
    const int sc0 = wg0.x, sc1 = wg0.y;
    for (int chunk = sc0; chunk < sc1; ++chunk) {
        int4 ch = wg1;
        if (chunk != sc0) ch = ((const int4*)P.vchunk)[chunk];
        const int l0 = ch.x, l1 = ch.y, f0 = ch.z, f1 = ch.w;
        const int nf = f1 - f0, nl = l1 - l0;
        __syncthreads();
        VSTAMP(1);
        if (t >= 256 && t <= 256 + nl) { const int q = t - 256; lms[q] = P.lm_start[l0 + q] - f0; if (q < nl) lanc[q] = P.vis_i[P.lm_start[l0 + q]]; }
        if (t < nf) {
            const int f = f0 + t;
            double c[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) c[k] = P.vis_c[(size_t)k * P.vis_stride + f];
            const int i = P.vis_i[f], j = P.vis_j[f], l = P.vis_l[f];
            const double* pi = x + xo_pose(P, i); const double* pj = x + xo_pose(P, j); const double* ex = x + xo_ex(P);
            VisJ o;
            const double lam = stepped ? xcur[xo_lam(P) + l] + cg * P.la[l] + cn * P.lb[l] : xcur[xo_lam(P) + l];
            if (O.precision)
                visual_eval_f32(c, quatR(pi + 3), V3{pi[0], pi[1], pi[2]}, quatR(pj + 3), V3{pj[0], pj[1], pj[2]}, quatR(ex + 3), V3{ex[0], ex[1], ex[2]},
                                lam, x[xo_td(P)], P.sqrt_info, P.k_tr, P.use_td, o);
            else
                visual_eval(c, quatR(pi + 3), V3{pi[0], pi[1], pi[2]}, quatR(pj + 3), V3{pj[0], pj[1], pj[2]}, quatR(ex + 3), V3{ex[0], ex[1], ex[2]},
                            lam, x[xo_td(P)], P.sqrt_info, P.k_tr, P.use_td, o);
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
            fj[t] = j; fl[t] = l - l0; if (mf) fa[t] = i;
            // observer-pose pieces that need no landmark-level sum
            for (int k = 0; k < 6; ++k) {
                const double j0 = cj ? 0.0 : sr * o.Jj[k], j1 = cj ? 0.0 : sr * o.Jj[6 + k];
                const double eo = j0 * w[38] + j1 * w[39];
                w[42 + k] = eo;
                sb.eO[(size_t)(P.vis_f0 + f) * 6 + k] = eo;
                if (!cj) {
                    const double g = j0 * w[40] + j1 * w[41];
                    lds_add(vbc + col_pose(P, j) + k, g);
                    lds_add(vgr + col_pose(P, j) + k, g);
                    lds_add(vdg + col_pose(P, j) + k, j0 * j0 + j1 * j1);
                }
            }
        }
        __syncthreads();
        VSTAMP(2);
        // ---- per landmark: pivots, e on the shared groups, gradients; 16 lanes per landmark ---------------
        // lane component k: 0..5 anchor pose, 6..11 extrinsic, 12 td, 13 -> (h, b) pivot pieces
        if (t < 16 * nl) {
            const int tl = t >> 4, k = t & 15;
            const int l = l0 + tl;
            const int fs = lms[tl], fe = lms[tl + 1];
            const int a = lanc[tl];
            double e = 0, g = 0, dg = 0, h = 0, b = 0;
            const int off = k < 6 ? k : (k < 12 ? 24 + (k - 6) : 36);
            const int rs = k < 12 ? 6 : 1;      // row stride inside the 2 x n block
            for (int q = fs; q < fe; ++q) {
                const double* w = Jf + q * VF_STRIDE;
                const double l0_ = w[38], l1_ = w[39], r0 = w[40], r1 = w[41];
                if (k < 13) { const double j0 = w[off], j1 = w[off + rs]; e += j0 * l0_ + j1 * l1_; g += j0 * r0 + j1 * r1; dg += j0 * j0 + j1 * j1; }
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
            const double ib = invp * b;
            double* lr = lmr + tl * 16;
            // (a landmark without a factor in this workgroup's table -- none at all, or owned by another rank -- leaves the set untouched:
            //  its entries stay zero here and the all-reduce takes them from the owner)
            if (k == 13) { if (fe > fs) { sb.hll[l] = h; sb.bl[l] = b; sb.invp[l] = invp; sb.sl[l] = Sl; } lr[0] = invp; lr[14] = (double)a; if (mf) sa[tl] = -invp; }
            if (k == 14 && fe > fs) xcand[xo_lam(P) + l] = stepped ? xcur[xo_lam(P) + l] + cg * P.la[l] + cn * P.lb[l] : xcur[xo_lam(P) + l];      // the same expression the factor threads evaluated
            if (k < 13) {
                lr[1 + k] = e; if (fe > fs) sb.eA[(size_t)l * 13 + k] = e;
                const int col = k < 6 ? col_pose(P, a) + k : (k < 12 ? col_ex(P) + k - 6 : col_td(P));
                if (mf) Em[tl * VIS_RS + col] = e;
                if (g != 0.0 || dg != 0.0) { lds_add(vbc + col, g); lds_add(vgr + col, g - ib * e); lds_add(vdg + col, dg); }
            }
            // observer columns: gred -= invp b eO ; lanes 0..5 of the group walk the factors
            if (k < 6) for (int q = fs; q < fe; ++q) { const double eo = Jf[q * VF_STRIDE + 42 + k]; if (mf) Em[tl * VIS_RS + col_pose(P, fj[q]) + k] = eo; if (eo != 0.0) lds_add(vgr + col_pose(P, fj[q]) + k, -ib * eo); }
        } else if (mf) {
            // the threads the landmarks do not use spread the staged Jacobian blocks into the dense operand rows (two per factor, zeroed before): item =
            // (factor, residual row, one of the 19 columns); written by the evaluating thread itself the six row / group addresses cost it registers it
            // does not have (19 VGPRs spilled, 3 MB of scratch traffic per launch)
            const int t0 = 16 * nl, nt = blockDim.x - t0;
            double* const Gm = opaque(Gm_);
            int tq = t; asm volatile("" : "+v"(tq));
            for (int it = tq - t0; it < nf * 38; it += nt) {
                const int q = it / 38, e = it - 38 * q, rr = e >= 19 ? 1 : 0, m = e - 19 * rr;
                const double* w = Jf + q * VF_STRIDE;
                int col; double v;
                if (m < 6) { col = col_pose(P, fa[q]) + m; v = w[rr * 6 + m]; }
                else if (m < 12) { col = col_pose(P, fj[q]) + (m - 6); v = w[12 + rr * 6 + (m - 6)]; }
                else if (m < 18) { col = col_ex(P) + (m - 12); v = w[24 + rr * 6 + (m - 12)]; }
                else { col = col_td(P); v = w[36 + rr]; }
                Gm[(2 * q + rr) * VIS_RS + col] = v;
            }
        }
        __syncthreads();
        VSTAMP(3);
        // ---- tri += sum_f Jc^T Jc - invp e e^T.  Three kinds of work items with (nearly) uniform trip counts inside a wave
        //      (divergent loops cost the maximum over the lanes); the kinds only read the staged factors and add into tri with LDS
        //      atomics, so they run side by side: every kind's item range is padded to whole waves and the waves of the workgroup
        //      walk the concatenation -- two or three rounds of latency instead of one or two per kind with barriers in between.
        //      Groups: A = anchor pose, X = extrinsic, T = td (shared by all factors of the landmark), O_f = observing pose of factor f.
        // Windows up to K = 12 (mf): the same update as G_A^T G_B on the fp64 matrix cores.  Jc_f (2 x NV) is non-zero on the anchor pose, the observing
        // pose, the extrinsic and td, e_l on the anchor, the extrinsic, td and the landmark's observers: the factor threads and the landmark lanes above
        // left them as dense rows (Gm: two per factor, Em: one per landmark), and sum_f Jc^T Jc - sum_l invp_l e_l e_l^T is one v_mfma_f64_16x16x4 per
        // 16 x 16 tile and four rows, -invp_l applied to the A operand on its way in.  Every wave owns two of the 15 upper tiles; all operand loads of
        // a batch are in flight before its first MFMA.  Deterministic, and 1.6 k cycles per chunk where the ~1200 atomic work items below take 5 k.
        if constexpr (mf) {
            int tq = t; asm volatile("" : "+v"(tq));      // (opaque: what is derived from it is computed here, not at the top of the role)
            const int wave = __builtin_amdgcn_readfirstlane(tq >> 6), lane = tq & 63;
            const bool first = chunk == sc0, last = chunk + 1 == sc1, has1 = wave + 8 < nvtile;
            double* const Gm = opaque(Gm_); double* const Em = opaque(Em_); double* const sa = opaque(sa_); double* const tri = opaque(tri_);
            int tI[VIS_T_SLOTS], tJ[VIS_T_SLOTS]; d4 acc[VIS_T_SLOTS];
#pragma unroll
            for (int u = 0; u < VIS_T_SLOTS; ++u) {
                const int g = min(wave + 8 * u, nvtile - 1);                    // (a wave without a second tile mirrors the last one and stores nothing)
                int I = 0, rem = g; while (rem >= VT - I) { rem -= VT - I; ++I; }      // upper tiles row by row: (I, I + rem)
                tI[u] = I; tJ[u] = I + rem;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[u][q] = first ? 0.0 : tri[g * 256 + ((lane >> 4) + 4 * q) * 16 + (lane & 15)];
            }
            auto mma = [&](const double* G, auto nk_c, auto scaled_c) {       // acc += G_A^T G over NK * 4 rows; G_A = G, or the rows of G scaled by sa[row]
                constexpr int NK = decltype(nk_c)::value; constexpr bool SCALED = decltype(scaled_c)::value;
                const double* p = G + (lane >> 4) * VIS_RS + (lane & 15);
                double a0[NK], b0[NK], a1[NK], b1[NK];
#pragma unroll
                for (int ks = 0; ks < NK; ++ks) {
                    a0[ks] = p[ks * 4 * VIS_RS + (tI[0] << 4)]; b0[ks] = p[ks * 4 * VIS_RS + (tJ[0] << 4)];
                    a1[ks] = p[ks * 4 * VIS_RS + (tI[1] << 4)]; b1[ks] = p[ks * 4 * VIS_RS + (tJ[1] << 4)];
                }
                if (has1) {
#pragma unroll
                    for (int ks = 0; ks < NK; ++ks) {
                        const double sc = SCALED ? sa[4 * ks + (lane >> 4)] : 1.0;
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(SCALED ? sc * a0[ks] : a0[ks], b0[ks], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(SCALED ? sc * a1[ks] : a1[ks], b1[ks], acc[1], 0, 0, 0);
                    }
                } else {                                                       // (one tile: the matrix core of this SIMD is shared with another wave)
#pragma unroll
                    for (int ks = 0; ks < NK; ++ks) {
                        const double sc = SCALED ? sa[4 * ks + (lane >> 4)] : 1.0;
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(SCALED ? sc * a0[ks] : a0[ks], b0[ks], acc[0], 0, 0, 0);
                    }
                }
            };
            mma(Gm, std::integral_constant<int, 4>{}, std::false_type{});
            if (nf > 8) mma(Gm + 16 * VIS_RS, std::integral_constant<int, 4>{}, std::false_type{});
            if (nf > 16) mma(Gm + 32 * VIS_RS, std::integral_constant<int, 4>{}, std::false_type{});
            if (nf > 24) mma(Gm + 48 * VIS_RS, std::integral_constant<int, 4>{}, std::false_type{});
            if (nf > 32) {                                                     // (wide chunks: windows with more chunks of 32 factors than workgroups)
                mma(Gm + 64 * VIS_RS, std::integral_constant<int, 4>{}, std::false_type{});
                if (nf > 40) mma(Gm + 80 * VIS_RS, std::integral_constant<int, 4>{}, std::false_type{});
                if (nf > 48) mma(Gm + 96 * VIS_RS, std::integral_constant<int, 4>{}, std::false_type{});
                if (nf > 56) mma(Gm + 112 * VIS_RS, std::integral_constant<int, 4>{}, std::false_type{});
            }
            mma(Em, std::integral_constant<int, 4>{}, std::true_type{});
            if (last) {
                // the accumulators go into the record's layout -- the packed upper triangle -- in LDS (over the operand rows, which every wave is done
                // with) and leave with whole-line stores below: written straight from the registers, 16 lanes x 8 bytes per row fragment straddle the
                // unaligned rows of the triangle and the partial lines doubled the kernel's write traffic (5.8 MB against 2.6)
                __syncthreads();
#pragma unroll
                for (int u = 0; u < VIS_T_SLOTS; ++u) if (wave + 8 * u < nvtile) {
                    const int j = (tJ[u] << 4) + (lane & 15);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int i = (tI[u] << 4) + (lane >> 4) + 4 * q; if (i <= j && j < NV) Gm[tri_idx(NV, i, j)] = acc[u][q]; }
                }
            } else {
#pragma unroll
                for (int u = 0; u < VIS_T_SLOTS; ++u) if (wave + 8 * u < nvtile) {
                    const int g = wave + 8 * u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) tri[g * 256 + ((lane >> 4) + 4 * q) * 16 + (lane & 15)] = acc[u][q];
                }
                __syncthreads();                                               // every wave is done with the operand rows: zero them for the next chunk
                for (int e = t; e < 2 * nf * VIS_RS; e += blockDim.x) Gm[e] = 0.0;
                for (int e = t; e < 16 * VIS_RS; e += blockDim.x) Em[e] = 0.0;
            }
        } else {
            // (a) shared x shared blocks: item = (landmark, pair of {A,X,T}, row); inner loop over the landmark's factors
            auto item_a = [&](int it) {
                const int tl = it / 36, pr = it - 36 * tl, p = pr / 6, r = pr - 6 * p;
                const int g1 = p < 3 ? 0 : (p < 5 ? 1 : 2), g2 = p < 3 ? p : (p < 5 ? p - 2 : 2);
                const int n1 = g1 == 2 ? 1 : 6, n2 = g2 == 2 ? 1 : 6;
                if (r >= n1) return;
                const int fs = lms[tl], fe = lms[tl + 1];
                const double* lr = lmr + tl * 16;
                const int a = lanc[tl];
                const int o1 = g1 == 0 ? 0 : (g1 == 1 ? 24 : 36), o2 = g2 == 0 ? 0 : (g2 == 1 ? 24 : 36);
                const int c1 = g1 == 0 ? col_pose(P, a) : (g1 == 1 ? col_ex(P) : col_td(P)), c2 = g2 == 0 ? col_pose(P, a) : (g2 == 1 ? col_ex(P) : col_td(P));
                const double* e1 = lr + 1 + (g1 == 0 ? 0 : (g1 == 1 ? 6 : 12)); const double* e2 = lr + 1 + (g2 == 0 ? 0 : (g2 == 1 ? 6 : 12));
                const int s1 = n1 == 1 ? 1 : 6, s2 = n2 == 1 ? 1 : 6;
                const double ie1 = lr[0] * e1[r];
                double acc[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) acc[c] = (c < n2) ? -ie1 * e2[c] : 0.0;
                for (int q = fs; q < fe; ++q) {
                    const double* w = Jf + q * VF_STRIDE;
                    const double w0 = w[o1 + r], w1 = w[o1 + s1 + r];
#pragma unroll
                    for (int c = 0; c < 6; ++c) if (c < n2) acc[c] += w0 * w[o2 + c] + w1 * w[o2 + s2 + c];
                }
#pragma unroll
                for (int c = 0; c < 6; ++c) if (c < n2 && (g1 != g2 || c >= r) && acc[c] != 0.0) lds_add(tri + tri_idx(NV, c1 + r, c2 + c), acc[c]);
            };
            // (b) shared x observer blocks: item = (factor, shared group, row); exactly one factor contributes
            auto item_b = [&](int it) {
                const int q = it / 18, rem = it - 18 * q, g1 = rem / 6, r = rem - 6 * g1;
                if (g1 == 2 && r > 0) return;
                const int tl = fl[q];
                const double* lr = lmr + tl * 16;
                const double* w = Jf + q * VF_STRIDE;
                const int o1 = g1 == 0 ? 0 : (g1 == 1 ? 24 : 36), s1 = g1 == 2 ? 1 : 6;
                const int c1 = g1 == 0 ? col_pose(P, lanc[tl]) : (g1 == 1 ? col_ex(P) : col_td(P)), c2 = col_pose(P, fj[q]);
                const double ie1 = lr[0] * lr[1 + (g1 == 0 ? 0 : (g1 == 1 ? 6 : 12)) + r];
                const double w0 = w[o1 + r], w1 = w[o1 + s1 + r];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const double v = w0 * w[12 + c] + w1 * w[18 + c] - ie1 * w[42 + c];
                    if (v != 0.0) lds_add(tri + tri_idx(NV, c1 + r, c2 + c), v);
                }
            };
            // (c) observer x observer blocks: item = (factor q, row); walks the later factors of the same landmark
            auto item_c = [&](int it) {
                const int q = it / 6, r = it - 6 * q;
                const int tl = fl[q], fe = lms[tl + 1];
                const double* w = Jf + q * VF_STRIDE;
                const double ieq = lmr[tl * 16] * w[42 + r];
                const int c1 = col_pose(P, fj[q]) + r;
                const double j0 = w[12 + r], j1 = w[18 + r];
#pragma unroll
                for (int c = 0; c < 6; ++c) if (c >= r) {          // diagonal block (q, q): upper part
                    const double v = j0 * w[12 + c] + j1 * w[18 + c] - ieq * w[42 + c];
                    if (v != 0.0) lds_add(tri + tri_idx(NV, c1, c1 - r + c), v);
                }
                for (int f2 = q + 1; f2 < fe; ++f2) {
                    const double* w2 = Jf + f2 * VF_STRIDE;
                    const int c2 = col_pose(P, fj[f2]);
#pragma unroll
                    for (int c = 0; c < 6; ++c) { const double v = -ieq * w2[42 + c]; if (v != 0.0) lds_add(tri + tri_idx(NV, c1, c2 + c), v); }
                }
            };
            const int na = nl * 36, nb_ = nf * 18, nc_ = nf * 6;
            const int pa = (na + 63) & ~63, pc = (nc_ + 63) & ~63, pb = (nb_ + 63) & ~63;      // the kinds with inner loops first
            for (int it = t; it < pa + pc + pb; it += blockDim.x) {
                if (it < pa) { if (it < na) item_a(it); }
                else if (it < pa + pc) { const int i2 = it - pa; if (i2 < nc_) item_c(i2); }
                else { const int i3 = it - pa - pc; if (i3 < nb_) item_b(i3); }
            }
        }
        VSTAMP_SC();
    }
    VSTAMP(4);
#ifdef VIL_STAMPS
    if (t == 0 && wg == 0) { P.dbg[62] = vacc[0]; P.dbg[63] = vacc[1]; P.dbg[47] = vacc[2]; }
#endif
    cost = block_sum(cost, red);
    double* out = P.vpart + (size_t)wg * P.VP;
    if (mf) {                                              // (the last chunk left the packed triangle in Gm; block_sum's barriers are behind it)
        for (int e = t; e < NVT; e += blockDim.x) out[e] = Gm[e];
        for (int e = t; e < 3 * NV; e += blockDim.x) out[NVT + e] = vbc[e];
    } else
    for (int e = t; e < NVT + 3 * NV; e += blockDim.x) out[e] = tri[e];
    if (t == 0) out[NVT + 3 * NV] = cost;
    VSTAMP(5);
#ifdef VIL_STAMPS
    if (t == 0) { long long vt1; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(vt1) :: "memory"); atomicMax((unsigned long long*)(P.dbg + 60), (unsigned long long)(vt1 - vt0)); atomicAdd((unsigned long long*)(P.dbg + 61), (unsigned long long)(vt1 - vt0)); }
#endif
}

// ---------------------------------------------------------------------------------------------
template <int NR>
__device__ __forceinline__ void sweep_lidar(const DevP& P, const SolveOpts& O, int wgc, const double* x, double* sm) {
    const int per = blockDim.x >> 8;                    // 256-point chunks per workgroup
    const int sub = threadIdx.x >> 8;
    const int nchunk = NR == 1 ? P.n_pchunk : P.n_echunk;
    const int chunk = wgc * per + sub;
    const bool have = chunk < nchunk;
    const int* ch = (NR == 1 ? P.pchunk : P.echunk) + 3 * (have ? chunk : 0);
    const int start = ch[0], cnt = have ? ch[1] : 0, k = ch[2];
    const int t = threadIdx.x & 255;
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
    if (have && t < 28) P.lpart[(size_t)gchunk * 28 + t] = sm[t] + sm[28 + t] + sm[56 + t] + sm[84 + t];
}

// ---------------------------------------------------------------------------------------------
__device__ inline const double* prior_block_ptr(const DevP& P, const double* x, int b) {
    const int kind = P.pblk_kind[b], idx = P.pblk_index[b];
    return kind == 0 ? x + xo_pose(P, idx) : (kind == 1 ? x + xo_sb(P, idx) : (kind == 2 ? x + xo_ex(P) : x + xo_td(P)));
}

// [prior] workgroup: r = r0 + J0 dx ;  J0^T J0 = pH, J0^T r0 = pg0, r0^T r0 = pc0 are pre-contracted, so the prior costs
// one n x n gemv.  8 threads per output column split the k range (loads in flight instead of one dependent chain of n
// global loads per thread), folded with three xor-shuffles.
__device__ __forceinline__ void sweep_prior(const DevP& P, const double* x, double* sm) {
    const int t = threadIdx.x;
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
    double part = 0;
    for (int i0 = 0; i0 < n; i0 += (int)(blockDim.x >> 3)) {
        const int i = i0 + (t >> 3), r = t & 7;
        double s = 0;
        if (i < n) for (int k = r; k < n; k += 8) s += P.pH[(size_t)k * n + i] * dx[k];
        s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
        if (i < n && r == 0) {
            const double g = P.pg0[i] + s;
            part += dx[i] * (P.pg0[i] + g);
            P.mpart[i] = g;
        }
    }
    part = block_sum(part, red);
    if (t == 0) P.mpart[n] = 0.5 * (P.pc0[0] + part);
}

// [rel] workgroup: the scan-to-scan ICP and LPS AutoDiff factors
__device__ __forceinline__ void sweep_misc(const DevP& P, const SolveOpts& O, const double* x, double* sm) {
    const int t = threadIdx.x;
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
        out0[e] = v;
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
//   entry splitting the sum over the visual partials; fixed summation order across workgroups (the partials themselves are
//   accumulated with LDS atomics inside a visual workgroup, so two runs agree to rounding, not bit for bit).
//   The last workgroup handles the vectors and the cost.
#define RED_EPW 32
// AG: the results are stored at agent scope (gather + step in one launch: the master reads them in the same launch)
// EPW: entries per workgroup = blockDim / 8 (32 with 256 threads: the gather kernel; 64 with 512: the gather workgroups of the merged launch,
// whose register allocation admits one workgroup per compute unit whatever its thread count)
template <bool AG = false, int EPW = RED_EPW>
__device__ __forceinline__ void reduce_gather(const DevP& P, const Ctl& ctl, const int blk /* gather workgroup index */) {
    using namespace vd;
    auto put = [](double* p, double v) { if (AG) st_ag(p, v); else *p = v; };
    if ((int)threadIdx.x >= 8 * EPW) return;
    const int cand = 1 - ctl.cur;
    SysBuf sb = P.sys[cand];
    const int D = P.D, NV = P.NV, K = P.K, t = threadIdx.x;
    const int n_rel = P.n_icp + P.n_lps;
    const double* rel0 = P.mpart + (P.pn > 0 ? P.pn + 1 : 0);
    const int NL = (D * (D + 1)) >> 1;
    const int nSblk = (NL + EPW - 1) / EPW;
    __shared__ double part[2][8][EPW];
    __shared__ int tab[64 + 48 + 2 * 66];      // imu (i, j) pairs | ICP/LPS pose ids (4 per factor) | LiDAR chunk ranges per pose
    int* t_imu = tab; int* t_rel = tab + 64; int* t_lch = tab + 112;
    if (blk < nSblk) {
        if (P.skip_mask & 32) return;
        // stage the small index tables once per workgroup
        if (t < 2 * P.n_imu && t < 64) t_imu[t] = (t & 1) ? P.imu_j[t >> 1] : P.imu_i[t >> 1];
        if (t >= 64 && t < 64 + 4 * n_rel) { const int q = t - 64, f = q >> 2, b = q & 3; t_rel[q] = f < P.n_icp ? P.icp_ids[4 * f + b] : (b < 2 ? P.lps_ids[2 * (f - P.n_icp) + b] : -1); }
        if (t >= 128 && t < 128 + 2 * (K + 1) && t < 128 + 132) t_lch[t - 128] = P.lchunk_pose[t - 128];
        __syncthreads();
        const int el = t & (EPW - 1), slice = t / EPW;
        const int idx = blk * EPW + el;
        int i = 0, j = 0;
        const bool ok = idx < NL;
        if (ok) {
            i = (int)((sqrtf(8.f * (float)idx + 1.f) - 1.f) * 0.5f);      // (the two loops below make it exact; the fp64 square root is a 1 k-cycle chain)
            while (((i + 1) * (i + 2)) / 2 <= idx) ++i;
            while ((i * (i + 1)) / 2 > idx) --i;
            j = idx - (i * (i + 1)) / 2;              // (i, j), j <= i, enumerates a lower triangle row by row ...
            const int a = i; i = D - 1 - a; j = D - 1 - j;   // ... mirrored to the upper entry (D-1-a, D-1-b): consecutive lanes -> consecutive columns (coalesced partial reads)
        }
        double vs = 0.0, vdg = 0.0;
        if (ok && j < NV) {
            const int tix = tri_idx(NV, i, j);
            // eight records per round, every load issued before the first add (clamped record index + select: no predicated loads)
            const double* vp = P.vpart + tix;
            const double* vd_ = P.vpart + P.NVT + 2 * NV + i;
            const int nw = P.n_vwg, wlast = nw - 1;
            for (int w = slice; w < nw; w += 64) {
                double a[8], d[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int wc = min(w + 8 * u, wlast); a[u] = vp[(size_t)wc * P.VP]; d[u] = (i == j) ? vd_[(size_t)wc * P.VP] : 0.0; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const bool in = w + 8 * u < nw; vs += in ? a[u] : 0.0; vdg += in ? d[u] : 0.0; }
            }
        }
        // the non-visual contributions are spread over the 8 slices of an entry
        double ms = 0.0;
        if (ok) {
            if (j < 6 * K && i / 6 == j / 6) {                 // LiDAR plane and edge points: pose-diagonal blocks.  The chunk records of a pose
                const int k = i / 6, a = i - 6 * k, b = j - 6 * k;   // (one per 256 points: 9 per pose at 24 k points, 37 at 96 k) are dealt to the eight slices
                const int li = a * 6 - ((a * (a - 1)) >> 1) + (b - a);   // of the entry -- one slice walking them all was the longest chain of this kernel
                for (int c = t_lch[k] + slice; c < t_lch[k + 1]; c += 8) ms += P.lpart[(size_t)c * 28 + li];
                for (int c = t_lch[K + 1 + k] + slice; c < t_lch[K + 2 + k]; c += 8) ms += P.lpart[(size_t)(P.n_pchunk + c) * 28 + li];
            }
            if (slice == 3 && j < 6 * K) {                     // ICP / LPS blocks live on pose columns
                const int pi = i / 6, pj = j / 6, ri = i - 6 * pi, rj = j - 6 * pj;
                for (int f = 0; f < n_rel; ++f)
                    for (int ba = 0; ba < 4; ++ba) if (t_rel[4 * f + ba] == pi) for (int bb = 0; bb < 4; ++bb) if (t_rel[4 * f + bb] == pj)
                        ms += rel0[(size_t)f * 601 + (ba * 6 + ri) * 24 + bb * 6 + rj];
            }
            if (slice == 4 || slice == 5) {                    // IMU blocks, two slices split the factors
                for (int f = slice - 4; f < P.n_imu; f += 2) {
                    const int la = imu_local(P, t_imu[2 * f], t_imu[2 * f + 1], i);
                    if (la < 0) continue;
                    const int lb = imu_local(P, t_imu[2 * f], t_imu[2 * f + 1], j);
                    if (lb >= 0) ms += P.ipart[(size_t)f * 931 + la * 30 + lb];
                }
            }
            if (slice == 6 && P.pn > 0) { const int pi = P.pinv[i], pj = P.pinv[j]; if (pi >= 0 && pj >= 0) ms += P.pH[(size_t)pi * P.pn + pj]; }
        }
        part[0][slice][el] = vs + ms; part[1][slice][el] = vdg - vs;     // [1]: un-reduced minus reduced visual diagonal
        __syncthreads();
        if (slice != 0 || !ok) return;
        double s = 0.0, dd = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { s += part[0][q][el]; dd += part[1][q][el]; }
        put(sb.S + (size_t)i * D + j, s);
        put(sb.S + (size_t)j * D + i, s);
        if (i == j) put(sb.diag + i, s + (i < NV ? dd : 0.0));   // un-reduced diagonal: add back the Schur term of the visual part
        return;
    }
    // ---- gradient vectors bc / gred: 2D entries, same 8-slice scheme -------------------------------------------------
    const int nVblk = (2 * D + EPW - 1) / EPW;
    if (blk < nSblk + nVblk) {
        if (P.skip_mask & 64) return;
        if (t < 2 * P.n_imu && t < 64) t_imu[t] = (t & 1) ? P.imu_j[t >> 1] : P.imu_i[t >> 1];
        if (t >= 64 && t < 64 + 4 * n_rel) { const int q = t - 64, f = q >> 2, b = q & 3; t_rel[q] = f < P.n_icp ? P.icp_ids[4 * f + b] : (b < 2 ? P.lps_ids[2 * (f - P.n_icp) + b] : -1); }
        if (t >= 128 && t < 128 + 2 * (K + 1) && t < 128 + 132) t_lch[t - 128] = P.lchunk_pose[t - 128];
        __syncthreads();
        const int el = t & (EPW - 1), slice = t / EPW;
        const int v = (blk - nSblk) * EPW + el;
        const bool ok = v < 2 * D;
        const int which = v >= D ? 1 : 0, i = which ? v - D : v;       // 0: bc, 1: gred
        double acc = 0.0;
        if (ok) {
            if (i < NV) for (int w = slice; w < P.n_vwg; w += 8) acc += P.vpart[(size_t)w * P.VP + P.NVT + which * NV + i];
            if (i < 6 * K) {
                const int k = i / 6, a = i - 6 * k;
                for (int c = t_lch[k] + slice; c < t_lch[k + 1]; c += 8) acc += P.lpart[(size_t)c * 28 + 21 + a];      // (chunk records dealt to the eight slices, as above)
                for (int c = t_lch[K + 1 + k] + slice; c < t_lch[K + 2 + k]; c += 8) acc += P.lpart[(size_t)(P.n_pchunk + c) * 28 + 21 + a];
                if (slice == 3) for (int f = 0; f < n_rel; ++f) for (int ba = 0; ba < 4; ++ba) if (t_rel[4 * f + ba] == k) acc += rel0[(size_t)f * 601 + 576 + ba * 6 + a];
            }
            if (slice == 4 || slice == 5) for (int f = slice - 4; f < P.n_imu; f += 2) { const int la = imu_local(P, t_imu[2 * f], t_imu[2 * f + 1], i); if (la >= 0) acc += P.ipart[(size_t)f * 931 + 900 + la]; }
            if (slice == 6 && P.pn > 0) { const int pi = P.pinv[i]; if (pi >= 0) acc += P.mpart[pi]; }
        }
        part[0][slice][el] = acc;
        __syncthreads();
        if (slice != 0 || !ok) return;
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) sum += part[0][q][el];
        put((which ? sb.gred : sb.bc) + i, sum);
        return;
    }
    // ---- cost (one workgroup, tree reduction) ---------------------------------------------------------------------
    __shared__ double red[8];
    if (P.skip_mask & 64) return;
    double c = 0.0;
    for (int w = t; w < P.n_vwg; w += 8 * EPW) c += P.vpart[(size_t)w * P.VP + P.NVT + 3 * NV];
    for (int q = t; q < P.n_pchunk + P.n_echunk; q += 8 * EPW) c += P.lpart[(size_t)q * 28 + 27];
    for (int f = t; f < P.n_imu; f += 8 * EPW) c += P.ipart[(size_t)f * 931 + 930];
    for (int f = t; f < n_rel; f += 8 * EPW) c += rel0[(size_t)f * 601 + 600];
    if (t == 0 && P.pn > 0) c += P.mpart[P.pn];
    c = wave_sum(c);
    if ((t & 63) == 0) red[t >> 6] = c;
    __syncthreads();
    if (t == 0) { double tot = 0.0; for (int w = 0; w < EPW / 8; ++w) tot += red[w]; put(sb.cost, tot); }      // (EPW / 8 waves)
}

// Gather of the sweep's partial records (reduce_gather above) as a launch of its own: vil_linearize / the marginalisation (no step kernel
// behind it), the multi-GPU path (the collective sits between gather and step) and windows too large for the merged launch (K > 12) -- for
// those the grid carries one more workgroup per 16 x 16 tile of W W^T (vil_prechain.hpp; n_gather = the gather workgroups).  A solve of a
// small window on one GPU gathers inside k_step (rs_merged).
__global__ __launch_bounds__(VIL_THREADS) void k_reduce(DevP P, int n_gather) {
    const Ctl ctl = *P.ctl;
    if (ctl.done) return;
    if ((int)blockIdx.x >= n_gather) { vd::prechain_ww_tile(P, (int)blockIdx.x - n_gather); return; }
    reduce_gather(P, ctl, (int)blockIdx.x);
}

// the IMU / prior workgroup `slot` of this launch has written its record (read by the chain workgroup of the same launch, prechain 2).
// Epoch of the flags: solve generation + Ctl::swe, which only the step kernel advances.
__device__ __forceinline__ void sweep_signal(const DevP& P, const Ctl& ctl, int slot) {
    __threadfence();                 // every wave's stores of the record (__syncthreads alone does not wait for global stores)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(P.swflag + slot, (int)((((unsigned)ctl.gen) << 12) + (unsigned)ctl.swe + 1u), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// The sweep: grid = n_imu + 2 (+ 1: prechain 2) + n_vwg + ceil(n_pchunk / 2) + ceil(n_echunk / 2) workgroups of VIL_SWEEP_THREADS threads.
// Workgroup order: [imu x n_imu | prior | rel | (chain) | visual x n_vwg | plane | edge] -- the short roles the chain workgroup waits for
// come first (roles: top of this file)
template <bool MF>
__global__ __launch_bounds__(VIL_SWEEP_THREADS) void k_sweep(DevP P, SolveOpts O) {
    extern __shared__ double sm[];
    const Ctl ctl = *P.ctl;
    if (ctl.done) return;                                // (the result of a finished solve is written out by k_finish, vil_finish.hpp)
    if (blockIdx.x == 0 && threadIdx.x == 0) P.ctl->n_sweeps = ctl.n_sweeps + 1;   // live (not early-exited) launches, for the profiler
    const int cand = 1 - ctl.cur;
    const double* x = P.x[cand];
    SysBuf sb = P.sys[cand];
    int b = blockIdx.x;
    const bool pre = P.prechain == 2 && ctl.lin_mode == 0;
    if (b < P.n_imu) { if (!(P.skip_mask & 2)) vd::sweep_imu(P, O, b, x, sm); if (pre) sweep_signal(P, ctl, b); return; }
    b -= P.n_imu;
    if (b == 0) { if (!(P.skip_mask & 16)) vd::sweep_prior(P, x, sm); if (pre) sweep_signal(P, ctl, P.n_imu); return; }
    if (b == 1) {
        if (!(P.skip_mask & 16)) vd::sweep_misc(P, O, x, sm);
        if (P.world > 1) {                               // factor set sharded over ranks: the visual workgroups of this rank form the candidate inverse depth of
            const double* xcur = P.x[ctl.cur];           // the landmarks it owns; every rank holds la / lb of ALL landmarks (the step kernel runs on the all-reduced
            double* xcand = P.x[1 - ctl.cur];            // system), so the rest is filled in here and the states stay identical on all ranks
            const bool stepped = ctl.cg != 0.0 || ctl.cn != 0.0;      // (as in sweep_visual: stale la / lb are not multiplied by zero)
            for (int l = threadIdx.x; l < P.L; l += blockDim.x) xcand[xo_lam(P) + l] = stepped ? xcur[xo_lam(P) + l] + ctl.cg * P.la[l] + ctl.cn * P.lb[l] : xcur[xo_lam(P) + l];
        }
        return;
    }
    b -= 2;
    if (P.prechain == 2) { if (b == 0) { if (pre) vd::prechain_wg(P, ctl, O.jacobi_scaling, sm, 0, true); return; } b -= 1; }
    // visual workgroups next: the longest-running factor role
    if (b < P.n_vwg) { if (!(P.skip_mask & 1)) vd::sweep_visual<MF>(P, O, ctl, b, x, sb, sm); return; }
    b -= P.n_vwg;
    const int per = VIL_SWEEP_THREADS / 256, npw = (P.n_pchunk + per - 1) / per;
    if (b < npw) { if (!(P.skip_mask & 4)) vd::sweep_lidar<1>(P, O, b, x, sm); return; }
    b -= npw;
    if (!(P.skip_mask & 8)) vd::sweep_lidar<3>(P, O, b, x, sm);
}
