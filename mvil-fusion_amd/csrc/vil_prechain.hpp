// EXPERIMENTAL (off unless VIL_PRECHAIN=1; measured r2 at K = 10: the step kernel gets 8.4 us shorter, the sweep 11.8 us longer --
// the chain workgroup waits 6 us for the IMU records, stages for 7.5 us and factors for 13 us, against 20 us of visual work).
// The speed-bias chain eliminated AHEAD of the step kernel (single GPU): every entry of S' with a row or a column in the chain
// part comes from the IMU factors and the prior alone -- visual, LiDAR, ICP and LPS factors never touch a speed-bias block
// (estimator.cpp:1179-1186, 1189-1242, 1298-1396) -- so the chain of vil_chain.hpp can be factored from the sweep's IMU / prior
// partial records by ONE extra workgroup of k_reduce, concurrently with the workgroups that gather S'.  The step kernel then
// only scales the pose rows of W (Jacobi scaling of the pose part needs the visual / LiDAR diagonal, which is not known here;
// row scaling commutes with the column elimination), subtracts W W^T, and solves the 6K + 8 dense rows.
#pragma once
#include "vil_chain.hpp"

namespace vd {

// the chain's view of the system: raw entries assembled from the partial records.  Loads are unconditional (clamped
// indices, 0/1 masks): per-lane predicated loads compile to exec-mask branches with their own waits.
struct ChainSrcPart {
    const DevP& P; const int* as_i; const int* as_j; const int* pinv;     // small index tables staged in LDS
    __device__ __forceinline__ double imu_entry(int f, int la, int lb) const {
        const double m = f >= 0 ? 1.0 : 0.0;
        return m * P.ipart[(size_t)max(f, 0) * 931 + la * 30 + lb];
    }
    __device__ __forceinline__ double prior_entry(int i, int j) const {
        if (P.pn <= 0) return 0.0;
        const int pi = pinv[i], pj = pinv[j];
        const double m = (pi >= 0 && pj >= 0) ? 1.0 : 0.0;
        return m * P.pH[(size_t)max(pi, 0) * P.pn + max(pj, 0)];
    }
    // raw S'(i, j), j in the chain part
    __device__ __forceinline__ double raw(int i, int j) const {
        const int NP = P.NV, K = P.K;
        const int kb = (j - NP) / 9, cb = (j - NP) - 9 * kb;
        const int fi = as_i[kb], fj = as_j[kb];                   // factor (kb, kb+1) and factor (kb-1, kb)
        double v;
        if (i >= NP) {
            const int ka = (i - NP) / 9, ca = (i - NP) - 9 * ka;
            // same block: both factors; neighbouring blocks: the factor that joins them
            const int f1 = ka == kb ? fi : (ka == kb + 1 ? fi : -1), l1a = ka == kb ? 6 + ca : 21 + ca;
            const int f2 = ka == kb ? fj : (ka == kb - 1 ? fj : -1), l2a = ka == kb ? 21 + ca : 6 + ca;
            v = imu_entry(f1, l1a, 6 + cb) + imu_entry(f2, l2a, 21 + cb);
        } else {
            const int pr = i < 6 * K ? i / 6 : -7, ci = i - 6 * pr;
            const int f1 = (pr == kb || pr == kb + 1) ? fi : -1, l1a = pr == kb ? ci : 15 + ci;
            const int f2 = (pr == kb - 1 || pr == kb) ? fj : -1, l2a = pr == kb - 1 ? ci : 15 + ci;
            v = imu_entry(f1, min(max(l1a, 0), 29), 6 + cb) + imu_entry(f2, min(max(l2a, 0), 29), 21 + cb);
        }
        return v + prior_entry(i, j);
    }
    __device__ __forceinline__ double rhsraw(int j) const {
        const int NP = P.NV;
        const int kb = (j - NP) / 9, cb = (j - NP) - 9 * kb;
        const int fi = as_i[kb], fj = as_j[kb];
        double v = (fi >= 0 ? 1.0 : 0.0) * P.ipart[(size_t)max(fi, 0) * 931 + 900 + 6 + cb] + (fj >= 0 ? 1.0 : 0.0) * P.ipart[(size_t)max(fj, 0) * 931 + 900 + 21 + cb];
        if (P.pn > 0) { const int pj = pinv[j]; v += (pj >= 0 ? 1.0 : 0.0) * P.mpart[max(pj, 0)]; }
        return v;
    }
};

// Raw entries staged in LDS (the chain workgroup's second phase reads them like the step kernel reads S'): assembling an entry
// from the partial records costs six dependent global loads and some index arithmetic -- on the recursion's critical path that
// made the chain three times slower than inside the step kernel (52 us against 14.6 us at K = 10, measured).
//   dg[k][45]  lower triangle of the diagonal block k         sub[k][81]  rows of block k+1 x columns of block k (k < K-1)
//   pb[col][row]  pose rows x chain columns, stride NPs        rhs[9K]
struct ChainSlab { double* dg; double* sub; double* pb; double* rhs; int NPs; };
__host__ __device__ inline size_t chain_slab_doubles(int K) { const int NPs = (6 * K + 7 + 1) & ~1; return even_up(45 * K) + even_up(81 * K) + (size_t)9 * K * NPs + even_up(9 * K) + (size_t)(2 * K + 15 * K + 7 + 9 * K + 8) / 2 + 8; }   // + the int tables
__device__ __forceinline__ ChainSlab chain_slab(double* p, int K) {
    ChainSlab S; S.NPs = (6 * K + 7 + 1) & ~1;
    S.dg = p; S.sub = S.dg + even_up(45 * K); S.pb = S.sub + even_up(81 * K); S.rhs = S.pb + (size_t)9 * K * S.NPs;
    return S;
}
struct ChainSrcSlab {
    const DevP& P; const ChainSlab& B; const double* scB; const double* dcB; const double* uB; double mu;
    __device__ __forceinline__ double raw(int i, int j) const {
        const int NP = P.NV;
        const int kb = (j - NP) / 9, cb = (j - NP) - 9 * kb;
        if (i < NP) return B.pb[(size_t)(9 * kb + cb) * B.NPs + i];
        const int ka = (i - NP) / 9, ca = (i - NP) - 9 * ka;
        if (ka == kb) return B.dg[45 * kb + (max(ca, cb) * (max(ca, cb) + 1) >> 1) + min(ca, cb)];
        return ka > kb ? B.sub[81 * kb + ca * 9 + cb] : B.sub[81 * ka + cb * 9 + ca];      // rows of the higher block x columns of the lower one
    }
    __device__ __forceinline__ double diag(int k, int i, int j) const { return B.dg[45 * k + (i * (i + 1) >> 1) + j]; }
    __device__ __forceinline__ double sub(int k, int kn, int q, int c) const { return kn > k ? B.sub[81 * k + q * 9 + c] : B.sub[81 * kn + c * 9 + q]; }
    __device__ __forceinline__ double prow(int r, int k, int c) const { return B.pb[(size_t)(9 * k + c) * B.NPs + r]; }
    __device__ __forceinline__ double rhsraw(int j) const { return B.rhs[j - P.NV]; }
    __device__ __forceinline__ double sc(int j) const { return scB[j - P.NV]; }
    __device__ __forceinline__ double madd(int j) const { const double d = dcB[j - P.NV]; return mu * d * d; }
    __device__ __forceinline__ double rowscale(int) const { return 1.0; }
    __device__ __forceinline__ double u(int j) const { return uB[j - P.NV]; }
    __device__ __forceinline__ void row_done(int d, int r, double zr, double&) const { P.chZ[(size_t)d * (P.NV + 1) + r] = zr; }
};

// the chain workgroup (>= 384 threads; dynamic LDS = chain scratch + 3 x 9K doubles + slab).  wait: spin until the IMU / prior
// workgroups of this launch have published their records (k_sweep); k_reduce_pc runs after the sweep and passes false.
__device__ __forceinline__ void reduce_chain_wg(const DevP& P, const Ctl& ctl, const int jacobi, double* lds, const bool wait) {
    const int t = threadIdx.x, K = P.K, NP = P.NV, NB = 9 * K, NT = blockDim.x;
#ifdef VIL_STAMPS
    #define PSTAMP(k) do { if (t == 0) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); P.dbg[k] = tt_; } } while (0)
#else
    #define PSTAMP(k) do {} while (0)
#endif
    const ChainLds L = chain_lds(lds, K);
    double* scB = lds + chain_scratch_doubles(K); double* dcB = scB + even_up(NB); double* uB = dcB + even_up(NB);
    const ChainSlab B = chain_slab(uB + even_up(NB), K);
    // ---- before the records exist: flags, index tables into LDS, zero the pose x chain slab ------------------------------------
    int* tab = (int*)(B.rhs + even_up(NB));            // as_i[K] | as_j[K] | pinv[D] | prior chain columns
    int* as_i = tab; int* as_j = tab + K; int* pinv = tab + 2 * K; int* pcol = pinv + P.D;
    if (t < 8) L.flag[t] = 0;
    for (int e = t; e < K; e += NT) { as_i[e] = P.imu_as_i[e]; as_j[e] = P.imu_as_j[e]; }
    for (int e = t; e < P.D; e += NT) pinv[e] = P.pinv[e];
    for (int e = t; e < P.ch_npc; e += NT) pcol[e] = P.ch_pcol[e];
    for (size_t e = t; e < (size_t)NB * B.NPs; e += NT) B.pb[e] = 0.0;
    PSTAMP(12);
    if (wait) {
        const int epoch = (int)((((unsigned)ctl.gen) << 12) + (unsigned)ctl.swe + 1u);       // (see sweep_signal)
        if (t <= P.n_imu && (t < P.n_imu || P.pn > 0)) while (__hip_atomic_load(P.swflag + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
    PSTAMP(13);
    // ---- phase 1: stage every structurally non-zero raw entry.  One flat item list, a fixed number of rounds per thread with the
    //      loads of all rounds issued before the first store: one L2 round trip for the whole phase instead of one per entry.
    const ChainSrcPart part{P, as_i, as_j, pinv};
    const int n0 = 45 * K, n1 = n0 + 81 * (K - 1), n2 = n1 + NB, n3 = n2 + 162 * K, n4 = n3 + P.ch_npc * NP;
    auto item = [&](int e, double*& dst) -> double {   // value and LDS destination of item e (dst = nullptr: nothing to store)
        dst = nullptr;
        if (e < n0) {
            const int k = e / 45, q = e - 45 * k;
            int di = 0; while ((di + 1) * (di + 2) / 2 <= q) ++di;
            const int dj = q - di * (di + 1) / 2;
            dst = B.dg + e; return part.raw(NP + 9 * k + di, NP + 9 * k + dj);
        }
        if (e < n1) { const int f = e - n0, k = f / 81, q = f - 81 * k; dst = B.sub + f; return part.raw(NP + 9 * (k + 1) + q / 9, NP + 9 * k + q % 9); }
        if (e < n2) { const int f = e - n1; dst = B.rhs + f; return part.rhsraw(NP + f); }
        if (e < n3) {                                  // pose rows of frames k-1, k, k+1 x the columns of block k (IMU factors)
            const int f = e - n2, k = f / 162, q = f - 162 * k, slot = q / 9, c = q - 9 * slot, fr = k - 1 + slot / 6;
            if (fr < 0 || fr >= K) return 0.0;
            const int row = 6 * fr + slot % 6;
            dst = B.pb + (size_t)(9 * k + c) * B.NPs + row; return part.raw(row, NP + 9 * k + c);
        }
        if (e < n4) {                                  // every pose row x the chain columns the prior holds (one speed-bias block in VINS)
            const int f = e - n3, q = f / NP, row = f - q * NP, jc = pcol[q];
            dst = B.pb + (size_t)jc * B.NPs + row; return part.raw(row, NP + jc);      // (an IMU row staged twice gets the same value twice)
        }
        return 0.0;
    };
    for (int e0 = t; e0 < n4; e0 += 4 * NT) {
        double v[4]; double* dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = item(min(e0 + u * NT, n4 - 1), dst[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (e0 + u * NT < n4 && dst[u]) *dst[u] = v[u];
    }
    __syncthreads();
    const ChainSrcSlab src{P, B, scB, dcB, uB, ctl.mu};
    PSTAMP(14);
    if (t < NB) {
        const int j = NP + t;
        const double dg = src.diag(t / 9, t % 9, t % 9), b = src.rhsraw(j);
        const double Sc = ctl.first ? (jacobi ? 1.0 / (1.0 + sqrt(dg)) : 1.0) : P.Sc[j];
        const double d = sqrt(fmin(fmax(Sc * Sc * dg, 1e-6), 1e32));
        scB[t] = Sc; dcB[t] = d; uB[t] = Sc * (Sc * b / d) / d;
        P.chSc[t] = Sc; P.chDc[t] = d;
    }
    __syncthreads();
    // ---- phase 2: the chain ----------------------------------------------------------------------------------------------------
    double qc = 0.0;
    PSTAMP(15);
    chain_eliminate<true>(src, K, NP, P.chain_rs, P.chW, L, qc);
    PSTAMP(16);
    if (t < 128) {                                     // the two recursion waves hold the chain x chain share of u^T S' u
        qc = wave_total(qc);
        if ((t & 63) == 0) P.chQ[t >> 6] = qc;
    }
    __syncthreads();
    for (int e = t; e < 54 * K; e += NT) P.chLdg[e] = L.Ldg[e];
    for (int e = t; e < 82 * K; e += NT) P.chLsb[e] = L.Lsb[e];
    if (t == 0) P.chOk[0] = L.flag[5] ? 0 : 1;
    PSTAMP(17);
}

}  // namespace vd

// Gather of the sweep's partial records (vil_sweep.hpp: reduce_gather): 256 threads, a handful of registers.
__global__ __launch_bounds__(VIL_THREADS) void k_reduce(DevP P) {
    const Ctl ctl = *P.ctl;
    if (ctl.done) return;
    reduce_gather(P, ctl);
}

// The sweep: grid = n_imu + 2 (+ 1 chain workgroup when P.prechain) + n_vwg + ceil(n_pchunk / 2) + ceil(n_echunk / 2)
// workgroups of VIL_SWEEP_THREADS threads (roles: vil_sweep.hpp)
__global__ __launch_bounds__(VIL_SWEEP_THREADS) void k_sweep(DevP P, SolveOpts O) {
    extern __shared__ double sm[];
    const Ctl ctl = *P.ctl;
    if (ctl.done) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) P.ctl->n_sweeps = ctl.n_sweeps + 1;   // live (not early-exited) launches, for the profiler
    const int cand = 1 - ctl.cur;
    const double* x = P.x[cand];
    SysBuf sb = P.sys[cand];
    int b = blockIdx.x;
    if (b < P.n_imu) { if (!(P.skip_mask & 2)) vd::sweep_imu(P, O, b, x, sm); if (P.prechain) sweep_signal(P, ctl, b); return; }
    b -= P.n_imu;
    if (b == 0) { if (!(P.skip_mask & 16)) vd::sweep_prior(P, x, sm); if (P.prechain) sweep_signal(P, ctl, P.n_imu); return; }
    if (b == 1) {
        if (!(P.skip_mask & 16)) vd::sweep_misc(P, O, x, sm);
        if (P.world > 1) {                               // factor set sharded over ranks: the visual workgroups of this rank form the candidate inverse depth of
            const double* xcur = P.x[ctl.cur];           // the landmarks it owns; every rank holds la / lb of ALL landmarks (the step kernel runs on the all-reduced
            double* xcand = P.x[1 - ctl.cur];            // system), so the rest is filled in here and the states stay identical on all ranks
            for (int l = threadIdx.x; l < P.L; l += blockDim.x) xcand[xo_lam(P) + l] = xcur[xo_lam(P) + l] + ctl.cg * P.la[l] + ctl.cn * P.lb[l];
        }
        return;
    }
    b -= 2;
    if (P.prechain) { if (b == 0) { vd::reduce_chain_wg(P, ctl, O.jacobi_scaling, sm, true); return; } b -= 1; }
    // visual workgroups next: the longest-running factor role
    if (b < P.n_vwg) { if (!(P.skip_mask & 1)) vd::sweep_visual(P, O, ctl, b, x, sb, sm); return; }
    b -= P.n_vwg;
    const int per = VIL_SWEEP_THREADS / 256, npw = (P.n_pchunk + per - 1) / per;
    if (b < npw) { if (!(P.skip_mask & 4)) vd::sweep_lidar<1>(P, O, b, x, sm); return; }
    b -= npw;
    if (!(P.skip_mask & 8)) vd::sweep_lidar<3>(P, O, b, x, sm);
}
