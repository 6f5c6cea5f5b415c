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
    const DevP& P; const double* scB; const double* dcB; const double* uB; double mu;     // LDS vectors over the chain columns
    __device__ __forceinline__ double imu_entry(int f, int la, int lb) const {
        const double m = f >= 0 ? 1.0 : 0.0;
        return m * P.ipart[(size_t)max(f, 0) * 931 + la * 30 + lb];
    }
    __device__ __forceinline__ double prior_entry(int i, int j) const {
        if (P.pn <= 0) return 0.0;
        const int pi = P.pinv[i], pj = P.pinv[j];
        const double m = (pi >= 0 && pj >= 0) ? 1.0 : 0.0;
        return m * P.pH[(size_t)max(pi, 0) * P.pn + max(pj, 0)];
    }
    // raw S'(i, j), j in the chain part
    __device__ __forceinline__ double raw(int i, int j) const {
        const int NP = P.NV, K = P.K;
        const int kb = (j - NP) / 9, cb = (j - NP) - 9 * kb;
        const int fi = P.imu_as_i[kb], fj = P.imu_as_j[kb];         // factor (kb, kb+1) and factor (kb-1, kb)
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
        const int fi = P.imu_as_i[kb], fj = P.imu_as_j[kb];
        double v = (fi >= 0 ? 1.0 : 0.0) * P.ipart[(size_t)max(fi, 0) * 931 + 900 + 6 + cb] + (fj >= 0 ? 1.0 : 0.0) * P.ipart[(size_t)max(fj, 0) * 931 + 900 + 21 + cb];
        if (P.pn > 0) { const int pj = P.pinv[j]; v += (pj >= 0 ? 1.0 : 0.0) * P.mpart[max(pj, 0)]; }
        return v;
    }
    __device__ __forceinline__ double sc(int j) const { return scB[j - P.NV]; }
    __device__ __forceinline__ double madd(int j) const { const double d = dcB[j - P.NV]; return mu * d * d; }
    __device__ __forceinline__ double rowscale(int) const { return 1.0; }
    __device__ __forceinline__ double u(int j) const { return uB[j - P.NV]; }
    __device__ __forceinline__ void row_done(int d, int r, double zr, double&) const { if (r < P.NV) P.chZ[(size_t)d * (P.NV + 1) + r] = zr; }
};

// the chain workgroup of k_reduce (VIL_REDUCE_THREADS threads, dynamic LDS = chain scratch + 3 x 9K doubles)
__device__ __forceinline__ void reduce_chain_wg(const DevP& P, const Ctl& ctl, const int jacobi, double* lds) {
    const int t = threadIdx.x, K = P.K, NP = P.NV, NB = 9 * K;
    const ChainLds L = chain_lds(lds, K);
    double* scB = lds + chain_scratch_doubles(K); double* dcB = scB + even_up(NB); double* uB = dcB + even_up(NB);
    const ChainSrcPart src{P, scB, dcB, uB, ctl.mu};
    if (t < 8) L.flag[t] = 0;
    if (t < NB) {
        const int j = NP + t;
        const double dg = src.raw(j, j), b = src.rhsraw(j);
        const double Sc = ctl.first ? (jacobi ? 1.0 / (1.0 + sqrt(dg)) : 1.0) : P.Sc[j];
        const double d = sqrt(fmin(fmax(Sc * Sc * dg, 1e-6), 1e32));
        scB[t] = Sc; dcB[t] = d; uB[t] = Sc * (Sc * b / d) / d;
        P.chSc[t] = Sc; P.chDc[t] = d;
    }
    __syncthreads();
    double qc = 0.0;
    chain_eliminate<true>(src, K, NP, P.chain_rs, P.chW, L, qc);
    if (t < 128) {                                     // the two recursion waves hold the chain x chain share of u^T S' u
        qc = wave_total(qc);
        if ((t & 63) == 0) P.chQ[t >> 6] = qc;
    }
    __syncthreads();
    for (int e = t; e < 54 * K; e += VIL_REDUCE_THREADS) P.chLdg[e] = L.Ldg[e];
    for (int e = t; e < 82 * K; e += VIL_REDUCE_THREADS) P.chLsb[e] = L.Lsb[e];
    if (t == 0) P.chOk[0] = L.flag[5] ? 0 : 1;
}

}  // namespace vd

// Gather of the sweep's partial records (vil_sweep.hpp: reduce_gather).  k_reduce is the lean kernel (256 threads, a handful of
// registers: several workgroups per CU); k_reduce_pc carries the chain workgroup as its last block and is launched instead
// when P.prechain -- the chain code needs 384 threads and ~250 registers, which would cut the occupancy of every gather block.
__global__ __launch_bounds__(VIL_THREADS) void k_reduce(DevP P) {
    const Ctl ctl = *P.ctl;
    if (ctl.done) return;
    reduce_gather(P, ctl);
}
__global__ __launch_bounds__(VIL_REDUCE_THREADS) void k_reduce_pc(DevP P, int jacobi) {
    extern __shared__ double rlds[];
    const Ctl ctl = *P.ctl;
    if (ctl.done) return;
    if (blockIdx.x == gridDim.x - 1) { vd::reduce_chain_wg(P, ctl, jacobi, rlds); return; }
    reduce_gather(P, ctl);
}
