// Device side of the fully resident window (include/vilsolve.h: vil_win_*).  What slideWindow() (estimator.cpp:1689-1814)
// leaves unchanged between two images stays in HBM, addressed by SLOTS that the host hands out and recycles -- sliding the window is
// an index remap on the host, nothing moves on the device:
//   observation store   obs[(frame slot * T + track slot) * 8 + c], c: x y z vx vy cur_td row -   (FeaturePerFrame, feature_manager.h:18-44)
//   IMU slots           raw samples (dt | acc | gyr), the interval's first measurement and bias linearisation point, its 287-double
//                       pre-integration record (k_preint writes it here) and the factored sqrt-information U (225)
//   LiDAR frame slabs   (vilsolve.hip: vil_lidar_push / drop)
//   prior slots (2)     J0 | r0 | x0 | J0^T J0 | J0^T r0 | r0^T r0: written by the marginalisation kernels, read by the next solve
// Per image the host sends ONE packet with the new frame (samples, observations, LiDAR points) and one with the small index tables of
// the window; k_win_pack expands the landmark table into the factor tables the sweep reads.
#pragma once
#include "vil_dev.hpp"
#include "vil_eval.hpp"

#define VIL_WIN_OBS 8
#define VIL_WIN_MAXK 20

struct WinFrameIn {           // one pushed frame, staged in device memory by a single DMA
    int n_obs; const int* track; const double* obs;          // n_obs x VIL_WIN_OBS
    double* store; int T, fslot;
    int ns; const double* samp;                              // [hdr 12 | dt ns | acc 3 ns | gyr 3 ns]
    double* hdr; double* dt; double* acc; double* gyr;       // the IMU slot's arrays
};
// blocks [0, nbo): observations scattered into the store (thread = (observation, component)); the rest: samples into the IMU slot
__global__ __launch_bounds__(256) void k_win_frame_in(WinFrameIn A, int nbo) {
    const int t = threadIdx.x;
    if ((int)blockIdx.x < nbo) {
        const int e = blockIdx.x * 256 + t, o = e >> 3, cidx = e & 7;
        if (o < A.n_obs) A.store[((size_t)A.fslot * A.T + A.track[o]) * VIL_WIN_OBS + cidx] = A.obs[(size_t)o * VIL_WIN_OBS + cidx];
        return;
    }
    const int b = blockIdx.x - nbo, nb = gridDim.x - nbo, ns = A.ns;
    for (int e = b * 256 + t; e < 12 + 7 * ns; e += nb * 256) {
        const double v = A.samp[e];
        if (e < 12) A.hdr[e] = v;
        else if (e < 12 + ns) A.dt[e - 12] = v;
        else if (e < 12 + 4 * ns) A.acc[e - 12 - ns] = v;
        else A.gyr[e - 12 - 4 * ns] = v;
    }
}

// MARGIN_SECOND_NEW (estimator.cpp:1763-1772): the newest interval's samples continue the previous interval's integration
__global__ __launch_bounds__(256) void k_win_append(double* dt, double* acc, double* gyr, int n0, const double* sdt, const double* sacc, const double* sgyr, int n1) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < 7 * n1; e += gridDim.x * 256) {
        if (e < n1) dt[n0 + e] = sdt[e];
        else if (e < 4 * n1) acc[3 * n0 + e - n1] = sacc[e - n1];
        else gyr[3 * n0 + e - 4 * n1] = sgyr[e - 4 * n1];
    }
}

// sqrt-information of one pre-integration record (imu_factor.h:64: LLT(cov^-1).matrixL()^T), once per record instead of once per upload
__global__ __launch_bounds__(VIL_THREADS) void k_imu_sqrtinfo(const double* rec, double* U, int* status, int skip_empty) {
    if (skip_empty && !(rec[16] > 0.0)) return;          // an interval without samples (the first frame of a window) is never a factor
    imu_sqrtinfo_wg(rec + 62, U, status);
}

struct WinPack {
    // visual: landmark l = track slot lm_track[l], anchored in window frame lm_startf[l], factors lm_start[l] .. lm_start[l+1] (one per later observation)
    int L, F, stride, T;
    const int* lm_start; const int* lm_track; const int* lm_startf; const double* store;
    int fslot[VIL_WIN_MAXK];
    double* vis_c; int* vis_i; int* vis_j; int* vis_l; int* fcol; const int* vfinv;
    // IMU factor f = (f, f + 1): record and U of the IMU slot of frame f + 1
    int n_imu; const double* rec; const double* U; int islot[VIL_WIN_MAXK]; double* imu_c; double* imu_U;
    // the state packet went into x[0]: the candidate buffer and the solve's origin are copies
    int NS; const double* x0; double* x1; double* xorig;
};
// blocks [0, nbf): thread = visual factor; [nbf, nbf + n_imu): one IMU factor each; the rest: state copies
__global__ __launch_bounds__(256) void k_win_pack(WinPack A, int nbf) {
    const int t = threadIdx.x;
    int b = blockIdx.x;
    if (b < nbf) {
        const int f = b * 256 + t;
        if (f >= A.F) return;
        int lo = 0, hi = A.L - 1;                        // last landmark with lm_start[l] <= f
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (A.lm_start[mid] <= f) lo = mid; else hi = mid - 1; }
        const int l = lo, q = f - A.lm_start[l] + 1, s = A.lm_startf[l], ts = A.lm_track[l];
        const double* oi = A.store + ((size_t)A.fslot[s] * A.T + ts) * VIL_WIN_OBS;
        const double* oj = A.store + ((size_t)A.fslot[s + q] * A.T + ts) * VIL_WIN_OBS;
        const int fs = A.vfinv[f];                       // sorted position of this factor: the sweep's tables are stored in its chunk order
        double* c = A.vis_c + fs; const size_t st = (size_t)A.stride;
        // [0:3) pts_i  [3:6) pts_j  [6:8) vel_i  [8:10) vel_j  [10] td_i  [11] td_j  [12] row_i  [13] row_j   (vilsolve.h)
        c[0] = oi[0]; c[st] = oi[1]; c[2 * st] = oi[2]; c[3 * st] = oj[0]; c[4 * st] = oj[1]; c[5 * st] = oj[2];
        c[6 * st] = oi[3]; c[7 * st] = oi[4]; c[8 * st] = oj[3]; c[9 * st] = oj[4];
        c[10 * st] = oi[5]; c[11 * st] = oj[5]; c[12 * st] = oi[6]; c[13 * st] = oj[6];
        A.vis_i[fs] = s; A.vis_j[fs] = s + q; A.vis_l[fs] = l; A.fcol[f] = 6 * (s + q);
        return;
    }
    b -= nbf;
    if (b < A.n_imu) {
        const double* r = A.rec + (size_t)A.islot[b + 1] * 287; const double* u = A.U + (size_t)A.islot[b + 1] * 225;
        for (int e = t; e < 287; e += 256) A.imu_c[(size_t)b * 287 + e] = r[e];
        for (int e = t; e < 225; e += 256) A.imu_U[(size_t)b * 225 + e] = u[e];
        return;
    }
    b -= A.n_imu;
    const int nb = gridDim.x - nbf - A.n_imu;
    for (int e = b * 256 + t; e < A.NS; e += nb * 256) { const double v = A.x0[e]; A.x1[e] = v; A.xorig[e] = v; }
}

// The new prior goes from the marginalisation kernels' work space straight into the resident prior slot: J0 (n x n column-major), r0,
// x0 = the kept blocks of the device state the factors were linearised at (marginalization_factor.cpp:110-139 keeps the very values),
// and the contractions the sweep's prior role works with (J0^T J0, J0^T r0, r0^T r0).  A non-finite entry raises the status word.
struct PriorCommit {
    int n, nblk; const double* J0; const double* r0; const double* x;      // x: device state [pose 7K | sb 9K | ex 7 | td 1 | ...]
    int K; int kind[VIL_WIN_MAXK + 4], index[VIL_WIN_MAXK + 4];           // kept blocks with their CURRENT frame index (before the shift)
    double* pJ0; double* pr0; double* px0; double* pH; double* pg0; double* pc0; int* status;
};
__global__ __launch_bounds__(256) void k_prior_commit(PriorCommit A) {
    extern __shared__ double sJ[];                       // J0 (n x n column-major, column stride n + 1: conflict-free column walks) | r0
    const int t = threadIdx.x, b = blockIdx.x, nb = gridDim.x, n = A.n, ns = n + 1;
    double* sr = sJ + (size_t)n * ns;
    bool bad = false;
    for (int e = t; e < n * n; e += 256) { const double v = A.J0[e]; if (!(v - v == 0.0)) bad = true; sJ[(e / n) * ns + (e % n)] = v; if (b == 0) A.pJ0[e] = v; }
    for (int e = t; e < n; e += 256) { const double v = A.r0[e]; if (!(v - v == 0.0)) bad = true; sr[e] = v; if (b == 0) A.pr0[e] = v; }
    if (bad) atomicExch(A.status, VIL_ERR_NON_FINITE);
    if (b == 0 && t == 0) {
        int xo = 0;
        for (int q = 0; q < A.nblk; ++q) {
            const int kind = A.kind[q], idx = A.index[q];
            const double* src = kind == VIL_BLK_POSE ? A.x + 7 * idx : (kind == VIL_BLK_SPEEDBIAS ? A.x + 7 * A.K + 9 * idx : (kind == VIL_BLK_EX ? A.x + 16 * A.K : A.x + 16 * A.K + 7));
            const int gs = (kind == VIL_BLK_POSE || kind == VIL_BLK_EX) ? 7 : (kind == VIL_BLK_SPEEDBIAS ? 9 : 1);
            for (int k = 0; k < gs; ++k) A.px0[xo + k] = src[k];
            xo += gs;
        }
    }
    __syncthreads();
    // the contractions out of LDS (column i of J0 at sJ[i * ns ...]): every workgroup staged the whole matrix (39 kB at n = 70), each computes a slice
    for (int e = b * 256 + t; e < n * n + n + 1; e += nb * 256) {
        if (e < n * n) {
            const int i = e / n, k = e % n;
            const double* ci = sJ + (size_t)i * ns; const double* ck = sJ + (size_t)k * ns;
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            int q = 0;
            for (; q + 3 < n; q += 4) { s0 += ci[q] * ck[q]; s1 += ci[q + 1] * ck[q + 1]; s2 += ci[q + 2] * ck[q + 2]; s3 += ci[q + 3] * ck[q + 3]; }
            for (; q < n; ++q) s0 += ci[q] * ck[q];
            A.pH[e] = (s0 + s1) + (s2 + s3);
        } else if (e < n * n + n) {
            const int i = e - n * n;
            const double* ci = sJ + (size_t)i * ns;
            double s0 = 0, s1 = 0;
            for (int q = 0; q < n; ++q) { if (q & 1) s1 += ci[q] * sr[q]; else s0 += ci[q] * sr[q]; }
            A.pg0[i] = s0 + s1;
        } else {
            double s0 = 0;
            for (int q = 0; q < n; ++q) s0 += sr[q] * sr[q];
            A.pc0[0] = s0;
        }
    }
}
