// The end of a solve, on the device and in stream order: the accepted state becomes x[0] (and x[1]), double2vector()'s yaw /
// translation gauge fix is applied (estimator.cpp:960-1011), and Ctl + the final state are left in the host's pinned, device-mapped
// mirror followed by the solve generation -- the host polls that word and has everything vil_solve returns without a further launch,
// copy or synchronisation.  Executed by block 0 of the first k_sweep launch that finds the solve finished (every chunk of iterations
// ends with one more sweep launch), or by k_finish when the host ended the solve (time cap).
#pragma once
#include <math.h>
#include "vil_dev.hpp"
#include "vil_math.hpp"

// estimator.cpp:960-1011 double2vector(): yaw + translation gauge fix.  One arithmetic for the host entry point (vil_gauge_fix)
// and the device kernel (vil_set_gauge_fix): pose K x 7 [p q(xyzw)], speed-bias K x 9, ex 7.
__host__ __device__ inline void gauge_q2R(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
__host__ __device__ inline void gauge_R2ypr(const double* R, double* ypr) {
    const double y = atan2(R[3], R[0]);
    const double pch = atan2(-R[6], R[0] * cos(y) + R[3] * sin(y));
    const double rl = atan2(R[2] * sin(y) - R[5] * cos(y), -R[1] * sin(y) + R[4] * cos(y));
    ypr[0] = y / M_PI * 180.0; ypr[1] = pch / M_PI * 180.0; ypr[2] = rl / M_PI * 180.0;
}
__host__ __device__ inline void gauge_R2q(const double* R, double* q /*xyzw*/) {      // Eigen's Quaternion(Matrix3): the three off-trace cases written out (no dynamic register indexing: scratch)
    double t = R[0] + R[4] + R[8];
    if (t > 0) { t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t; }
    else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > (i ? R[4] : R[0])) i = 2;
        if (i == 0) {          // j = 1, k = 2
            t = sqrt(R[0] - R[4] - R[8] + 1.0);
            q[0] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[7] - R[5]) * t; q[1] = (R[3] + R[1]) * t; q[2] = (R[6] + R[2]) * t;
        } else if (i == 1) {   // j = 2, k = 0
            t = sqrt(R[4] - R[8] - R[0] + 1.0);
            q[1] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[2] - R[6]) * t; q[2] = (R[7] + R[5]) * t; q[0] = (R[1] + R[3]) * t;
        } else {               // j = 0, k = 1
            t = sqrt(R[8] - R[0] - R[4] + 1.0);
            q[2] = 0.5 * t; t = 0.5 / t;
            q[3] = (R[3] - R[1]) * t; q[0] = (R[2] + R[6]) * t; q[1] = (R[5] + R[7]) * t;
        }
    }
}
__host__ __device__ inline void gauge_rot(const double* pose0_before, const double* pose0_now, double* rot) {
    double R0[9], R00[9], a0[3], a00[3];
    gauge_q2R(pose0_before + 3, R0); gauge_q2R(pose0_now + 3, R00);
    gauge_R2ypr(R0, a0); gauge_R2ypr(R00, a00);
    const double yd = (a0[0] - a00[0]) / 180.0 * M_PI;
    rot[0] = cos(yd); rot[1] = -sin(yd); rot[2] = 0; rot[3] = sin(yd); rot[4] = cos(yd); rot[5] = 0; rot[6] = 0; rot[7] = 0; rot[8] = 1;
    if (fabs(fabs(a0[1]) - 90) < 1.0 || fabs(fabs(a00[1]) - 90) < 1.0)
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double v = 0; for (int k = 0; k < 3; ++k) v += R0[3 * i + k] * R00[3 * j + k]; rot[3 * i + j] = v; }
}
__host__ __device__ inline void gauge_frame(const double* rot, const double* p0, const double* pose0_before, double* pp, double* sb) {
    {
        double qn[4], Rf[9], Rn[9], d[3], Pn[3], V[3];
        { const double n = sqrt(pp[3] * pp[3] + pp[4] * pp[4] + pp[5] * pp[5] + pp[6] * pp[6]); for (int i = 0; i < 4; ++i) qn[i] = pp[3 + i] / n; }
        gauge_q2R(qn, Rf);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double v = 0; for (int k = 0; k < 3; ++k) v += rot[3 * i + k] * Rf[3 * k + j]; Rn[3 * i + j] = v; }
        for (int i = 0; i < 3; ++i) d[i] = pp[i] - p0[i];
        for (int i = 0; i < 3; ++i) Pn[i] = rot[3 * i] * d[0] + rot[3 * i + 1] * d[1] + rot[3 * i + 2] * d[2] + pose0_before[i];
        gauge_R2q(Rn, pp + 3); pp[0] = Pn[0]; pp[1] = Pn[1]; pp[2] = Pn[2];
        for (int i = 0; i < 3; ++i) V[i] = rot[3 * i] * sb[0] + rot[3 * i + 1] * sb[1] + rot[3 * i + 2] * sb[2];
        sb[0] = V[0]; sb[1] = V[1]; sb[2] = V[2];
    }
}
__host__ __device__ inline void gauge_ex(double* ex_pose) {
    double qe[4], Re[9];
    { const double n = sqrt(ex_pose[3] * ex_pose[3] + ex_pose[4] * ex_pose[4] + ex_pose[5] * ex_pose[5] + ex_pose[6] * ex_pose[6]); for (int i = 0; i < 4; ++i) qe[i] = ex_pose[3 + i] / n; }
    gauge_q2R(qe, Re); gauge_R2q(Re, ex_pose + 3);
}
__host__ __device__ inline void gauge_fix_core(const double* pose0_before, int K, double* pose, double* speedbias, double* ex_pose) {
    double rot[9];
    gauge_rot(pose0_before, pose, rot);
    const double p0[3] = {pose[0], pose[1], pose[2]};
    for (int f = 0; f < K; ++f) gauge_frame(rot, p0, pose0_before, pose + 7 * f, speedbias + 9 * f);
    gauge_ex(ex_pose);
}

namespace vd {
// all threads of ONE workgroup; returns after the sequence word has been stored
// (cur / status / gen: the scalars of Ctl the caller already holds; the record itself is copied from device memory)
// cam: 16 K + 8 + 12 doubles of LDS for the camera part of the final state and the gauge correction
// AG (the persistent solve, k_solve): the accepted state and Ctl were written by workgroups of THIS launch -- read at agent scope, Ctl from the caller's copy `lctl`
template <bool AG = false>
__device__ __forceinline__ void solve_finish(double* const x, double* const xb, const double* const xorig, double* const hs, Ctl* const dctl, Ctl* const hctl, int* const hseq,
                                             const int K, const int NS, const int gauge_on, const int cur, const int status, const int gen, double* const cam, const Ctl* const lctl = nullptr) {
    const int t = vil_tid(), NT = blockDim.x;
    const double* xs = cur ? xb : x;
    const int NC = 16 * K + 8;
    const int o_ex = 16 * K;                            // (xo_pose(k) = 7 k, xo_sb(k) = 7 K + 9 k, xo_ex = 16 K: vil_dev.hpp)
    for (int i = t; i < NC; i += NT) cam[i] = vd::ldx<AG>(xs + i);
    __syncthreads();
    if (gauge_on && status == 0) {
        // the yaw (or, near the singular pitch, the full) correction is derived from frame 0 by ONE lane and handed on through LDS (behind the camera part:
        // cam has 16 K + 8 + 12 doubles); then one thread per frame (+ one for the extrinsic)
        double* const gr = cam + NC;
        if (t == 0) { double rot[9]; gauge_rot(xorig, cam, rot); for (int i = 0; i < 9; ++i) gr[i] = rot[i]; for (int i = 0; i < 3; ++i) gr[9 + i] = cam[i]; }
        __syncthreads();
        if (t <= K) {
            double rot[9], p0[3];
            for (int i = 0; i < 9; ++i) rot[i] = gr[i];
            for (int i = 0; i < 3; ++i) p0[i] = gr[9 + i];
            if (t < K) gauge_frame(rot, p0, xorig, cam + 7 * t, cam + 7 * K + 9 * t);
            else gauge_ex(cam + o_ex);
        }
        __syncthreads();
    }
    for (int i = t; i < NS; i += NT) {
        const double v = i < NC ? cam[i] : vd::ldx<AG>(xs + i);
        x[i] = v; xb[i] = v;
        if (hs) hs[i] = v;
    }
    if (hctl) {
        const double* src = lctl ? (const double*)lctl : (const double*)dctl; double* h = (double*)hctl;
        for (int i = t; i < (int)(sizeof(Ctl) / 8); i += NT) h[i] = src[i];
    }
    __threadfence_system();                              // every thread's stores (device and host) before the barrier: __syncthreads alone does not wait for them
    __syncthreads();
    if (t == 0) {
        dctl->outd = 1;
        if (hseq) __hip_atomic_store(hseq, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
}  // namespace vd

#ifndef VIL_PERSIST_TU
// term < 0: the last launch of every chunk of iterations -- writes the result out if the solve ended in the chunk's last iteration (no
// sweep launch behind it to do so), a no-op otherwise.  term >= 0: the host ended the solve (max_solver_time_in_seconds,
// estimator.cpp:1411): mark it done with that termination and write the accepted state out.
__global__ __launch_bounds__(VIL_SWEEP_THREADS) void k_finish(DevP P, int term) {
    __shared__ int sc[4];
    __shared__ double cam[16 * 20 + 8 + 12];             // camera part of the final state (K <= 20) + the gauge rotation / origin
    if (vil_tid() == 0) {
        Ctl* c = P.ctl;
        if (!c->done && term >= 0) { c->done = 1; c->term = term; }
        if (!c->done) c = nullptr;
        if (!c) { sc[0] = 1; }
        else {
        sc[0] = c->outd || c->lin_mode != 0; sc[1] = c->cur; sc[2] = c->status; sc[3] = c->gen;
        }
        __threadfence();
    }
    __syncthreads();
    if (sc[0]) return;
    vd::solve_finish(P.x[0], P.x[1], P.xorig, P.hstate, P.ctl, P.hctl, P.hseq, P.K, P.NS, P.gauge_on, sc[1], sc[2], sc[3], cam);
}

// Start of a solve / linearisation / marginalisation sweep: the trust-region record is written by a kernel from its ARGUMENTS (no pinned staging
// buffer whose contents a later call could overwrite before an asynchronous copy has read it).  reset: both state buffers are first restored to the
// state the window was uploaded with (vil_reset_state + vil_solve_resident of a bench loop: one launch).
// xsave != null (a solve): the state the solve starts from (x0 = x1 here) is kept aside -- a one-launch solve whose wait gave up is re-run from it with the
// multi-launch structure, and a solve that fails altogether leaves the resident state as it found it (vilsolve.hip, vil_solve_resident)
__global__ __launch_bounds__(256) void k_solve_init(Ctl* ctl, int gen, double radius, double mu, int lin_mode, int* abortf /* cleared: vil_math.hpp, spin_until_eq */, const double* x0, double* xsave, int n) {
    if (xsave) { const int i = blockIdx.x * 256 + vil_tid(); if (i < n) xsave[i] = x0[i]; }
    if (blockIdx.x != 0) return;
    double* w = (double*)ctl;
    for (int i = vil_tid(); i < (int)(sizeof(Ctl) / 8); i += blockDim.x) w[i] = 0.0;
    if (abortf && vil_tid() == 0) *abortf = 0;
    __syncthreads();
    if (vil_tid() == 0) { ctl->gen = gen; ctl->first = 1; ctl->radius = radius; ctl->mu = mu; ctl->lin_mode = lin_mode; }
}
// (vil_reset_state is deferred to the next call that touches the state: in front of a solve it rides in the init launch -- one launch and one host
//  call gap less per solve of a bench / re-solve loop)
__global__ __launch_bounds__(256) void k_solve_init_reset(Ctl* ctl, int gen, double radius, double mu, int lin_mode, double* x0, double* x1, const double* src, int n, int* abortf, double* xsave) {
    const int i = blockIdx.x * 256 + vil_tid();
    if (i < n) { const double v = src[i]; x0[i] = v; x1[i] = v; if (xsave) xsave[i] = v; }
    if (blockIdx.x == 0) {
        double* w = (double*)ctl;
        for (int q = vil_tid(); q < (int)(sizeof(Ctl) / 8); q += blockDim.x) w[q] = 0.0;
        if (abortf && vil_tid() == 0) *abortf = 0;
        __syncthreads();
        if (vil_tid() == 0) { ctl->gen = gen; ctl->first = 1; ctl->radius = radius; ctl->mu = mu; ctl->lin_mode = lin_mode; }
    }
}
__global__ __launch_bounds__(256) void k_state_reset(double* x0, double* x1, const double* src, int n) {
    const int i = blockIdx.x * 256 + vil_tid();
    if (i < n) { const double v = src[i]; x0[i] = v; x1[i] = v; }
}
#endif
