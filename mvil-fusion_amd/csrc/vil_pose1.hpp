// One-pose registration solve in ONE launch: the Ceres problem lidar_mapping builds per round (localMapping.cpp:596-600,
// :766-777: one 7-parameter pose block, HuberLoss(0.1) edge / plane factors, DOGLEG, 4 iterations) has six degrees of
// freedom and no landmarks, so the three-launches-per-iteration machinery of the window solver (sweep / reduce / step, made
// for 150 x 150 reduced systems) is all launch latency here.  k_pose_solve runs the same trust-region algorithm -- the one
// the window solver and oracle/oracle_solver.cpp implement: Jacobi scaling fixed at the first linearisation, traditional
// dogleg in the scaled space, mu retry loop around the Cholesky, step acceptance and the three tolerances -- with one
// launch: a handful of workgroups evaluate the factors (a thread per point, 21 + 6 + 1 accumulators in registers, wave
// butterfly, partial sums gathered in workgroup order), thread 0 of workgroup 0 does the 6 x 6 algebra.  The factor counts and the starting pose are read from device
// memory, the result pose is written back there, so a whole registration (2 x {search, fit, compact, solve}) is enqueued
// without a host round trip.
#pragma once
#include "../../include/vilsolve.h"
#include "vil_factors.hpp"

namespace vp1 {
using namespace vd;

#define VP1_THREADS 256
#define VP1_WAVES (VP1_THREADS / 64)
#define VP1_MAXG 64
#define VP1_MAX_ITER (1 << 20)       // iteration cap of one launch (the host reserves that many epochs)

struct PoseRT { double R[9]; double t[3]; };   // what the search kernel consumes (pointAssociateToMap)

struct Pose1Out {                // one record per round, read back once per registration
    double pose[7];              // t, q (x y z w)
    double initial_cost, final_cost;
    int iterations, successful_steps, termination, status, n_edge, n_plane;
};

struct Pose1In { double pose[7]; int use; int pad; };      // starting pose in the kernel arguments (use != 0) instead of a 56-byte upload
struct Pose1Shared {
    double red[VP1_WAVES][28];
    double mine[28];             // this workgroup's partial sums of the current evaluation
    double gath[VP1_MAXG][28];   // workgroup 0: everybody's partial sums
    double sysn[28];             // the gathered linearisation of the current evaluation: H upper triangle (21, row-major a <= b), g = J^T r (6), cost
    double x[7], cand[7];
    int go;
};

// Evaluation is spread over a few workgroups (one factor per thread at lidar_mapping's scan sizes; a single compute unit's
// fp64 rate would otherwise be the whole solve time); they meet through this block of device memory, zeroed at allocation.
// The exchange is symmetric: every workgroup publishes its 28 partial sums and a flag, waits for everybody's flag, sums all
// partials in workgroup order and runs the trust-region pass itself -- one hop per evaluation, no command to wait for.
// A workgroup can be one evaluation ahead of the slowest (it cannot finish evaluation e + 1 before everybody has published
// e + 1, i.e. has finished reading e), hence two partial buffers.  Every launch gets a fresh, growing range of epoch numbers
// from the host, so nothing is reset between launches.
struct Pose1Coop {
    int flag[VP1_MAXG];          // workgroup g: "my partial sums of evaluation `epoch` are in part[epoch & 1][g]" (epochs only grow)
    double part[2][VP1_MAXG][28];
};

// R(q) exactly as the host computes it for the first round (no fused multiply-adds: the scan points are rounded to float
// after the transform, the two rounds must see the same arithmetic)
__device__ __forceinline__ void pose_rt(const double* p /* t, q */, PoseRT& T) {
    const double x = p[3], y = p[4], z = p[5], w = p[6];
    auto m = [](double a, double b) { return __dmul_rn(a, b); };
    auto ad = [](double a, double b) { return __dadd_rn(a, b); };
    auto sb = [](double a, double b) { return __dsub_rn(a, b); };
    T.R[0] = sb(1.0, m(2.0, ad(m(y, y), m(z, z)))); T.R[1] = m(2.0, sb(m(x, y), m(w, z))); T.R[2] = m(2.0, ad(m(x, z), m(w, y)));
    T.R[3] = m(2.0, ad(m(x, y), m(w, z))); T.R[4] = sb(1.0, m(2.0, ad(m(x, x), m(z, z)))); T.R[5] = m(2.0, sb(m(y, z), m(w, x)));
    T.R[6] = m(2.0, sb(m(x, z), m(w, y))); T.R[7] = m(2.0, ad(m(y, z), m(w, x))); T.R[8] = sb(1.0, m(2.0, ad(m(x, x), m(y, y))));
    T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
}

// this workgroup's share of the cost and the normal equations at `pose`: factors g * VP1_THREADS + t, then strides of the whole
// grid; 28 sums (H upper triangle 21, g = J^T r 6, cost) -> sh.mine
__device__ __forceinline__ void pose1_eval(Pose1Shared& sh, const double* pose, int ne, const double* __restrict__ ed, int es, int np, const double* __restrict__ pl, int ps,
                                           int loss, double loss_scale, int prec) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, first = blockIdx.x * VP1_THREADS + t, stride = gridDim.x * VP1_THREADS;
    const M3 R = quatR(pose + 3);
    M3 I;
#pragma unroll
    for (int q = 0; q < 9; ++q) I.m[q] = (q % 4 == 0) ? 1.0 : 0.0;
    const V3 Pk{pose[0], pose[1], pose[2]}, zero{0.0, 0.0, 0.0};
    double acc[28];
#pragma unroll
    for (int q = 0; q < 28; ++q) acc[q] = 0.0;
    for (int f = first; f < np; f += stride) {
        double r, J[6], rho, rho1;
        plane_eval(V3{pl[f], pl[ps + f], pl[2 * ps + f]}, V3{pl[3 * ps + f], pl[4 * ps + f], pl[5 * ps + f]}, pl[6 * ps + f], I, zero, R, Pk, r, J, prec);
        loss_eval(loss, loss_scale, r * r, rho, rho1);
        acc[27] += 0.5 * rho;
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = a; b < 6; ++b) acc[idx++] += rho1 * (J[a] * J[b]);
            acc[21 + a] += rho1 * (J[a] * r);
        }
    }
    for (int f = first; f < ne; f += stride) {
        double r[3], J[18], rho, rho1;
        edge_eval(V3{ed[f], ed[es + f], ed[2 * es + f]}, V3{ed[3 * es + f], ed[4 * es + f], ed[5 * es + f]}, V3{ed[6 * es + f], ed[7 * es + f], ed[8 * es + f]}, I, zero, R, Pk, r, J, prec);
        loss_eval(loss, loss_scale, r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho, rho1);
        acc[27] += 0.5 * rho;
        int idx = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int b = a; b < 6; ++b) acc[idx++] += rho1 * (J[a] * J[b] + J[6 + a] * J[6 + b] + J[12 + a] * J[12 + b]);
            acc[21 + a] += rho1 * (J[a] * r[0] + J[6 + a] * r[1] + J[12 + a] * r[2]);
        }
    }
    double f0, f1;
    wave_fold<28>(acc, f0, f1);                           // lanes 0..15: totals of values rev4(lane) and 16 + rev4(lane)
    if (lane < 16) { const int q = fold_slot(lane); sh.red[wave][q] = f0; if (q + 16 < 28) sh.red[wave][q + 16] = f1; }
    __syncthreads();
    if (t < 28) { double s = 0.0; for (int w = 0; w < VP1_WAVES; ++w) s += sh.red[w][t]; sh.mine[t] = s; }
    __syncthreads();
}

__device__ __forceinline__ int tri6(int a, int b) { return a <= b ? a * 6 - ((a * (a - 1)) >> 1) + (b - a) : b * 6 - ((b * (b - 1)) >> 1) + (a - b); }

// the 6 x 6 trust-region state of thread 0 (oracle_solver.cpp Dogleg, restricted to the six free columns: the constant
// blocks of the one-pose window contribute identity rows, zero gradient and zero step, so nothing else enters).  Every
// loop is unrolled over compile-time indices, so the object and its scratch live in registers.
#define VP1_U _Pragma("unroll")
struct Dog6 {
    double Sc[6], dc[6], idc[6], grad[6], gn[6];     // idc = 1 / dc: the divisions of the reference become multiplications by a correctly rounded reciprocal
    double alpha, radius, mu, step_norm;             // (a lone thread's fp64 divide / sqrt latency is what this code costs)
    bool reuse;
    __device__ __forceinline__ static double quad(const double* H /*21, registers*/, const double* v) {
        double q = 0.0;
        VP1_U for (int i = 0; i < 6; ++i) { double row = 0.0; VP1_U for (int j = 0; j < 6; ++j) row += H[tri6(i, j)] * v[j]; q += v[i] * row; }
        return q;
    }
    __device__ __forceinline__ bool compute_system(const double* H, const double* b, double min_mu, double max_mu) {
        double v[6], g2 = 0.0;
        VP1_U for (int i = 0; i < 6; ++i) { const double s = fmin(fmax(Sc[i] * Sc[i] * H[tri6(i, i)], 1e-6), 1e32); idc[i] = rsqrt_nr(s); dc[i] = s * idc[i]; }
        VP1_U for (int i = 0; i < 6; ++i) grad[i] = Sc[i] * b[i] * idc[i];
        VP1_U for (int i = 0; i < 6; ++i) { v[i] = Sc[i] * grad[i] * idc[i]; g2 += grad[i] * grad[i]; }
        alpha = g2 / quad(H, v);
        bool ok = false;
        double xc[6];
        while (true) {
            double Lo[21], rd[6];                 // lower triangle, row i >= column j at tri6(j, i); rd = 1 / diagonal
            bool chol = true;
            VP1_U for (int j = 0; j < 6; ++j) {
                double d = Sc[j] * H[tri6(j, j)] * Sc[j] + mu * dc[j] * dc[j];
                VP1_U for (int k = 0; k < j; ++k) d -= Lo[tri6(k, j)] * Lo[tri6(k, j)];
                chol = chol && d > 0.0;
                rd[j] = rsqrt_nr(d); Lo[tri6(j, j)] = d * rd[j];          // hardware estimate + two Newton steps: the six pivots are one dependency chain
                VP1_U for (int i = j + 1; i < 6; ++i) { double s = Sc[i] * H[tri6(i, j)] * Sc[j]; VP1_U for (int k = 0; k < j; ++k) s -= Lo[tri6(k, i)] * Lo[tri6(k, j)]; Lo[tri6(j, i)] = s * rd[j]; }
            }
            if (chol) {
                VP1_U for (int i = 0; i < 6; ++i) { double s = Sc[i] * b[i]; VP1_U for (int k = 0; k < i; ++k) s -= Lo[tri6(k, i)] * xc[k]; xc[i] = s * rd[i]; }
                VP1_U for (int i = 5; i >= 0; --i) { double s = xc[i]; VP1_U for (int k = i + 1; k < 6; ++k) s -= Lo[tri6(i, k)] * xc[k]; xc[i] = s * rd[i]; }
                bool finite = true;
                VP1_U for (int i = 0; i < 6; ++i) finite = finite && isfinite(xc[i]);
                if (finite) { ok = true; break; }
            }
            mu *= 10.0;
            if (!(mu < max_mu)) break;
        }
        if (!ok) return false;
        mu = fmax(min_mu, 2.0 * mu / 10.0);
        VP1_U for (int i = 0; i < 6; ++i) gn[i] = -xc[i] * dc[i];
        return true;
    }
    __device__ __forceinline__ void dogleg_step(double* step) {
        double gn2 = 0.0, g2 = 0.0, gdotgn = 0.0;
        VP1_U for (int i = 0; i < 6; ++i) { gn2 += gn[i] * gn[i]; g2 += grad[i] * grad[i]; gdotgn += grad[i] * gn[i]; }
        const double gn_norm = sqrt(gn2), g_norm = sqrt(g2);
        double cg, cn;
        if (gn_norm <= radius) { cg = 0.0; cn = 1.0; step_norm = gn_norm; }
        else if (g_norm * alpha >= radius) { cg = -(radius / g_norm); cn = 0.0; step_norm = radius; }
        else {
            const double b_dot_a = -alpha * gdotgn, a2 = (alpha * g_norm) * (alpha * g_norm), bma2 = a2 - 2.0 * b_dot_a + gn2, cc = b_dot_a - a2;
            const double dd = sqrt(cc * cc + bma2 * (radius * radius - a2));
            const double beta = (cc <= 0.0) ? (dd - cc) / bma2 : (radius * radius - a2) / (dd + cc);
            cg = -alpha * (1.0 - beta); cn = beta; step_norm = radius;
        }
        VP1_U for (int i = 0; i < 6; ++i) step[i] = Sc[i] * (cg * grad[i] + cn * gn[i]) * idc[i];
    }
};

__device__ __forceinline__ void pose1_plus(const double* in, const double* d, double* o) {   // pose_local_parameterization.cpp:3-18
    for (int k = 0; k < 3; ++k) o[k] = in[k] + d[k];
    const Q4 q = qmul(qload(in + 3), Q4{1.0, 0.5 * d[3], 0.5 * d[4], 0.5 * d[5]});
    const double n = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    o[3] = q.x * n; o[4] = q.y * n; o[5] = q.z * n; o[6] = q.w * n;
}

// loop counters and costs of the trust-region logic (LDS; the trust-region state proper and the linearisation at x stay in
// thread 0's registers across evaluations: 256-thread workgroups leave it 512 of them)
struct Pose1State {
    double cost, initial_cost, model_change;
    int iter, nsucc, invalid_run, term, status, first;
    long long tk[6];             // VP1_STAMPS only
};

// One pass of the trust-region logic between two evaluations, thread 0 of workgroup 0: judge the candidate that was just
// evaluated (sh.sysn), then the checks before an iteration, the dogleg step and the next candidate (invalid steps retry
// without an evaluation).  Returns false when the solve is finished.  Not inlined: its register working set (everything
// unrolled over compile-time indices) must not compete with the evaluation loop's.
__device__ __forceinline__ bool pose1_serial(Pose1Shared& sh, Pose1State& st, Dog6& dg, double* Hs /*21*/, double* bs /*6*/, const vil_options& O) {
#ifdef VP1_STAMPS
    long long q0 = wall_clock64(), q1;
#define VP1_SK(k) do { q1 = wall_clock64(); st.tk[k] += q1 - q0; q0 = q1; } while (0)
#else
#define VP1_SK(k)
#endif
    double cost = st.cost, model_change = st.model_change;
    int iter = st.iter, invalid_run = st.invalid_run, term = st.term;
    bool done = false;
    if (st.first) {
        st.first = 0;
        VP1_U for (int q = 0; q < 21; ++q) Hs[q] = sh.sysn[q];
        VP1_U for (int q = 0; q < 6; ++q) bs[q] = sh.sysn[21 + q];
        cost = st.initial_cost = sh.sysn[27];
        VP1_U for (int i = 0; i < 6; ++i) dg.Sc[i] = O.jacobi_scaling ? 1.0 / (1.0 + sqrt(Hs[tri6(i, i)])) : 1.0;
        dg.radius = O.initial_radius; dg.mu = O.min_mu; dg.reuse = false; dg.alpha = 0.0; dg.step_norm = 0.0;
        if (!isfinite(cost)) { term = VIL_TERM_FAILURE; st.status = VIL_ERR_NON_FINITE; done = true; }
    } else {
        const double cand_cost = sh.sysn[27];
        double xn = 0.0, sn = 0.0;
        VP1_U for (int q = 0; q < 7; ++q) { xn += sh.x[q] * sh.x[q]; sn += (sh.x[q] - sh.cand[q]) * (sh.x[q] - sh.cand[q]); }
        if (sqrt(sn) <= O.parameter_tolerance * (sqrt(xn) + O.parameter_tolerance)) { term = VIL_TERM_PARAMETER_TOLERANCE; done = true; }
        else if (fabs(cost - cand_cost) <= O.function_tolerance * cost) { term = VIL_TERM_FUNCTION_TOLERANCE; done = true; }
        else {
            const double rel = (cost - cand_cost) / model_change;
            if (isfinite(cand_cost) && rel > O.min_relative_decrease) {
                VP1_U for (int q = 0; q < 7; ++q) sh.x[q] = sh.cand[q];
                VP1_U for (int q = 0; q < 21; ++q) Hs[q] = sh.sysn[q];
                VP1_U for (int q = 0; q < 6; ++q) bs[q] = sh.sysn[21 + q];
                cost = cand_cost; ++st.nsucc;
                if (rel < 0.25) dg.radius *= 0.5;
                if (rel > 0.75) dg.radius = fmax(dg.radius, 3.0 * dg.step_norm);
                dg.radius = fmin(O.max_radius, dg.radius); dg.reuse = false;
            } else { dg.radius *= 0.5; dg.reuse = true; }
        }
    }
    VP1_SK(0);
    while (!done) {
        double gmax = 0.0;
        VP1_U for (int i = 0; i < 6; ++i) gmax = fmax(gmax, fabs(bs[i]));
        if (iter >= min(O.max_iterations, VP1_MAX_ITER)) { term = VIL_TERM_MAX_ITERATIONS; done = true; break; }
        if (gmax <= O.gradient_tolerance) { term = VIL_TERM_GRADIENT_TOLERANCE; done = true; break; }
        if (dg.radius <= 1e-32) { term = VIL_TERM_FAILURE; done = true; break; }
        ++iter;
        bool valid = true;
        if (!dg.reuse) valid = dg.compute_system(Hs, bs, O.min_mu, O.max_mu);
        VP1_SK(1);
        double step[6];
        if (valid) {
            dg.dogleg_step(step);
            double gd = 0.0;
            VP1_U for (int i = 0; i < 6; ++i) gd += bs[i] * step[i];
            model_change = -(0.5 * Dog6::quad(Hs, step) + gd);
            valid = model_change > 0.0;
        }
        VP1_SK(2);
        if (valid) { invalid_run = 0; pose1_plus(sh.x, step, sh.cand); VP1_SK(3); break; }
        if (++invalid_run >= 5) { term = VIL_TERM_FAILURE; done = true; break; }
        dg.mu *= 10.0; dg.reuse = false;
    }
    st.cost = cost; st.model_change = model_change; st.iter = iter; st.invalid_run = invalid_run; st.term = term;
    VP1_SK(4);
    return !done;
}

// pose_io: t, q of the starting pose, overwritten with the result (left untouched when the solve fails); rt_out: the same
// pose as rotation matrix + translation for the next round's search.  cnt[0] = edges, cnt[1] = planes (device).
// prev: the previous round's record (nullptr in the first round) -- a failed round is propagated, not built upon.
// Grid: G <= VP1_MAXG workgroups, all resident (G is a handful); workgroup 0 writes the result.  Each candidate is linearised
// where it is evaluated (the window solver does the same), so an accepted step costs one evaluation, not two.
__global__ __launch_bounds__(VP1_THREADS) void k_pose_solve(const int* __restrict__ cnt, const double* __restrict__ ed, int es, const double* __restrict__ pl, int ps, double* pose_io,
                                                            PoseRT* rt_out, vil_options O, const Pose1Out* prev, Pose1Out* out, Pose1Coop* coop, int epoch,
                                                            Pose1In pin, Pose1Out* hout, int* hseq) {      // hout / hseq: pinned host mirror of the record + its sequence word (or null)
    const int epoch0 = epoch;
    __shared__ Pose1Shared sh;
    __shared__ Pose1State st;
    const int t = threadIdx.x, g = blockIdx.x, G = gridDim.x;
    const int ne = cnt[0], np = cnt[1];
    const int prev_status = prev ? prev->status : 0;
    if (prev_status != 0) {
        if (t == 0 && g == 0) {
            Pose1Out o = *prev; o.n_edge = ne; o.n_plane = np; *out = o;
            if (hout) { *hout = o; __threadfence_system(); __hip_atomic_store(hseq, epoch0 + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
        }
        return;
    }
    if (t < 7) sh.cand[t] = sh.x[t] = pin.use ? pin.pose[t] : pose_io[t];
    if (t == 0) { st.cost = st.initial_cost = st.model_change = 0.0; st.iter = st.nsucc = st.invalid_run = 0; st.term = VIL_TERM_NONE; st.status = 0; st.first = 1; for (int q = 0; q < 6; ++q) st.tk[q] = 0; }
    __syncthreads();
#ifdef VP1_STAMPS
    long long st0 = wall_clock64(), st_eval0 = 0, st_eval = 0, st_gather = 0, st_serial = 0, st_call = 0, stq;
#define VP1_ST(acc) do { stq = wall_clock64(); acc += stq - st0; st0 = stq; } while (0)
#else
#define VP1_ST(acc)
#endif
    Dog6 dg;                                  // live in thread 0 only
    double Hs[21], bs[6];
    VP1_U for (int q = 0; q < 21; ++q) Hs[q] = 0.0;
    VP1_U for (int q = 0; q < 6; ++q) bs[q] = 0.0;
    VP1_U for (int q = 0; q < 6; ++q) { dg.Sc[q] = 1.0; dg.dc[q] = dg.idc[q] = 1.0; dg.grad[q] = dg.gn[q] = 0.0; }
    dg.alpha = dg.radius = dg.mu = dg.step_norm = 0.0; dg.reuse = false;
    while (true) {
        ++epoch;
        double (*part)[28] = coop->part[epoch & 1];
        pose1_eval(sh, sh.cand, ne, ed, es, np, pl, ps, O.lidar_loss, O.lidar_loss_scale, O.precision);
        VP1_ST(st_eval0);
        // The partials are stored (and read) with agent-scope atomics -- at the level all XCDs share -- so waiting for wave 0's stores is
        // all the ordering the flag needs: a release fence would also write back this XCD's L2, an acquire on the other side invalidate theirs.
        if (t < 64) {
            if (t < 28) __hip_atomic_store(&part[g][t], sh.mine[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t == 0) __hip_atomic_store(&coop->flag[g], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        VP1_ST(st_eval);
        // every workgroup gathers everybody's partial sums (workgroup order: the same bits everywhere) ...
        if (t < G) while (__hip_atomic_load(&coop->flag[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch < 0) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        {   // 28 G <= 1792 values, up to 7 per thread: issue every load before the first use
            constexpr int NLD = (28 * VP1_MAXG + VP1_THREADS - 1) / VP1_THREADS;
            double gv[NLD];
#pragma unroll
            for (int u = 0; u < NLD; ++u) { const int e = t + u * VP1_THREADS; gv[u] = e < 28 * G ? __hip_atomic_load(&part[0][0] + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0; }
#pragma unroll
            for (int u = 0; u < NLD; ++u) { const int e = t + u * VP1_THREADS; if (e < 28 * G) (&sh.gath[0][0])[e] = gv[u]; }
        }
        __syncthreads();
        if (t < 28) {                  // four interleaved chains, combined in a fixed order: the same bits in every workgroup
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int w = 0;
            for (; w + 3 < G; w += 4) { s0 += sh.gath[w][t]; s1 += sh.gath[w + 1][t]; s2 += sh.gath[w + 2][t]; s3 += sh.gath[w + 3][t]; }
            for (; w < G; ++w) s0 += sh.gath[w][t];
            sh.sysn[t] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
        VP1_ST(st_gather);
        // ... and runs the same trust-region pass on them: identical code on identical numbers, so all workgroups agree on
        // the next candidate without another exchange
        if (t == 0) { sh.go = pose1_serial(sh, st, dg, Hs, bs, O) ? 1 : 0; VP1_ST(st_call); }
        __syncthreads();
        VP1_ST(st_serial);
        if (!sh.go) break;
    }
    if (g != 0) return;
#ifdef VP1_STAMPS
    if (t == 0) printf("pose1: G %d iters %d  eval-compute %lld  publish %lld  gather %lld  serial %lld  [10 ns ticks]  call %lld = judge %lld system %lld dogleg %lld plus %lld save %lld\n", G, st.iter, st_eval0, st_eval, st_gather, st_serial, st_call, st.tk[0], st.tk[1], st.tk[2], st.tk[3], st.tk[4]);
#endif
    if (t == 0) {
        const double cost = st.cost;
        bool finite = isfinite(cost);
        for (int q = 0; q < 7; ++q) finite = finite && isfinite(sh.x[q]);
        int status = st.status;
        if (status == 0) status = !finite ? VIL_ERR_NON_FINITE : (st.term == VIL_TERM_FAILURE ? VIL_ERR_NOT_POSITIVE_DEFINITE : 0);
        Pose1Out o;
        for (int q = 0; q < 7; ++q) o.pose[q] = finite ? sh.x[q] : (pin.use ? pin.pose[q] : pose_io[q]);
        o.initial_cost = st.initial_cost; o.final_cost = cost; o.iterations = st.iter; o.successful_steps = st.nsucc; o.termination = st.term; o.status = status; o.n_edge = ne; o.n_plane = np;
        *out = o;
        if (finite) {
            for (int q = 0; q < 7; ++q) pose_io[q] = sh.x[q];
            PoseRT T; pose_rt(sh.x, T); *rt_out = T;
        } else if (pin.use) { for (int q = 0; q < 7; ++q) pose_io[q] = pin.pose[q]; }
        if (hout) { *hout = o; __threadfence_system(); __hip_atomic_store(hseq, epoch0 + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }      // the host polls this word
    }
}

}  // namespace vp1
