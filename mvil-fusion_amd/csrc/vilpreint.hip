// vilpreint.hip -- IMU pre-integration on gfx950 behind include/vilpreint.h (SURVEY 8(f) row 3 / 8(a) A5,
// factor/integration_base.h:30-158).
//
// One workgroup per interval; the intervals of a window run concurrently.  The recurrences of one interval are sequential in
// the samples only on paper: the transition matrices F_s (15 x 15) and noise maps V_s (15 x 18) depend on the 10-number state
// (delta_p, delta_q, delta_v) alone, the state itself is a prefix product of quaternion increments plus two prefix sums, and
// the affine covariance maps compose associatively -- see k_preint.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/vilpreint.h"
#include "vil_tuning.hpp"

#define VP_OK 0
#define VP_ERR_INVALID -1
#define VP_ERR_DEVICE -2
#define VPCHK(x) do { if ((x) != hipSuccess) return VP_ERR_DEVICE; } while (0)
#define PRE_B 24      // samples per batch: an inter-keyframe interval at 200 Hz / 10 Hz (20 samples) is ONE batch; 24 x 720 doubles = 138 kB of LDS
#define PRE_THREADS 256

namespace {

struct PreArgs {
    int n; const int* start; const double* dt; const double* acc; const double* gyr; const double* acc0; const double* gyr0; const double* ba; const double* bg;
    double nz[4]; double* out; double* jac;
    int s0_, s1_;               // start == nullptr: the one interval [s0_, s1_) (the resident window's IMU slots, vil_window.hpp)
};

// Eigen::Quaternion::toRotationMatrix of a (possibly un-normalised) quaternion w, x, y, z
__device__ __forceinline__ void q_to_R(double w, double x, double y, double z, double* R) {
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// Eigen quaternion * vector (_transformVector): v + w * uv + u x uv with uv = 2 u x v
__device__ __forceinline__ void q_rot(double w, double x, double y, double z, const double* v, double* o) {
    const double ux = 2 * (y * v[2] - z * v[1]), uy = 2 * (z * v[0] - x * v[2]), uz = 2 * (x * v[1] - y * v[0]);
    o[0] = v[0] + w * ux + (y * uz - z * uy); o[1] = v[1] + w * uy + (z * ux - x * uz); o[2] = v[2] + w * uz + (x * uy - y * ux);
}
__device__ __forceinline__ void mm3(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void skew3(const double* v, double* S) { S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0; }

// out(3x3 block br, bc) = sum_q L(3br+a, q) * R(q, 3bc+b)   (transR: R(3bc+b, q)); 15 x 15 row-major operands in LDS
__device__ __forceinline__ void blk33(const double* L, const double* R, bool transR, int br, int bc, double* o) {
#pragma unroll
    for (int e = 0; e < 9; ++e) o[e] = 0.0;
    const double* l0 = L + 45 * br;
#pragma unroll
    for (int q = 0; q < 15; ++q) {
        const double x0 = l0[q], x1 = l0[15 + q], x2 = l0[30 + q];
        double y0, y1, y2;
        if (transR) { y0 = R[(3 * bc) * 15 + q]; y1 = R[(3 * bc + 1) * 15 + q]; y2 = R[(3 * bc + 2) * 15 + q]; }
        else { y0 = R[q * 15 + 3 * bc]; y1 = R[q * 15 + 3 * bc + 1]; y2 = R[q * 15 + 3 * bc + 2]; }
        o[0] += x0 * y0; o[1] += x0 * y1; o[2] += x0 * y2; o[3] += x1 * y0; o[4] += x1 * y1; o[5] += x1 * y2; o[6] += x2 * y0; o[7] += x2 * y1; o[8] += x2 * y2;
    }
}
__device__ __forceinline__ void put33(double* M, int br, int bc, const double* o) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) M[(3 * br + a) * 15 + 3 * bc + b] = o[3 * a + b];
}
__device__ __forceinline__ double shfl_up_d(double v, int o) { return __shfl_up(v, o, 64); }

// The maps (F_s, Q_s): (J, P) -> (F_s J, F_s P F_s^T + Q_s) compose associatively,
//   (F_b, Q_b) o (F_a, Q_a) = (F_b F_a, F_b Q_a F_b^T + Q_b),
// so a batch of PRE_B samples is folded by a pairwise tree (log2 PRE_B levels, every level spread over all lanes as 3 x 3
// register blocks) instead of PRE_B dependent steps; the 10-number state chain is a prefix product of the per-sample
// quaternion increments plus two prefix sums, done by shuffles inside one wave.
__global__ __launch_bounds__(PRE_THREADS) void k_preint(PreArgs A) {
    __shared__ double sF[PRE_B * 225], sQ[PRE_B * 225], sV[PRE_B * 270];
    __shared__ double sJ[225], sC[225], sState[17];
    double* sTmpF = sV;                                        // after Q is formed the V region holds the tree's temporaries:
    double* sTmpT = sV + (PRE_B / 2) * 225;                    // F_b F_a and F_b Q_a of up to PRE_B / 2 pairs
    const int k = blockIdx.x, t = threadIdx.x;
    const int s0 = A.start ? A.start[k] : A.s0_, s1 = A.start ? A.start[k + 1] : A.s1_;
    if (t < 225) { sJ[t] = (t / 15 == t % 15) ? 1.0 : 0.0; sC[t] = 0.0; }
    if (t == 0) {
        // dp(0:3) dv(3:6) dq w x y z (6:10) - - - - - - sum_dt(16)
        for (int q = 0; q < 17; ++q) sState[q] = 0.0;
        sState[6] = 1.0;
    }
    const double ba[3] = {A.ba[3 * k], A.ba[3 * k + 1], A.ba[3 * k + 2]}, bg[3] = {A.bg[3 * k], A.bg[3 * k + 1], A.bg[3 * k + 2]};
    double nd[18];
#pragma unroll
    for (int q = 0; q < 3; ++q) { nd[q] = A.nz[0] * A.nz[0]; nd[3 + q] = A.nz[1] * A.nz[1]; nd[6 + q] = A.nz[0] * A.nz[0]; nd[9 + q] = A.nz[1] * A.nz[1]; nd[12 + q] = A.nz[2] * A.nz[2]; nd[15 + q] = A.nz[3] * A.nz[3]; }
    __syncthreads();
    for (int b0 = s0; b0 < s1; b0 += PRE_B) {
        const int nb = min(PRE_B, s1 - b0);
        for (int e = t; e < nb * 225; e += PRE_THREADS) sF[e] = 0.0;
        for (int e = t; e < nb * 270; e += PRE_THREADS) sV[e] = 0.0;
        __syncthreads();
        // (1) + (2): wave 0, lane = sample of the batch (integration_base.h:63-120, :147-157)
        if (t < 64) {
            const int lane = t, g = b0 + min(lane, nb - 1);
            const bool live = lane < nb;
            const double dt = live ? A.dt[g] : 0.0;
            double a1[3], g1[3], a0[3], g0[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                a1[q] = A.acc[3 * (size_t)g + q]; g1[q] = A.gyr[3 * (size_t)g + q];
                a0[q] = g == s0 ? A.acc0[3 * k + q] : A.acc[3 * (size_t)(g - 1) + q]; g0[q] = g == s0 ? A.gyr0[3 * k + q] : A.gyr[3 * (size_t)(g - 1) + q];
            }
            const double d0[3] = {a0[0] - ba[0], a0[1] - ba[1], a0[2] - ba[2]}, d1[3] = {a1[0] - ba[0], a1[1] - ba[1], a1[2] - ba[2]};
            const double w[3] = {0.5 * (g0[0] + g1[0]) - bg[0], 0.5 * (g0[1] + g1[1]) - bg[1], 0.5 * (g0[2] + g1[2]) - bg[2]};
            const double hx = live ? w[0] * dt / 2 : 0.0, hy = live ? w[1] * dt / 2 : 0.0, hz = live ? w[2] * dt / 2 : 0.0;      // increment Quaterniond(1, hx, hy, hz)
            // inclusive prefix product of the increments (earlier factors on the left)
            double pw = 1.0, px = hx, py = hy, pz = hz;
#pragma unroll
            for (int o = 1; o < PRE_B; o <<= 1) {
                const double ow = shfl_up_d(pw, o), ox = shfl_up_d(px, o), oy = shfl_up_d(py, o), oz = shfl_up_d(pz, o);
                if (lane >= o) {
                    const double nw = ow * pw - ox * px - oy * py - oz * pz, nx = ow * px + ox * pw + oy * pz - oz * py, ny = ow * py + oy * pw + oz * px - ox * pz, nz = ow * pz + oz * pw + ox * py - oy * px;
                    pw = nw; px = nx; py = ny; pz = nz;
                }
            }
            // delta_q at the start of this sample: normalize(q_batch (x) product of the earlier increments)
            double ew = shfl_up_d(pw, 1), ex = shfl_up_d(px, 1), ey = shfl_up_d(py, 1), ez = shfl_up_d(pz, 1);
            if (lane == 0) { ew = 1.0; ex = 0.0; ey = 0.0; ez = 0.0; }
            const double bw = sState[6], bx = sState[7], by = sState[8], bz = sState[9];
            double qw = bw * ew - bx * ex - by * ey - bz * ez, qx = bw * ex + bx * ew + by * ez - bz * ey, qy = bw * ey + by * ew + bz * ex - bx * ez, qz = bw * ez + bz * ew + bx * ey - by * ex;
            const double qn = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
            qw /= qn; qx /= qn; qy /= qn; qz /= qn;
            const double rw = qw - qx * hx - qy * hy - qz * hz, rx = qw * hx + qx + qy * hz - qz * hy, ry = qw * hy + qy + qz * hx - qx * hz, rz = qw * hz + qz + qx * hy - qy * hx;
            double u0[3], u1[3], ua[3];
            q_rot(qw, qx, qy, qz, d0, u0); q_rot(rw, rx, ry, rz, d1, u1);
#pragma unroll
            for (int q = 0; q < 3; ++q) ua[q] = 0.5 * (u0[q] + u1[q]);
            // delta_v at the start of the sample = exclusive prefix sum of ua dt ; delta_p likewise of dv dt + ua dt^2 / 2
            double iv[3] = {ua[0] * dt, ua[1] * dt, ua[2] * dt}, isum = dt;
#pragma unroll
            for (int o = 1; o < PRE_B; o <<= 1) {
                const double c0 = shfl_up_d(iv[0], o), c1 = shfl_up_d(iv[1], o), c2 = shfl_up_d(iv[2], o), c3 = shfl_up_d(isum, o);
                if (lane >= o) { iv[0] += c0; iv[1] += c1; iv[2] += c2; isum += c3; }
            }
            double dv[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) dv[q] = sState[3 + q] + (iv[q] - ua[q] * dt);
            double ip[3] = {dv[0] * dt + 0.5 * ua[0] * dt * dt, dv[1] * dt + 0.5 * ua[1] * dt * dt, dv[2] * dt + 0.5 * ua[2] * dt * dt};
#pragma unroll
            for (int o = 1; o < PRE_B; o <<= 1) {
                const double c0 = shfl_up_d(ip[0], o), c1 = shfl_up_d(ip[1], o), c2 = shfl_up_d(ip[2], o);
                if (lane >= o) { ip[0] += c0; ip[1] += c1; ip[2] += c2; }
            }
            if (live) {
                double Rd[9], Rr[9], Ra0[9], Ra1[9], Rw[9], M0[9], M1[9], M2[9], ImW[9];
                q_to_R(qw, qx, qy, qz, Rd); q_to_R(rw, rx, ry, rz, Rr);
                skew3(d0, Ra0); skew3(d1, Ra1); skew3(w, Rw);
#pragma unroll
                for (int q = 0; q < 9; ++q) ImW[q] = ((q & 3) == 0 ? 1.0 : 0.0) - Rw[q] * dt;
                mm3(Rd, Ra0, M0); mm3(Rr, Ra1, M1); mm3(M1, ImW, M2);
                double* F = sF + 225 * lane; double* V = sV + 270 * lane;
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const int q = 3 * a + b; const double id = a == b ? 1.0 : 0.0;
                        F[a * 15 + b] = id;
                        F[a * 15 + 3 + b] = -0.25 * M0[q] * dt * dt + -0.25 * M2[q] * dt * dt;
                        F[a * 15 + 6 + b] = id * dt;
                        F[a * 15 + 9 + b] = -0.25 * (Rd[q] + Rr[q]) * dt * dt;
                        F[a * 15 + 12 + b] = -0.25 * M1[q] * dt * dt * -dt;
                        F[(3 + a) * 15 + 3 + b] = ImW[q];
                        F[(3 + a) * 15 + 12 + b] = -1.0 * id * dt;
                        F[(6 + a) * 15 + 3 + b] = -0.5 * M0[q] * dt + -0.5 * M2[q] * dt;
                        F[(6 + a) * 15 + 6 + b] = id;
                        F[(6 + a) * 15 + 9 + b] = -0.5 * (Rd[q] + Rr[q]) * dt;
                        F[(6 + a) * 15 + 12 + b] = -0.5 * M1[q] * dt * -dt;
                        F[(9 + a) * 15 + 9 + b] = id;
                        F[(12 + a) * 15 + 12 + b] = id;
                        const double v03 = 0.25 * -M1[q] * dt * dt * 0.5 * dt, v63 = 0.5 * -M1[q] * dt * 0.5 * dt;
                        V[a * 18 + b] = 0.25 * Rd[q] * dt * dt;
                        V[a * 18 + 3 + b] = v03;
                        V[a * 18 + 6 + b] = 0.25 * Rr[q] * dt * dt;
                        V[a * 18 + 9 + b] = v03;
                        V[(3 + a) * 18 + 3 + b] = 0.5 * id * dt;
                        V[(3 + a) * 18 + 9 + b] = 0.5 * id * dt;
                        V[(6 + a) * 18 + b] = 0.5 * Rd[q] * dt;
                        V[(6 + a) * 18 + 3 + b] = v63;
                        V[(6 + a) * 18 + 6 + b] = 0.5 * Rr[q] * dt;
                        V[(6 + a) * 18 + 9 + b] = v63;
                        V[(9 + a) * 18 + 12 + b] = id * dt;
                        V[(12 + a) * 18 + 15 + b] = id * dt;
                    }
            }
            if (lane == nb - 1) {                                              // state after the batch
                const double nr = sqrt(rw * rw + rx * rx + ry * ry + rz * rz);
                const double e0 = sState[0] + ip[0], e1 = sState[1] + ip[1], e2 = sState[2] + ip[2];
                const double f0 = sState[3] + iv[0], f1 = sState[4] + iv[1], f2 = sState[5] + iv[2];
                const double sd = sState[16] + isum;
                sState[0] = e0; sState[1] = e1; sState[2] = e2; sState[3] = f0; sState[4] = f1; sState[5] = f2;
                sState[6] = rw / nr; sState[7] = rx / nr; sState[8] = ry / nr; sState[9] = rz / nr; sState[16] = sd;
            }
        }
        __syncthreads();
        // (3) Q_s = V_s N V_s^T for the whole batch
        for (int e = t; e < nb * 225; e += PRE_THREADS) {
            const int s = e / 225, ij = e - 225 * s, qi = ij / 15, qj = ij - 15 * qi;
            const double* Vi = sV + 270 * s + 18 * qi; const double* Vj = sV + 270 * s + 18 * qj;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;              // three partial sums: a dependent fp64 op costs ~40 cycles here
#pragma unroll
            for (int q = 0; q < 18; q += 3) { a0 += Vi[q] * nd[q] * Vj[q]; a1 += Vi[q + 1] * nd[q + 1] * Vj[q + 1]; a2 += Vi[q + 2] * nd[q + 2] * Vj[q + 2]; }
            sQ[e] = (a0 + a1) + a2;
        }
        __syncthreads();
        // (4) pairwise tree over the batch: slot a <- slot b o slot a  (b = a + st the later samples).  Wide levels run as
        //     3 x 3 register blocks (a third of the LDS reads per multiply-add); narrow levels -- and the final fold into the
        //     running (J, P) -- as one output per lane, which keeps all 256 lanes busy when only one or two pairs are left.
        for (int st = 1; st < nb; st <<= 1) {
            const int npair = (nb - st + 2 * st - 1) / (2 * st);              // slots a = 2 st p with a + st < nb
            const bool wide = npair * 50 >= PRE_THREADS;
            if (wide) {
                for (int task = t; task < npair * 50; task += PRE_THREADS) {
                    const int p = task / 50, r = task - 50 * p, which = r / 25, blk = r - 25 * which, br = blk / 5, bc = blk - 5 * br;
                    const int sa = 2 * st * p, sb = sa + st;
                    double o[9];
                    blk33(sF + 225 * sb, which ? sQ + 225 * sa : sF + 225 * sa, false, br, bc, o);
                    put33((which ? sTmpT : sTmpF) + 225 * p, br, bc, o);
                }
            } else {
                for (int task = t; task < npair * 450; task += PRE_THREADS) {
                    const int p = task / 450, r = task - 450 * p, which = r / 225, ij = r - 225 * which, i = ij / 15, j = ij - 15 * i;
                    const int sa = 2 * st * p, sb = sa + st;
                    const double* L = sF + 225 * sb + 15 * i; const double* R = (which ? sQ + 225 * sa : sF + 225 * sa) + j;
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
                    for (int q = 0; q < 15; q += 3) { a0 += L[q] * R[15 * q]; a1 += L[q + 1] * R[15 * (q + 1)]; a2 += L[q + 2] * R[15 * (q + 2)]; }
                    (which ? sTmpT : sTmpF)[225 * p + ij] = (a0 + a1) + a2;
                }
            }
            __syncthreads();
            if (wide) {
                for (int task = t; task < npair * 25; task += PRE_THREADS) {
                    const int p = task / 25, blk = task - 25 * p, br = blk / 5, bc = blk - 5 * br;
                    const int sa = 2 * st * p, sb = sa + st;
                    double o[9];
                    blk33(sTmpT + 225 * p, sF + 225 * sb, true, br, bc, o);
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) o[3 * a + b] += sQ[225 * sb + (3 * br + a) * 15 + 3 * bc + b];
                    put33(sQ + 225 * sa, br, bc, o);
                }
            } else {
                for (int task = t; task < npair * 225; task += PRE_THREADS) {
                    const int p = task / 225, ij = task - 225 * p, i = ij / 15, j = ij - 15 * i;
                    const int sa = 2 * st * p, sb = sa + st;
                    const double* L = sTmpT + 225 * p + 15 * i; const double* R = sF + 225 * sb + 15 * j;
                    double a0 = sQ[225 * sb + ij], a1 = 0.0, a2 = 0.0;
#pragma unroll
                    for (int q = 0; q < 15; q += 3) { a0 += L[q] * R[q]; a1 += L[q + 1] * R[q + 1]; a2 += L[q + 2] * R[q + 2]; }
                    sQ[225 * sa + ij] = (a0 + a1) + a2;
                }
            }
            // this phase reads only the F of the b slots, so the a slots can take their new F at the same time
            for (int e = t; e < npair * 225; e += PRE_THREADS) { const int p = e / 225; sF[225 * (2 * st * p) + (e - 225 * p)] = sTmpF[e]; }
            __syncthreads();
        }
        // fold the batch into the running jacobian / covariance (:122-123): J <- F J, T = F P, then P <- T F^T + Q
        if (t < 225) {
            const int i = t / 15, j = t - 15 * i;
            const double* L = sF + 15 * i;
            double aj0 = 0.0, aj1 = 0.0, aj2 = 0.0, ap0 = 0.0, ap1 = 0.0, ap2 = 0.0;
#pragma unroll
            for (int q = 0; q < 15; q += 3) {
                const double f0 = L[q], f1 = L[q + 1], f2 = L[q + 2];
                aj0 += f0 * sJ[15 * q + j]; aj1 += f1 * sJ[15 * (q + 1) + j]; aj2 += f2 * sJ[15 * (q + 2) + j];
                ap0 += f0 * sC[15 * q + j]; ap1 += f1 * sC[15 * (q + 1) + j]; ap2 += f2 * sC[15 * (q + 2) + j];
            }
            sTmpF[t] = (aj0 + aj1) + aj2; sTmpT[t] = (ap0 + ap1) + ap2;
        }
        __syncthreads();
        if (t < 225) {
            const int i = t / 15, j = t - 15 * i;
            const double* L = sTmpT + 15 * i; const double* R = sF + 15 * j;
            double a0 = sQ[t], a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int q = 0; q < 15; q += 3) { a0 += L[q] * R[q]; a1 += L[q + 1] * R[q + 1]; a2 += L[q + 2] * R[q + 2]; }
            sC[t] = (a0 + a1) + a2; sJ[t] = sTmpF[t];
        }
        __syncthreads();
    }
    // pack (vilsolve.h: VIL_IMU_CONST layout)
    double* o = A.out + (size_t)287 * k;
    if (t < 3) { o[t] = sState[t]; o[7 + t] = sState[3 + t]; o[10 + t] = ba[t]; o[13 + t] = bg[t]; o[3 + t] = sState[7 + t]; }
    if (t == 3) { o[6] = sState[6]; o[16] = sState[16]; }
    if (t < 9) {
        const int a = t / 3, b = t % 3;
        o[17 + t] = sJ[a * 15 + 9 + b]; o[26 + t] = sJ[a * 15 + 12 + b]; o[35 + t] = sJ[(3 + a) * 15 + 12 + b]; o[44 + t] = sJ[(6 + a) * 15 + 9 + b]; o[53 + t] = sJ[(6 + a) * 15 + 12 + b];
    }
    if (t < 225) { o[62 + t] = sC[t]; if (A.jac) A.jac[(size_t)225 * k + t] = sJ[t]; }
}

}  // namespace

// the resident window (vilsolve.hip): one interval whose samples, first measurement, biases and record all live on the device
void vpre_launch_slot(hipStream_t stream, int ns, const double* dt, const double* acc, const double* gyr, const double* hdr12, const double* noise4, double* rec) {
    PreArgs A;
    A.n = 1; A.start = nullptr; A.s0_ = 0; A.s1_ = ns; A.dt = dt; A.acc = acc; A.gyr = gyr;
    A.acc0 = hdr12; A.gyr0 = hdr12 + 3; A.ba = hdr12 + 6; A.bg = hdr12 + 9;
    for (int q = 0; q < 4; ++q) A.nz[q] = noise4[q];
    A.out = rec; A.jac = nullptr;
    hipLaunchKernelGGL(k_preint, dim3(1), dim3(PRE_THREADS), 0, stream, A);
}

struct vpre_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    char* d_buf = nullptr; size_t cap = 0;
    char* h = nullptr; size_t hcap = 0;              // pinned staging: one H2D in, one D2H back
    bool profiling = false; hipEvent_t ev0 = nullptr, ev1 = nullptr; long long prof_n = 0; double prof_ms = 0.0;
};

extern "C" {

int vpre_create(int32_t device, vpre_ctx** out) {
    if (!out) return VP_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VP_ERR_DEVICE;      // no CPU fallback
    VPCHK(hipSetDevice(device));
    vpre_ctx* c = new vpre_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return VP_ERR_DEVICE; }
    *out = c;
    return VP_OK;
}
void vpre_destroy(vpre_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipFree(c->d_buf); if (c->h) hipHostFree(c->h);
    if (c->ev0) { hipEventDestroy(c->ev0); hipEventDestroy(c->ev1); }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int vpre_integrate(vpre_ctx* c, int32_t n, const int32_t* start, const double* dt, const double* acc, const double* gyr, const double* acc0, const double* gyr0,
                   const double* ba, const double* bg, const double* noise4, double* imu_const, double* jacobian) {
    if (!c || n < 0 || !start || !acc0 || !gyr0 || !ba || !bg || !noise4 || !imu_const) return VP_ERR_INVALID;
    if (n == 0) return VP_OK;
    for (int k = 0; k < n; ++k) if (start[k + 1] < start[k]) return VP_ERR_INVALID;
    if (start[0] != 0) return VP_ERR_INVALID;
    const size_t ns = (size_t)start[n];
    if (ns && (!dt || !acc || !gyr)) return VP_ERR_INVALID;
    VPCHK(hipSetDevice(c->device));
    static const bool timing = VIL_TUNE_ENV("VPRE_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const auto t0 = now();
    // one staging buffer: [start | dt | acc | gyr | acc0 | gyr0 | ba | bg] in, [out | jac] back
    const size_t o_start = 0, o_dt = (4 * (size_t)(n + 1) + 7) & ~(size_t)7, o_acc = o_dt + 8 * ns, o_gyr = o_acc + 24 * ns, o_a0 = o_gyr + 24 * ns, o_g0 = o_a0 + 24 * (size_t)n,
                 o_ba = o_g0 + 24 * (size_t)n, o_bg = o_ba + 24 * (size_t)n, in_bytes = o_bg + 24 * (size_t)n, o_out = in_bytes, o_jac = o_out + 8 * 287 * (size_t)n, total = o_jac + 8 * 225 * (size_t)n;
    if (total > c->cap) { hipFree(c->d_buf); c->d_buf = nullptr; c->cap = 0; VPCHK(hipMalloc(&c->d_buf, 2 * total)); c->cap = 2 * total; }
    if (total > c->hcap) { if (c->h) hipHostFree(c->h); c->h = nullptr; c->hcap = 0; VPCHK(hipHostMalloc((void**)&c->h, 2 * total, hipHostMallocDefault)); c->hcap = 2 * total; }
    memcpy(c->h + o_start, start, 4 * (size_t)(n + 1));
    if (ns) { memcpy(c->h + o_dt, dt, 8 * ns); memcpy(c->h + o_acc, acc, 24 * ns); memcpy(c->h + o_gyr, gyr, 24 * ns); }
    memcpy(c->h + o_a0, acc0, 24 * (size_t)n); memcpy(c->h + o_g0, gyr0, 24 * (size_t)n); memcpy(c->h + o_ba, ba, 24 * (size_t)n); memcpy(c->h + o_bg, bg, 24 * (size_t)n);
    const auto t1 = now();
    VPCHK(hipMemcpyAsync(c->d_buf, c->h, in_bytes, hipMemcpyHostToDevice, c->stream));
    const auto t2 = now();
    PreArgs A;
    A.n = n; A.start = (const int*)(c->d_buf + o_start); A.dt = (const double*)(c->d_buf + o_dt); A.acc = (const double*)(c->d_buf + o_acc); A.gyr = (const double*)(c->d_buf + o_gyr);
    A.acc0 = (const double*)(c->d_buf + o_a0); A.gyr0 = (const double*)(c->d_buf + o_g0); A.ba = (const double*)(c->d_buf + o_ba); A.bg = (const double*)(c->d_buf + o_bg);
    for (int q = 0; q < 4; ++q) A.nz[q] = noise4[q];
    A.out = (double*)(c->d_buf + o_out); A.jac = jacobian ? (double*)(c->d_buf + o_jac) : nullptr;
    if (c->profiling) hipEventRecord(c->ev0, c->stream);
    hipLaunchKernelGGL(k_preint, dim3(n), dim3(PRE_THREADS), 0, c->stream, A);
    if (c->profiling) hipEventRecord(c->ev1, c->stream);
    VPCHK(hipMemcpyAsync(c->h + o_out, c->d_buf + o_out, (jacobian ? total : o_jac) - o_out, hipMemcpyDeviceToHost, c->stream));
    const auto t3 = now();
    VPCHK(hipStreamSynchronize(c->stream));
    const auto t4 = now();
    VPCHK(hipGetLastError());
    memcpy(imu_const, c->h + o_out, 8 * 287 * (size_t)n);
    if (jacobian) memcpy(jacobian, c->h + o_jac, 8 * 225 * (size_t)n);
    if (timing) fprintf(stderr, "vpre: stage %.1f h2d %.1f launch+d2h %.1f sync %.1f copy-out %.1f us\n", us(t0, t1), us(t1, t2), us(t2, t3), us(t3, t4), us(t4, now()));
    if (c->profiling) { float ms = 0.f; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) { c->prof_ms += ms; c->prof_n++; } }
    return VP_OK;
}

int vpre_profile_enable(vpre_ctx* c, int32_t enable) {
    if (!c) return VP_ERR_INVALID;
    VPCHK(hipSetDevice(c->device));
    if (enable && !c->ev0) { VPCHK(hipEventCreate(&c->ev0)); VPCHK(hipEventCreate(&c->ev1)); }
    c->profiling = enable != 0;
    return VP_OK;
}
int vpre_profile_read(vpre_ctx* c, int64_t* launches, double* total_ms) {
    if (!c || !launches || !total_ms) return VP_ERR_INVALID;
    *launches = c->prof_n; *total_ms = c->prof_ms; c->prof_n = 0; c->prof_ms = 0.0;
    return VP_OK;
}

}  // extern "C"
