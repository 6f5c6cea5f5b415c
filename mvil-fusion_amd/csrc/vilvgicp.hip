// vilvgicp.hip -- voxelised GICP scan-to-scan registration on gfx950 behind include/vilvgicp.h.
//
// What the reference does per LiDAR scan (estimator.cpp:269-300 -> fast_gicp FastVGICP, OpenMP on the host):
//   voxel map of the target (once per scan pair)  |  per LM iteration: correspondences by voxel look-up of every transformed
//   source point, a 3x3 inverse per correspondence, then a 6x6 / 6x1 / scalar reduction over all correspondences.
// MI355X-first layout of the same arithmetic:
//   * target voxel map: built on the host in the reference's sequential order (it is summed once per scan and must not
//     depend on atomics), stored as an open-addressing table of packed 63-bit voxel keys + SoA voxel records in HBM;
//   * ONE kernel per linearisation: thread = (source point, neighbour offset) does transform, hash probe, RCR inverse,
//     stores the correspondence (voxel id + Mahalanobis matrix, what compute_error reuses) and its 28 partial sums,
//     wave64 butterflies + LDS fold them per workgroup, a second tiny kernel adds the workgroup partials in fixed order
//     (deterministic, no fp64 atomics);
//   * compute_error: thread = stored correspondence, one sum;
//   * the 6x6 LM / GN step logic (lsq_registration_impl.hpp:88-165) runs on the host between launches: it is a handful of flops.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <unordered_map>
#include <vector>

#include "../../include/vilvgicp.h"
#include "vil_coop.hpp"
#include "vil_knn.hpp"
#include "vil_math.hpp"
#include "vil_tuning.hpp"

#define VG_OK 0
#define VG_ERR_INVALID -1
#define VG_ERR_DEVICE -2
#define VG_ERR_NONFINITE -3
#define VG_THREADS 256
#define VGA_THREADS 512     // k_vgicp_align: one slot per thread on the usual scan (a pass is a chain of dependent gathers per slot; two waves per SIMD hide part of it)
#define VGCHK(x) do { if ((x) != hipSuccess) return VG_ERR_DEVICE; } while (0)

namespace {

struct Iso { double m[12]; };                       // rows of [R | t]
struct VoxTab { const long long* keys; const int* slot_vox; int mask; const int* num; const double* mean; const double* cov; const double* wsq; };   // mean 3 x nv, cov 9 x nv (SoA); wsq = sqrt(num), rounded once on the host

using vknn::pack_key; using vknn::hash_key;

__device__ __forceinline__ double wave_sum64(double v) { return vd::wave_total(v); }   // DPP rotate-and-add, no LDS round trips

__device__ __forceinline__ void inv3(const double* a, double* o) {
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double id = vd::rcp_nr(a[0] * c0 + a[1] * c1 + a[2] * c2);      // (1-2 ulp; the IEEE divide is a 1 k-cycle chain in the middle of every slot)
    o[0] = c0 * id; o[1] = (a[2] * a[7] - a[1] * a[8]) * id; o[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    o[3] = c1 * id; o[4] = (a[0] * a[8] - a[2] * a[6]) * id; o[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    o[6] = c2 * id; o[7] = (a[1] * a[6] - a[0] * a[7]) * id; o[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

// per-workgroup fold of NV per-thread values: wave butterflies, then the 4 wave results through LDS
template <int NV>
__device__ __forceinline__ void block_fold(double* v, double* part /* gridDim.x x NV */) {
    __shared__ double sm[4][NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (NV > 2) {                                        // all values in one pass (vil_math.hpp wave_fold): lanes 0..15 end with the totals
        double f0, f1;
        vd::wave_fold<NV>(v, f0, f1);
        if (lane < 16) { const int q = vd::fold_slot(lane); if (q < NV) sm[wave][q] = f0; if (q + 16 < NV) sm[wave][q + 16] = f1; }
    } else {
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] = wave_sum64(v[q]);
        if (lane == 0) for (int q = 0; q < NV; ++q) sm[wave][q] = v[q];
    }
    __syncthreads();
    if (threadIdx.x < NV) part[(size_t)blockIdx.x * NV + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// floor(x / res - 0.5) of fast_vgicp_voxel.hpp:140 without the IEEE divide on every slot's dependent chain: x * (1 / res) is within a
// few ulp of the quotient, so the floor can only differ when that product sits within 1e-9 (relative) of a cell boundary -- those
// lanes (about one in 1e8) take the exact expression
__device__ __forceinline__ int vox_coord(double x, double res, double ires) {
    const double q = x * ires - 0.5, f = floor(q);
    const double m = fmin(q - f, f + 1.0 - q);
    if (m <= 1e-9 * (fabs(q) + 1.0)) return (int)floor(x / res - 0.5);
    return (int)f;
}
// FastVGICP::update_correspondences + linearize (fast_vgicp_impl.hpp:73-170) for ONE (source point, offset) slot: the
// correspondence (voxel id + Mahalanobis matrix) is stored, the 29 sums are ADDED to acc (21 H upper | 6 b | error | count)
__device__ __forceinline__ void vgicp_lin_slot(int tid, int noff, const float* __restrict__ sxyz, const double* __restrict__ scov, const Iso& T, double res, const VoxTab& V,
                                               int* __restrict__ c_vox, double* __restrict__ c_M, double* acc, int want_H) {
        const int i = tid / noff, o = tid - i * noff;
        const double ires = vd::rcp_nr(res);
        const double ax = (double)sxyz[3 * i], ay = (double)sxyz[3 * i + 1], az = (double)sxyz[3 * i + 2];
        const double tx = T.m[0] * ax + T.m[1] * ay + T.m[2] * az + T.m[3];
        const double ty = T.m[4] * ax + T.m[5] * ay + T.m[6] * az + T.m[7];
        const double tz = T.m[8] * ax + T.m[9] * ay + T.m[10] * az + T.m[11];
        int kx = vox_coord(tx, res, ires), ky = vox_coord(ty, res, ires), kz = vox_coord(tz, res, ires);
        if (noff == 7) { const int d = (o + 1) >> 1, s = (o & 1) ? 1 : -1; if (o) { if (d == 1) kx += s; else if (d == 2) ky += s; else kz += s; } }   // (0) (+x -x) (+y -y) (+z -z)
        else if (noff == 27) { kx += o / 9 - 1; ky += (o / 3) % 3 - 1; kz += o % 3 - 1; }
        const long long key = pack_key(kx, ky, kz);
        // the source covariance does not depend on the probe: its loads go out before the dependent chain key -> voxel -> mean / cov
        double ca[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) ca[q] = scov[(size_t)9 * i + q];
        int v = -1;
        for (unsigned h = hash_key(key) & V.mask;; h = (h + 1) & V.mask) {      // linear probing; the table is never full
            const long long k = V.keys[h];
            const int sv = V.slot_vox[h];                                        // (same index as the key: one round trip, not two)
            if (k == key) { v = sv; break; }
            if (k < 0) break;
        }
        c_vox[tid] = v;
        if (v >= 0) {
            double cb[9], RC[9], RCR[9], M[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) cb[q] = V.cov[(size_t)9 * v + q];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) RC[3 * r + q] = T.m[4 * r] * ca[q] + T.m[4 * r + 1] * ca[3 + q] + T.m[4 * r + 2] * ca[6 + q];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) RCR[3 * r + q] = cb[3 * r + q] + RC[3 * r] * T.m[4 * q] + RC[3 * r + 1] * T.m[4 * q + 1] + RC[3 * r + 2] * T.m[4 * q + 2];
            inv3(RCR, M);
#pragma unroll
            for (int q = 0; q < 9; ++q) c_M[(size_t)9 * tid + q] = M[q];
            const double e0 = V.mean[(size_t)3 * v] - tx, e1 = V.mean[(size_t)3 * v + 1] - ty, e2 = V.mean[(size_t)3 * v + 2] - tz;
            const double w = V.wsq[v];
            const double m0 = M[0] * e0 + M[1] * e1 + M[2] * e2, m1 = M[3] * e0 + M[4] * e1 + M[5] * e2, m2 = M[6] * e0 + M[7] * e1 + M[8] * e2;
            acc[27] += w * (e0 * m0 + e1 * m1 + e2 * m2); acc[28] += 1.0;
            if (want_H) {
                // J = [skew(ta) | -I] ; H += w J^T M J (upper triangle, 21 values) ; b += w J^T M e
                const double J[18] = {0.0, -tz, ty, -1.0, 0.0, 0.0, tz, 0.0, -tx, 0.0, -1.0, 0.0, -ty, tx, 0.0, 0.0, 0.0, -1.0};
                double MJ[18];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int q = 0; q < 6; ++q) MJ[6 * r + q] = M[3 * r] * J[q] + M[3 * r + 1] * J[6 + q] + M[3 * r + 2] * J[12 + q];
                int idx = 0;
#pragma unroll
                for (int p = 0; p < 6; ++p) {
#pragma unroll
                    for (int q = p; q < 6; ++q) acc[idx++] += w * (J[p] * MJ[q] + J[6 + p] * MJ[6 + q] + J[12 + p] * MJ[12 + q]);
                    acc[21 + p] += w * (J[p] * m0 + J[6 + p] * m1 + J[12 + p] * m2);
                }
            }
        }
}

// one thread per (source point, offset)
__global__ __launch_bounds__(VG_THREADS) void k_vgicp_lin(int n, int noff, const float* __restrict__ sxyz, const double* __restrict__ scov, Iso T, double res, VoxTab V,
                                                          int* __restrict__ c_vox, double* __restrict__ c_M, double* __restrict__ part, int want_H) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[29];                                   // 21 H (upper) | 6 b | error | number of correspondences
#pragma unroll
    for (int q = 0; q < 29; ++q) acc[q] = 0.0;
    if (tid < n * noff) vgicp_lin_slot(tid, noff, sxyz, scov, T, res, V, c_vox, c_M, acc, want_H);
    block_fold<29>(acc, part);
}

// FastVGICP::compute_error (fast_vgicp_impl.hpp:173-196): stored correspondences and Mahalanobis matrices, new transform
__device__ __forceinline__ double vgicp_err_slot(int tid, int noff, const float* __restrict__ sxyz, const Iso& T, const VoxTab& V, const int* __restrict__ c_vox, const double* __restrict__ c_M) {
    const int v = c_vox[tid];
    if (v < 0) return 0.0;
    const int i = tid / noff;
    const double ax = (double)sxyz[3 * i], ay = (double)sxyz[3 * i + 1], az = (double)sxyz[3 * i + 2];
    const double e0 = V.mean[(size_t)3 * v] - (T.m[0] * ax + T.m[1] * ay + T.m[2] * az + T.m[3]);
    const double e1 = V.mean[(size_t)3 * v + 1] - (T.m[4] * ax + T.m[5] * ay + T.m[6] * az + T.m[7]);
    const double e2 = V.mean[(size_t)3 * v + 2] - (T.m[8] * ax + T.m[9] * ay + T.m[10] * az + T.m[11]);
    const double* M = c_M + (size_t)9 * tid;
    const double m0 = M[0] * e0 + M[1] * e1 + M[2] * e2, m1 = M[3] * e0 + M[4] * e1 + M[5] * e2, m2 = M[6] * e0 + M[7] * e1 + M[8] * e2;
    return V.wsq[v] * (e0 * m0 + e1 * m1 + e2 * m2);
}
__global__ __launch_bounds__(VG_THREADS) void k_vgicp_err(int ncorr_slots, int noff, const float* __restrict__ sxyz, Iso T, VoxTab V, const int* __restrict__ c_vox, const double* __restrict__ c_M, double* __restrict__ part) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[1] = {0.0};
    if (tid < ncorr_slots) acc[0] = vgicp_err_slot(tid, noff, sxyz, T, V, c_vox, c_M);
    block_fold<1>(acc, part);
}

// fixed-order sum of the workgroup partials: value q is summed by the 8 lanes 8q..8q+7 (strided partial sums, then a
// 3-step butterfly inside the group) -- one pass, no barrier, same bits on every run
template <int NV>
// hout / hseq (or null): the sums also go to pinned, device-mapped host memory, followed by the call's tag -- the host polls that word
// instead of a copy + stream synchronisation
__global__ __launch_bounds__(VG_THREADS) void k_vgicp_sum(int nblk, const double* __restrict__ part, double* __restrict__ out, double* hout, int* hseq, int tag) {
    const int t = threadIdx.x, q = t >> 3, r = t & 7;
    double s = 0.0;
    if (q < NV) for (int b = r; b < nblk; b += 8) s += part[(size_t)b * NV + q];
    s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
    if (q < NV && r == 0) { out[q] = s; if (hout) { hout[q] = s; __threadfence_system(); } }
    if (hout) {
        __syncthreads();
        if (t == 0) __hip_atomic_store(hseq, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// FastGICP::calculate_covariances (fast_gicp_impl.hpp:241-300), k <= 20, RegularizationMethod::PLANE.
// Exact k-nearest-neighbour search by tiles: thread = query point, the candidate points stream through LDS 256 at a time,
// the running k best (float squared distance, index) live in registers as a sorted list updated by an unrolled,
// branch-free insertion (static register indices: no scratch).  Same float arithmetic and tie order (smaller index
// first) as the oracle, so both select the same neighbours.  Then mean / covariance in fp64 and
// U diag(1, 1, 1e-3) V^T = I - (1 - 1e-3) n n^T with n the eigenvector of the smallest eigenvalue (cyclic Jacobi, 3 x 3).
using namespace vknn;

// mean / covariance of the k neighbours in fp64, then U diag(1, 1, 1e-3) V^T = I - (1 - 1e-3) n n^T (cyclic Jacobi, 3 x 3)
__device__ __forceinline__ void knn_plane_cov(const KnnList& L, const float* __restrict__ xyz, int k, double* __restrict__ o) {
    double mx = 0, my = 0, mz = 0;
#pragma unroll
    for (int m = 0; m < KNN_MAX; ++m) if (m < k && L.bi[m] != 0x7fffffff) { mx += (double)xyz[3 * L.bi[m]]; my += (double)xyz[3 * L.bi[m] + 1]; mz += (double)xyz[3 * L.bi[m] + 2]; }
    mx /= k; my /= k; mz /= k;
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < KNN_MAX; ++m) if (m < k && L.bi[m] != 0x7fffffff) {
        const double c0 = (double)xyz[3 * L.bi[m]] - mx, c1 = (double)xyz[3 * L.bi[m] + 1] - my, c2 = (double)xyz[3 * L.bi[m] + 2] - mz;
        A[0] += c0 * c0; A[1] += c0 * c1; A[2] += c0 * c2; A[3] += c1 * c0; A[4] += c1 * c1; A[5] += c1 * c2; A[6] += c2 * c0; A[7] += c2 * c1; A[8] += c2 * c2;
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) A[q] /= k;
    double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off <= 1e-40 * (A[0] * A[0] + A[4] * A[4] + A[8] * A[8]) || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[3 * p + q];
                if (apq != 0.0) {
                    const double tau = (A[3 * q + q] - A[3 * p + p]) / (2.0 * apq);
                    const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    const double c = 1.0 / sqrt(1.0 + t * t), sn = t * c;
#pragma unroll
                    for (int r = 0; r < 3; ++r) { const double akp = A[3 * r + p], akq = A[3 * r + q]; A[3 * r + p] = c * akp - sn * akq; A[3 * r + q] = sn * akp + c * akq; }
#pragma unroll
                    for (int r = 0; r < 3; ++r) { const double apk = A[3 * p + r], aqk = A[3 * q + r]; A[3 * p + r] = c * apk - sn * aqk; A[3 * q + r] = sn * apk + c * aqk; }
#pragma unroll
                    for (int r = 0; r < 3; ++r) { const double vkp = V[3 * r + p], vkq = V[3 * r + q]; V[3 * r + p] = c * vkp - sn * vkq; V[3 * r + q] = sn * vkp + c * vkq; }
                }
            }
    }
    double n0 = V[0], n1 = V[3], n2 = V[6], lmin = A[0];
    if (A[4] < lmin) { lmin = A[4]; n0 = V[1]; n1 = V[4]; n2 = V[7]; }
    if (A[8] < lmin) { n0 = V[2]; n1 = V[5]; n2 = V[8]; }
    const double f = 1.0 - 1e-3;
    o[0] = 1.0 - f * n0 * n0; o[1] = -f * n0 * n1; o[2] = -f * n0 * n2;
    o[3] = -f * n1 * n0; o[4] = 1.0 - f * n1 * n1; o[5] = -f * n1 * n2;
    o[6] = -f * n2 * n0; o[7] = -f * n2 * n1; o[8] = 1.0 - f * n2 * n2;
}

// small clouds: every candidate streams through LDS 256 at a time (O(n^2))
__global__ __launch_bounds__(VG_THREADS) void k_knn_cov(int n, const float* __restrict__ xyz, int k, double* __restrict__ cov9) {
    __shared__ float sx[VG_THREADS], sy[VG_THREADS], sz[VG_THREADS];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const float qx = live ? xyz[3 * i] : 0.f, qy = live ? xyz[3 * i + 1] : 0.f, qz = live ? xyz[3 * i + 2] : 0.f;
    KnnList L; knn_init(L);
    for (int t0 = 0; t0 < n; t0 += VG_THREADS) {
        const int j0 = t0 + threadIdx.x;
        __syncthreads();
        if (j0 < n) { sx[threadIdx.x] = xyz[3 * j0]; sy[threadIdx.x] = xyz[3 * j0 + 1]; sz[threadIdx.x] = xyz[3 * j0 + 2]; }
        __syncthreads();
        const int cnt = min(VG_THREADS, n - t0);
        for (int jj = 0; jj < cnt; ++jj) knn_insert(L, sqdist_nofma(qx, qy, qz, sx[jj], sy[jj], sz[jj]), t0 + jj);
    }
    if (live) knn_plane_cov(L, xyz, k, cov9 + (size_t)9 * i);
}

// larger clouds: exact grid search with one WAVE per query point (vil_knn.hpp) writing the k neighbour indices in (distance,
// index) order, then one thread per point for the fp64 mean / covariance / regularisation
#define VG_QPB 4
__global__ __launch_bounds__(64 * VG_QPB) void k_knn_wave(int n, const float* __restrict__ xyz, GridTab G, const int* __restrict__ order, const float* __restrict__ cxyz, int k, int* __restrict__ nn) {
    __shared__ int wl_all[VG_QPB * KNN_WL_CAP];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pos = blockIdx.x * VG_QPB + wave;                              // queries in cell order: neighbouring waves share cells in L2
    if (pos >= n) return;
    unsigned long long best;
    knn_wave_query(best, cxyz[3 * pos], cxyz[3 * pos + 1], cxyz[3 * pos + 2], min(k, n), n, G, order, cxyz, wl_all + wave * KNN_WL_CAP, 3.0e38f, 1);
    if (lane < KNN_MAX) nn[(size_t)KNN_MAX * order[pos] + lane] = lane < min(k, n) ? (int)(unsigned)best : 0x7fffffff;
}
__global__ __launch_bounds__(VG_THREADS) void k_knn_fit(int n, const float* __restrict__ xyz, const int* __restrict__ nn, int k, double* __restrict__ cov9) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    KnnList L; knn_init(L);
#pragma unroll
    for (int m = 0; m < KNN_MAX; ++m) L.bi[m] = nn[(size_t)KNN_MAX * i + m];
    knn_plane_cov(L, xyz, k, cov9 + (size_t)9 * i);
}

struct HostKeyHash { size_t operator()(long long k) const { return (size_t)hash_key(k) * 2654435761u ^ (size_t)(k >> 17); } };

}  // namespace

struct vgicp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // target voxel map
    double res = 1.0; int nvox = 0, cap = 0;
    long long* d_keys = nullptr; int* d_slot = nullptr; int* d_num = nullptr; double* d_mean = nullptr; double* d_cov = nullptr; double* d_wsq = nullptr;
    // source
    int n = 0; float* d_sxyz = nullptr; double* d_scov = nullptr;
    // correspondences of the last linearisation
    int noff = 1, slots = 0, slots_cap = 0; int* d_cvox = nullptr; double* d_cM = nullptr;
    char* h_lin = nullptr; void* d_lin = nullptr; int lin_tag = 0;          // pinned + mapped: 32 doubles | tag word of vgicp_linearize / vgicp_compute_error
    int slots_cap2 = 0; int* d_cvox2 = nullptr; double* d_cM2 = nullptr;            // second correspondence cache of k_vgicp_align (speculative linearisation)
    double* d_part = nullptr; int part_cap = 0; double* d_out = nullptr; double* h_out = nullptr;
    bool linearized = false;
    bool profiling = false; hipEvent_t ev0 = nullptr, ev1 = nullptr; long long prof_n = 0; double prof_ms = 0.0;
    void* d_coop = nullptr; void* d_aout = nullptr; void* h_aout = nullptr; int coop_epoch = 0;       // one-launch alignment (k_vgicp_align)
    // neighbour search of the covariance estimation
    vknn::GridBuild gb; float grid_h = 1.0f; int grid_min = 4096; int coop_cap = -1; int* d_nn = nullptr; size_t nn_cap = 0;
};

static void free_target(vgicp_ctx* c) { hipFree(c->d_keys); hipFree(c->d_slot); hipFree(c->d_num); hipFree(c->d_mean); hipFree(c->d_cov); hipFree(c->d_wsq); c->d_wsq = nullptr; c->d_keys = nullptr; c->d_slot = nullptr; c->d_num = nullptr; c->d_mean = nullptr; c->d_cov = nullptr; c->nvox = 0; }
static void free_source(vgicp_ctx* c) { hipFree(c->d_sxyz); hipFree(c->d_scov); c->d_sxyz = nullptr; c->d_scov = nullptr; c->n = 0; }

static Iso to_iso(const double* T) { Iso r; for (int q = 0; q < 12; ++q) r.m[q] = T[q]; return r; }
static VoxTab tab(const vgicp_ctx* c) { return VoxTab{c->d_keys, c->d_slot, c->cap - 1, c->d_num, c->d_mean, c->d_cov, c->d_wsq}; }

// device covariances of a device-resident cloud (d_xyz) into d_cov (n x 9)
static int covariances_dev(vgicp_ctx* c, int n, const float* d_xyz, int k, double* d_cov) {
    if (k < 1 || k > KNN_MAX) return VG_ERR_INVALID;
    const int nblk = (n + VG_THREADS - 1) / VG_THREADS;
    const int grid_min = c->grid_min;                            // below this the O(n^2) tiled search is faster than building the grid (vgicp_set_knn_grid)
    if (n < grid_min) {
        hipLaunchKernelGGL(k_knn_cov, dim3(nblk), dim3(VG_THREADS), 0, c->stream, n, d_xyz, k, d_cov);
    } else {
        if ((size_t)n * KNN_MAX * 4 > c->nn_cap) { hipFree(c->d_nn); c->d_nn = nullptr; c->nn_cap = 0; VGCHK(hipMalloc(&c->d_nn, (size_t)n * KNN_MAX * 6)); c->nn_cap = (size_t)n * KNN_MAX * 6; }
        VGCHK(vknn::grid_build_adaptive(c->gb, n, d_xyz, 3, c->grid_h, 8.0, c->stream));
        hipLaunchKernelGGL(k_knn_wave, dim3((n + VG_QPB - 1) / VG_QPB), dim3(64 * VG_QPB), 0, c->stream, n, d_xyz, c->gb.G, c->gb.order, c->gb.cxyz, k, c->d_nn);
        hipLaunchKernelGGL(k_knn_fit, dim3(nblk), dim3(VG_THREADS), 0, c->stream, n, d_xyz, c->d_nn, k, d_cov);
    }
    VGCHK(hipStreamSynchronize(c->stream));
    VGCHK(hipGetLastError());
    return VG_OK;
}

// the fixed-order sum of the workgroup partials, result in c->h_out: polled from pinned memory (profiling / VIL_NO_POLL: copy + synchronise)
template <int NV>
static hipError_t sum_and_fetch(vgicp_ctx* c, int nblk) {
    if (!c->h_lin && !c->lin_tag && !getenv("VIL_NO_POLL")) {
        if (hipHostMalloc((void**)&c->h_lin, 8 * 32 + 64, hipHostMallocMapped) == hipSuccess) {
            memset(c->h_lin, 0, 8 * 32 + 64);
            if (hipHostGetDevicePointer(&c->d_lin, c->h_lin, 0) != hipSuccess) { hipHostFree(c->h_lin); c->h_lin = nullptr; c->d_lin = nullptr; }
        } else c->h_lin = nullptr;
        c->lin_tag = 1;
    }
    if (c->h_lin && !c->profiling) {
        const int tag = ++c->lin_tag;
        volatile int* seq = (volatile int*)(c->h_lin + 8 * 32);
        hipLaunchKernelGGL((k_vgicp_sum<NV>), dim3(1), dim3(VG_THREADS), 0, c->stream, nblk, c->d_part, c->d_out, (double*)c->d_lin, (int*)((char*)c->d_lin + 8 * 32), tag);
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (long spin = 1;; ++spin) {
            if (*seq == tag) { seen = true; break; }
            if ((spin & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
        }
        if (seen) { std::atomic_thread_fence(std::memory_order_acquire); memcpy(c->h_out, c->h_lin, 8 * NV); return hipSuccess; }
        const hipError_t e = hipStreamSynchronize(c->stream);       // surfaces a fault; a merely slow device gets the copy below
        if (e != hipSuccess) return e;
    } else hipLaunchKernelGGL((k_vgicp_sum<NV>), dim3(1), dim3(VG_THREADS), 0, c->stream, nblk, c->d_part, c->d_out, (double*)nullptr, (int*)nullptr, 0);
    hipError_t e = hipMemcpyAsync(c->h_out, c->d_out, 8 * NV, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(c->stream);
}

extern "C" {

int vgicp_covariances(vgicp_ctx* c, int32_t n, const float* xyz, int32_t k, double* out) {
    if (!c || n <= 0 || !xyz || !out) return VG_ERR_INVALID;
    VGCHK(hipSetDevice(c->device));
    float* dx = nullptr; double* dc = nullptr;
    VGCHK(hipMalloc(&dx, 12 * (size_t)n));
    if (hipMalloc(&dc, 72 * (size_t)n) != hipSuccess) { hipFree(dx); return VG_ERR_DEVICE; }
    int st = hipMemcpy(dx, xyz, 12 * (size_t)n, hipMemcpyHostToDevice) == hipSuccess ? covariances_dev(c, n, dx, k, dc) : VG_ERR_DEVICE;
    if (st == VG_OK && hipMemcpy(out, dc, 72 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) st = VG_ERR_DEVICE;
    hipFree(dx); hipFree(dc);
    return st;
}

int vgicp_create(int32_t device, vgicp_ctx** out) {
    if (!out) return VG_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VG_ERR_DEVICE;      // no CPU fallback
    VGCHK(hipSetDevice(device));
    vgicp_ctx* c = new vgicp_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return VG_ERR_DEVICE; }
    if (hipMalloc(&c->d_out, 8 * 32) != hipSuccess || hipHostMalloc(&c->h_out, 8 * 32) != hipSuccess) { delete c; return VG_ERR_DEVICE; }
    *out = c;
    return VG_OK;
}
void vgicp_destroy(vgicp_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    free_target(c); free_source(c);
    hipFree(c->d_cvox); hipFree(c->d_cM); hipFree(c->d_cvox2); hipFree(c->d_cM2); hipFree(c->d_part); hipFree(c->d_out); if (c->h_out) hipHostFree(c->h_out);
    hipFree(c->gb.ws); hipFree(c->d_nn);
    if (c->d_coop) hipFree(c->d_coop); if (c->h_aout) hipHostFree(c->h_aout); if (c->h_lin) hipHostFree(c->h_lin);
    if (c->ev0) { hipEventDestroy(c->ev0); hipEventDestroy(c->ev1); }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}
void vgicp_default_options(vgicp_options* o) {
    if (!o) return;
    o->neighbor_mode = VGICP_DIRECT1; o->optimizer = VGICP_LM; o->max_iterations = 64; o->lm_max_iterations = 10;
    o->rotation_epsilon = 2e-3; o->transformation_epsilon = 5e-4; o->lm_init_lambda_factor = 1e-9;
}

int vgicp_set_target(vgicp_ctx* c, int32_t n, const float* xyz, const double* cov9_in, double resolution) {
    if (!c || n <= 0 || !xyz || !(resolution > 0.0)) return VG_ERR_INVALID;
    VGCHK(hipSetDevice(c->device));
    std::vector<double> own;
    const double* cov9 = cov9_in;
    if (!cov9) { own.resize(9 * (size_t)n); const int st = vgicp_covariances(c, n, xyz, KNN_MAX, own.data()); if (st != VG_OK) return st; cov9 = own.data(); }
    // GaussianVoxelMap::create_voxelmap (fast_vgicp_voxel.hpp:128-159), ADDITIVE voxels: sequential sums in point order
    std::unordered_map<long long, int, HostKeyHash> index;
    std::vector<long long> keys; std::vector<int> num; std::vector<double> mean, cov;
    index.reserve((size_t)n);
    for (int i = 0; i < n; ++i) {
        const double p[3] = {(double)xyz[3 * i], (double)xyz[3 * i + 1], (double)xyz[3 * i + 2]};
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) return VG_ERR_NONFINITE;
        const long long k = pack_key((int)std::floor(p[0] / resolution - 0.5), (int)std::floor(p[1] / resolution - 0.5), (int)std::floor(p[2] / resolution - 0.5));
        auto it = index.find(k);
        int v;
        if (it == index.end()) { v = (int)keys.size(); index.emplace(k, v); keys.push_back(k); num.push_back(0); mean.insert(mean.end(), 3, 0.0); cov.insert(cov.end(), 9, 0.0); }
        else v = it->second;
        num[v]++;
        for (int q = 0; q < 3; ++q) mean[3 * (size_t)v + q] += p[q];
        for (int q = 0; q < 9; ++q) cov[9 * (size_t)v + q] += cov9[9 * (size_t)i + q];
    }
    const int nv = (int)keys.size();
    for (int v = 0; v < nv; ++v) { for (int q = 0; q < 3; ++q) mean[3 * (size_t)v + q] /= num[v]; for (int q = 0; q < 9; ++q) cov[9 * (size_t)v + q] /= num[v]; }   // finalize()
    int cap = 64; while (cap < 2 * nv) cap <<= 1;
    std::vector<long long> tkeys((size_t)cap, -1); std::vector<int> tslot((size_t)cap, -1);
    for (int v = 0; v < nv; ++v) { unsigned h = hash_key(keys[v]) & (cap - 1); while (tkeys[h] >= 0) h = (h + 1) & (cap - 1); tkeys[h] = keys[v]; tslot[h] = v; }
    free_target(c);
    VGCHK(hipMalloc(&c->d_keys, 8 * (size_t)cap)); VGCHK(hipMalloc(&c->d_slot, 4 * (size_t)cap)); VGCHK(hipMalloc(&c->d_num, 4 * (size_t)nv));
    VGCHK(hipMalloc(&c->d_mean, 8 * 3 * (size_t)nv)); VGCHK(hipMalloc(&c->d_cov, 8 * 9 * (size_t)nv));
    VGCHK(hipMemcpy(c->d_keys, tkeys.data(), 8 * (size_t)cap, hipMemcpyHostToDevice)); VGCHK(hipMemcpy(c->d_slot, tslot.data(), 4 * (size_t)cap, hipMemcpyHostToDevice));
    { std::vector<double> wsq((size_t)nv); for (int v = 0; v < nv; ++v) wsq[v] = std::sqrt((double)num[v]); VGCHK(hipMalloc(&c->d_wsq, 8 * (size_t)nv)); VGCHK(hipMemcpy(c->d_wsq, wsq.data(), 8 * (size_t)nv, hipMemcpyHostToDevice)); }
    VGCHK(hipMemcpy(c->d_num, num.data(), 4 * (size_t)nv, hipMemcpyHostToDevice)); VGCHK(hipMemcpy(c->d_mean, mean.data(), 8 * 3 * (size_t)nv, hipMemcpyHostToDevice));
    VGCHK(hipMemcpy(c->d_cov, cov.data(), 8 * 9 * (size_t)nv, hipMemcpyHostToDevice));
    c->res = resolution; c->nvox = nv; c->cap = cap; c->linearized = false;
    return VG_OK;
}

int vgicp_set_source(vgicp_ctx* c, int32_t n, const float* xyz, const double* cov9) {
    if (!c || n <= 0 || !xyz) return VG_ERR_INVALID;
    VGCHK(hipSetDevice(c->device));
    free_source(c);
    VGCHK(hipMalloc(&c->d_sxyz, 4 * 3 * (size_t)n)); VGCHK(hipMalloc(&c->d_scov, 8 * 9 * (size_t)n));
    VGCHK(hipMemcpy(c->d_sxyz, xyz, 4 * 3 * (size_t)n, hipMemcpyHostToDevice));
    if (cov9) VGCHK(hipMemcpy(c->d_scov, cov9, 8 * 9 * (size_t)n, hipMemcpyHostToDevice));
    else { const int st = covariances_dev(c, n, c->d_sxyz, KNN_MAX, c->d_scov); if (st != VG_OK) return st; }      // source covariances never leave the device
    c->n = n; c->linearized = false;
    return VG_OK;
}

int vgicp_linearize(vgicp_ctx* c, const double* T, int32_t mode, double* err, double* H, double* b, int32_t* n_corr) {
    if (!c || !T || !err || !c->nvox || !c->n || (mode != VGICP_DIRECT1 && mode != VGICP_DIRECT7 && mode != VGICP_DIRECT27)) return VG_ERR_INVALID;
    VGCHK(hipSetDevice(c->device));
    const int slots = c->n * mode, nblk = (slots + VG_THREADS - 1) / VG_THREADS;
    if (slots > c->slots_cap) { hipFree(c->d_cvox); hipFree(c->d_cM); c->d_cvox = nullptr; c->d_cM = nullptr; c->slots_cap = 0; VGCHK(hipMalloc(&c->d_cvox, 4 * (size_t)slots)); VGCHK(hipMalloc(&c->d_cM, 8 * 9 * (size_t)slots)); c->slots_cap = slots; }
    if (nblk > c->part_cap) { hipFree(c->d_part); c->d_part = nullptr; c->part_cap = 0; VGCHK(hipMalloc(&c->d_part, 8 * 29 * (size_t)nblk)); c->part_cap = nblk; }
    const int want = (H && b) ? 1 : 0;
    if (c->profiling) hipEventRecord(c->ev0, c->stream);
    hipLaunchKernelGGL(k_vgicp_lin, dim3(nblk), dim3(VG_THREADS), 0, c->stream, c->n, (int)mode, c->d_sxyz, c->d_scov, to_iso(T), c->res, tab(c), c->d_cvox, c->d_cM, c->d_part, want);
    if (c->profiling) hipEventRecord(c->ev1, c->stream);
    VGCHK(sum_and_fetch<29>(c, nblk));
    VGCHK(hipGetLastError());
    if (c->profiling) { float ms = 0.f; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) { c->prof_ms += ms; c->prof_n++; } }
    c->noff = mode; c->slots = slots; c->linearized = true;
    *err = c->h_out[27];
    if (n_corr) *n_corr = (int32_t)c->h_out[28];
    if (want) {
        int idx = 0;
        for (int p = 0; p < 6; ++p) { for (int q = p; q < 6; ++q) { H[6 * p + q] = c->h_out[idx]; H[6 * q + p] = c->h_out[idx]; ++idx; } b[p] = c->h_out[21 + p]; }
    }
    return std::isfinite(*err) ? VG_OK : VG_ERR_NONFINITE;
}

int vgicp_profile_enable(vgicp_ctx* c, int32_t enable) {
    if (!c) return VG_ERR_INVALID;
    VGCHK(hipSetDevice(c->device));
    if (enable && !c->ev0) { VGCHK(hipEventCreate(&c->ev0)); VGCHK(hipEventCreate(&c->ev1)); }
    c->profiling = enable != 0;
    return VG_OK;
}
int vgicp_profile_read(vgicp_ctx* c, int64_t* launches, double* total_ms) {
    if (!c || !launches || !total_ms) return VG_ERR_INVALID;
    *launches = c->prof_n; *total_ms = c->prof_ms; c->prof_n = 0; c->prof_ms = 0.0;
    return VG_OK;
}

int vgicp_compute_error(vgicp_ctx* c, const double* T, double* err) {
    if (!c || !T || !err || !c->linearized) return VG_ERR_INVALID;
    VGCHK(hipSetDevice(c->device));
    const int nblk = (c->slots + VG_THREADS - 1) / VG_THREADS;
    hipLaunchKernelGGL(k_vgicp_err, dim3(nblk), dim3(VG_THREADS), 0, c->stream, c->slots, c->noff, c->d_sxyz, to_iso(T), tab(c), c->d_cvox, c->d_cM, c->d_part);
    VGCHK(sum_and_fetch<1>(c, nblk));
    *err = c->h_out[0];
    return std::isfinite(*err) ? VG_OK : VG_ERR_NONFINITE;
}

// ---- host side of LsqRegistration (lsq_registration_impl.hpp:48-165): 6 x 6 algebra between device reductions --------------
__host__ __device__ static bool solve6(const double* A, const double* rhs, double* x) {
    double L[36] = {0};
    for (int j = 0; j < 6; ++j) {
        double d = A[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k];
        if (!(d > 0.0)) return false;
        L[6 * j + j] = sqrt(d);
        for (int i = j + 1; i < 6; ++i) { double s = A[6 * i + j]; for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k]; L[6 * i + j] = s / L[6 * j + j]; }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) { double s = rhs[i]; for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k]; y[i] = s / L[6 * i + i]; }
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k]; x[i] = s / L[6 * i + i]; }
    return true;
}
__host__ __device__ static void so3_exp_R(const double* w, double* R) {      // so3.hpp:53-77, then Quaternion::toRotationMatrix
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double im, re;
    if (t2 < 1e-10) { const double t4 = t2 * t2; im = 0.5 - 1.0 / 48.0 * t2 + 1.0 / 3840.0 * t4; re = 1.0 - 1.0 / 8.0 * t2 + 1.0 / 384.0 * t4; }
    else { const double t = sqrt(t2), h = 0.5 * t; double sh, ch; sincos(h, &sh, &ch); im = sh / t; re = ch; }      // (one argument reduction for both)
    const double qw = re, qx = im * w[0], qy = im * w[1], qz = im * w[2];
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz, twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy; R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx; R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
__host__ __device__ static void compose(const double* d, const double* x0, double* xi) {
    double R[9]; so3_exp_R(d, R);
    for (int r = 0; r < 3; ++r) {
        for (int q = 0; q < 4; ++q) xi[4 * r + q] = R[3 * r] * x0[q] + R[3 * r + 1] * x0[4 + q] + R[3 * r + 2] * x0[8 + q];
        xi[4 * r + 3] += d[3 + r];
    }
    xi[12] = 0; xi[13] = 0; xi[14] = 0; xi[15] = 1;
}
// the same two with the increment's rotation matrix shared (the device loop calls both on the same d: one sin / cos less per pass)
__device__ static void compose_R(const double* d, const double* x0, double* xi, double* R) {
    so3_exp_R(d, R);
    for (int r = 0; r < 3; ++r) {
        for (int q = 0; q < 4; ++q) xi[4 * r + q] = R[3 * r] * x0[q] + R[3 * r + 1] * x0[4 + q] + R[3 * r + 2] * x0[8 + q];
        xi[4 * r + 3] += d[3 + r];
    }
    xi[12] = 0; xi[13] = 0; xi[14] = 0; xi[15] = 1;
}
// is_converged without its twelve divides (one lane, every workgroup waiting): for eps > 0, fl(a / eps) < 1 exactly when a < eps
// (a < eps: the quotient is below 1 and rounds to at most the largest double below 1; a >= eps: it is at least 1), and the
// maximum is below 1 exactly when every term is -- the same decision, bit for bit
__device__ static bool converged_R(const double* R, const double* d, double reps, double teps) {
    if (!(reps > 0.0) || !(teps > 0.0)) {
        double m = 0;
        for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) m = fmax(m, fabs(R[3 * r + q] - (r == q ? 1.0 : 0.0)) / reps);
        for (int r = 0; r < 3; ++r) m = fmax(m, fabs(d[3 + r]) / teps);
        return m < 1;
    }
    bool ok = true;
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) ok = ok && fabs(R[3 * r + q] - (r == q ? 1.0 : 0.0)) < reps;
    for (int r = 0; r < 3; ++r) ok = ok && fabs(d[3 + r]) < teps;
    return ok;
}
__host__ __device__ static bool is_converged(const double* d, double reps, double teps) {
    double R[9]; so3_exp_R(d, R);
    double m = 0;
    for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) m = fmax(m, fabs(R[3 * r + q] - (r == q ? 1.0 : 0.0)) / reps);
    for (int r = 0; r < 3; ++r) m = fmax(m, fabs(d[3 + r]) / teps);
    return m < 1;
}

}  // extern "C"

// ---- the whole LsqRegistration::align loop (lsq_registration_impl.hpp:48-165) in ONE launch ------------------------------------
// G resident workgroups share the slots.  Passes: (1) a linearisation at x0; (3) Levenberg-Marquardt's trial -- the error of the
// trial transform xi with the stored correspondences AND, in the same sweep over the slots, the linearisation at xi with fresh
// correspondences into the other cache: when the trial is accepted (the usual case) xi is the next x0 and that linearisation
// is exactly the one the next iteration would have computed (same slots per thread, same order of sums: the same bits), so an
// iteration costs one pass instead of two; a rejected trial just discards it.  Every pass ends in a symmetric exchange: each workgroup publishes its partial sums + an epoch flag, waits for all
// flags, adds all partials in workgroup order -- the same bits everywhere -- and its thread 0 advances the same Levenberg-
// Marquardt / Gauss-Newton state machine on them.  Identical code on identical numbers: every workgroup arrives at the same
// next transform, so a pass costs one exchange and the host sees one launch and one 0.5 kB read-back per alignment.
#define VG_MAXG 128
struct VgCoop { int flag[VG_MAXG]; double part[2][VG_MAXG][32]; };
struct VgAlignOut { double T[16]; double H[36]; double err; int iterations, converged, n_corr, lm_failed, status, pad; int seq, pad2; };      // seq: written last, the host polls it
struct VgGuess { double m[16]; };      // the initial transform travels in the kernel arguments, the result is written straight into pinned host memory
struct VgLm {                          // state of the optimiser (thread 0 of every workgroup)
    double x0[16], xi[16], H[36], Hout[36], b[6], d[6], Rd[9];      // Rd: rotation of the current increment d
    double lambda, nu, y0;
    int it, inner, converged, lm_failed, n_corr, phase;     // phase 1: linearise at x0, 2: error at xi
    int flip;                                               // set when the speculative linearisation was taken: its cache becomes the current one
};
// solve6 for the one lane that advances the optimiser between two passes (every workgroup waits for it): right-looking Cholesky
// with reciprocal pivots from v_rsq_f64 + two Newton steps (1-2 ulp) -- no IEEE divide or square root on the dependent chain
// (those cost about 1 k cycles each in one lane; solve6 has 33 of them in sequence).  Same algebra, same failure rule.
__device__ static bool solve6_dev(const double* A, const double* rhs, double* x) {
    double a[6][6], ri[6], y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) a[i][j] = A[6 * i + j];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double d = a[j][j];
        ok = ok && d > 0.0;
        const double r = vd::rsqrt_nr(d);
        ri[j] = r;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) a[i][j] *= r;
#pragma unroll
        for (int i = j + 1; i < 6; ++i)
#pragma unroll
            for (int k = j + 1; k <= i; ++k) a[i][k] -= a[i][j] * a[k][j];
    }
    if (!ok) return false;
#pragma unroll
    for (int i = 0; i < 6; ++i) { double s = rhs[i]; for (int k = 0; k < i; ++k) s -= a[i][k] * y[k]; y[i] = s * ri[i]; }
#pragma unroll
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= a[k][i] * x[k]; x[i] = s * ri[i]; }
    return true;
}
// consumes the sums of the pass just exchanged, returns the next pass (0 = finished); mirrors the host loop statement by statement
__device__ static int vg_lm_advance(VgLm& L, const double* tot, const vgicp_options& o) {
    if (L.phase == 1) {
        int idx = 0;
        for (int p = 0; p < 6; ++p) { for (int q = p; q < 6; ++q) { L.H[6 * p + q] = tot[idx]; L.H[6 * q + p] = tot[idx]; ++idx; } L.b[p] = tot[21 + p]; }
        L.y0 = tot[27]; L.n_corr = (int)tot[28];
        double nb[6];
        for (int k = 0; k < 6; ++k) { nb[k] = -L.b[k]; L.d[k] = 0.0; }
        if (o.optimizer == VGICP_GN) {
            if (!solve6_dev(L.H, nb, L.d)) { L.lm_failed = 1; return 0; }
            compose_R(L.d, L.x0, L.xi, L.Rd);
            for (int k = 0; k < 16; ++k) L.x0[k] = L.xi[k];
            for (int k = 0; k < 36; ++k) L.Hout[k] = L.H[k];
            ++L.it;
            L.converged = converged_R(L.Rd, L.d, o.rotation_epsilon, o.transformation_epsilon) ? 1 : 0;
            return (L.converged || L.it >= o.max_iterations) ? 0 : 1;
        }
        if (L.lambda < 0.0) { double m = 0; for (int k = 0; k < 6; ++k) m = fmax(m, fabs(L.H[7 * k])); L.lambda = o.lm_init_lambda_factor * m; }
        L.nu = 2.0; L.inner = 0;
    } else {
        // trial step evaluated: accept / reject (step_lm)
        const double yi = tot[29];
        bool moved = false;
        double den = 0; for (int k = 0; k < 6; ++k) den += L.d[k] * (L.lambda * L.d[k] - L.b[k]);
        const double rho = (L.y0 - yi) / den;
        bool stepped = false;
        if (rho < 0) {
            if (converged_R(L.Rd, L.d, o.rotation_epsilon, o.transformation_epsilon)) stepped = true;
            else { L.lambda = L.nu * L.lambda; L.nu = 2 * L.nu; ++L.inner; }
        } else {
            for (int k = 0; k < 16; ++k) L.x0[k] = L.xi[k];
            const double c3 = 2 * rho - 1;
            L.lambda = L.lambda * fmax(1.0 / 3.0, 1 - c3 * c3 * c3);
            for (int k = 0; k < 36; ++k) L.Hout[k] = L.H[k];
            stepped = true; moved = true;
        }
        if (stepped) {
            ++L.it;
            L.converged = converged_R(L.Rd, L.d, o.rotation_epsilon, o.transformation_epsilon) ? 1 : 0;
            if (L.converged || L.it >= o.max_iterations) return 0;
            if (!moved) { L.phase = 1; return 1; }           // x0 stayed: the linearisation at xi is of no use
            // x0 = xi: the pass that judged the trial has linearised there already (tot[0 .. 29), correspondences in the other cache)
            if (!isfinite(tot[27])) { L.lm_failed = -1; return 0; }
            L.flip = 1;
            int idx = 0;
            for (int p = 0; p < 6; ++p) { for (int q = p; q < 6; ++q) { L.H[6 * p + q] = tot[idx]; L.H[6 * q + p] = tot[idx]; ++idx; } L.b[p] = tot[21 + p]; }
            L.y0 = tot[27]; L.n_corr = (int)tot[28];
            L.nu = 2.0; L.inner = 0;
        }
    }
    // next trial of the inner loop: damped solve until one succeeds or the budget is spent
    double nb[6]; for (int k = 0; k < 6; ++k) nb[k] = -L.b[k];
    for (; L.inner < o.lm_max_iterations; ++L.inner) {
        double Hl[36]; for (int k = 0; k < 36; ++k) Hl[k] = L.H[k];
        for (int k = 0; k < 6; ++k) Hl[7 * k] += L.lambda;
        if (!solve6_dev(Hl, nb, L.d)) { L.lambda = L.nu * L.lambda; L.nu = 2 * L.nu; continue; }
        compose_R(L.d, L.x0, L.xi, L.Rd);
        L.phase = 2; return 3;
    }
    L.lm_failed = 1; ++L.it;                              // "lm not converged!!"
    return 0;
}

__global__ __launch_bounds__(VGA_THREADS) void k_vgicp_align(int n, int noff, const float* __restrict__ sxyz, const double* __restrict__ scov, double res, VoxTab V,
                                                            int* __restrict__ c_vox0, double* __restrict__ c_M0, int* __restrict__ c_vox1, double* __restrict__ c_M1,
                                                            vgicp_options o, VgCoop* coop, int epoch, VgGuess guess, VgAlignOut* out) {
    __shared__ VgLm L;
    __shared__ double mine[VGA_THREADS / 64][30], gath[VGA_THREADS / 32][32], tot[32];
    __shared__ Iso Tsh; __shared__ int action, cbuf;
    const int t = threadIdx.x, g = blockIdx.x, G = gridDim.x, slots = n * noff;
    const int epoch0 = epoch;                              // (unique per call: the host hands out growing epochs)
    if (t == 0) {
        for (int k = 0; k < 16; ++k) L.x0[k] = guess.m[k];
        for (int k = 0; k < 36; ++k) L.Hout[k] = (k % 7 == 0) ? 1.0 : 0.0;
        L.lambda = -1.0; L.nu = 2.0; L.y0 = 0.0; L.it = 0; L.inner = 0; L.converged = 0; L.lm_failed = 0; L.n_corr = 0; L.phase = 1; L.flip = 0; cbuf = 0;
        for (int q = 0; q < 12; ++q) Tsh.m[q] = L.x0[q];
        action = o.max_iterations > 0 ? 1 : 0;
    }
    __syncthreads();
#ifdef VG_STAMPS
    long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sp = wall_clock64(), sq; int npass = 0;
    #define VGS(k) do { sq = wall_clock64(); st[k] += sq - sp; sp = sq; } while (0)
#else
    #define VGS(k)
#endif
    while (action) {
        ++epoch;
        VGS(7);
        const int nv = action == 1 ? 29 : 30;
        const Iso T = Tsh;
        int* const cv_cur = cbuf ? c_vox1 : c_vox0; double* const cM_cur = cbuf ? c_M1 : c_M0;
        int* const cv_new = cbuf ? c_vox0 : c_vox1; double* const cM_new = cbuf ? c_M0 : c_M1;
        double acc[29], eacc = 0.0;
#pragma unroll
        for (int q = 0; q < 29; ++q) acc[q] = 0.0;
        if (action == 1) { for (int tid = g * VGA_THREADS + t; tid < slots; tid += G * VGA_THREADS) vgicp_lin_slot(tid, noff, sxyz, scov, T, res, V, cv_cur, cM_cur, acc, 1); }
        else {
            for (int tid = g * VGA_THREADS + t; tid < slots; tid += G * VGA_THREADS) {
                eacc += vgicp_err_slot(tid, noff, sxyz, T, V, cv_cur, cM_cur);
                vgicp_lin_slot(tid, noff, sxyz, scov, T, res, V, cv_new, cM_new, acc, 1);
            }
        }
        VGS(0);
        {   // workgroup fold (as block_fold), result in mine
            const int lane = t & 63, wave = t >> 6;
            double f0, f1;
            vd::wave_fold<29>(acc, f0, f1);
            if (lane < 16) { const int q = vd::fold_slot(lane); if (q < 29) mine[wave][q] = f0; if (q + 16 < 29) mine[wave][q + 16] = f1; }
            if (action != 1) { const double v = wave_sum64(eacc); if (lane == 0) mine[wave][29] = v; }
        }
        __syncthreads();
        double (*part)[32] = coop->part[epoch & 1];
        if (t < 64) {                      // wave 0 publishes: its lanes' stores, then lane 0's release (a fence waits for the whole wave's stores)
            if (t < nv) {
                double ws = 0.0;
#pragma unroll
                for (int w = 0; w < VGA_THREADS / 64; w += 4) ws += (mine[w][t] + mine[w + 1][t]) + (mine[w + 2][t] + mine[w + 3][t]);      // fixed order
                __hip_atomic_store(&part[g][t], ws, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // everything another workgroup reads from this one (the partials) was stored with agent-scope atomics, i.e. at the level
            // all XCDs share: waiting for those stores is all the ordering the flag needs -- a release fence would also write back
            // this XCD's whole L2 (the correspondence cache just written), an acquire on the other side would invalidate theirs
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t == 0) __hip_atomic_store(&coop->flag[g], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        VGS(1);
        if (t < G) while (__hip_atomic_load(&coop->flag[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch < 0) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        VGS(2);
        {   // thread (chunk c = t / 32, value q = t % 32) adds the partials of workgroups c, c + 16, ... in that order -- every load issued
            // before the first add (the gather is bound by L2 round trips) -- then 16 chunk sums per value are combined by a fixed tree
            constexpr int NCH = VGA_THREADS / 32, NLD = (VG_MAXG + NCH - 1) / NCH;
            const int q = t & 31, c = t >> 5;
            double gv[NLD];
#pragma unroll
            for (int u = 0; u < NLD; ++u) { const int w = c + u * NCH; gv[u] = (w < G && q < nv) ? __hip_atomic_load(&part[w][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0; }
            double cs = 0.0;
#pragma unroll
            for (int u = 0; u < NLD; ++u) cs += gv[u];
            gath[c][q] = cs;
        }
        __syncthreads();
        if (t < nv) {
            double r[VGA_THREADS / 32];
#pragma unroll
            for (int c = 0; c < VGA_THREADS / 32; ++c) r[c] = gath[c][t];
#pragma unroll
            for (int st = 1; st < VGA_THREADS / 32; st <<= 1)
#pragma unroll
                for (int c = 0; c < VGA_THREADS / 32; c += 2 * st) r[c] += r[c + st];
            tot[t] = r[0];
        }
        __syncthreads();
        VGS(3);
        if (t == 0) {
            int a = 0;
            if (!isfinite(tot[action == 1 ? 27 : 29])) { L.lm_failed = -1; a = 0; }      // non-finite error: stop, the host reports it
            else a = vg_lm_advance(L, tot, o);
            if (L.flip) { cbuf ^= 1; L.flip = 0; }
            if (a == 1) for (int q = 0; q < 12; ++q) Tsh.m[q] = L.x0[q];
            if (a == 3) for (int q = 0; q < 12; ++q) Tsh.m[q] = L.xi[q];
            action = a;
        }
        __syncthreads();
        VGS(4);
#ifdef VG_STAMPS
        ++npass;
#endif
    }
#ifdef VG_STAMPS
    if (t == 0 && g != 0) printf("wg %d eval %lld wait %lld\n", g, st[0], st[2]);
    if (g == 0 && t == 0) printf("passes %d; 100 MHz ticks: eval %lld, fold+publish %lld, wait %lld, gather+sum %lld, LM %lld\n", npass, st[0], st[1], st[2], st[3], st[4]);
#endif
    if (g == 0 && t == 0) {
        for (int k = 0; k < 16; ++k) out->T[k] = L.x0[k];
        for (int k = 0; k < 36; ++k) out->H[k] = L.Hout[k];
        out->err = L.y0; out->iterations = L.it; out->converged = L.converged; out->n_corr = L.n_corr;
        out->lm_failed = L.lm_failed > 0 ? 1 : 0; out->status = L.lm_failed < 0 ? VG_ERR_NONFINITE : VG_OK;
        out->pad = cbuf;                                    // which cache holds the correspondences of the last linearisation used
        __threadfence_system();                             // the record is in (pinned) host memory before its sequence number
        __hip_atomic_store(&out->seq, epoch0 + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

extern "C" {

static int vgicp_align_host(vgicp_ctx* c, const double* guess, const vgicp_options* o, double* T_out, vgicp_summary* out);

int vgicp_set_knn_grid(vgicp_ctx* c, int32_t min_points, double cell) {
    if (!c || min_points < 0 || !(cell > 0.0)) return VG_ERR_INVALID;
    c->grid_min = min_points; c->grid_h = (float)cell;
    return VG_OK;
}

int vgicp_align(vgicp_ctx* c, const double* guess, const vgicp_options* o, double* T_out, vgicp_summary* out) {
    if (!c || !guess || !o || !T_out || !out) return VG_ERR_INVALID;
    if (!c->nvox || !c->n || (o->neighbor_mode != VGICP_DIRECT1 && o->neighbor_mode != VGICP_DIRECT7 && o->neighbor_mode != VGICP_DIRECT27)) return VG_ERR_INVALID;
    if (VIL_TUNE_ENV("VGICP_HOST_LOOP")) return vgicp_align_host(c, guess, o, T_out, out);       // the step logic on the host between launches (cross-check)
    VGCHK(hipSetDevice(c->device));
    const int mode = o->neighbor_mode, slots = c->n * mode, nblk = (slots + VGA_THREADS - 1) / VGA_THREADS;
    if (slots > c->slots_cap) { hipFree(c->d_cvox); hipFree(c->d_cM); c->d_cvox = nullptr; c->d_cM = nullptr; c->slots_cap = 0; VGCHK(hipMalloc(&c->d_cvox, 4 * (size_t)slots)); VGCHK(hipMalloc(&c->d_cM, 8 * 9 * (size_t)slots)); c->slots_cap = slots; }
    if (slots > c->slots_cap2) { hipFree(c->d_cvox2); hipFree(c->d_cM2); c->d_cvox2 = nullptr; c->d_cM2 = nullptr; c->slots_cap2 = 0; VGCHK(hipMalloc(&c->d_cvox2, 4 * (size_t)c->slots_cap)); VGCHK(hipMalloc(&c->d_cM2, 8 * 9 * (size_t)c->slots_cap)); c->slots_cap2 = c->slots_cap; }
    if (!c->d_coop) { VGCHK(hipMalloc(&c->d_coop, sizeof(VgCoop))); VGCHK(hipMemsetAsync(c->d_coop, 0, sizeof(VgCoop), c->stream)); VGCHK(hipHostMalloc(&c->h_aout, sizeof(VgAlignOut), hipHostMallocMapped)); VGCHK(hipHostGetDevicePointer(&c->d_aout, c->h_aout, 0)); c->coop_epoch = 0; }
    VgAlignOut* ho = (VgAlignOut*)c->h_aout;
    VgGuess gs; std::memcpy(gs.m, guess, sizeof gs.m);
    if (c->coop_cap < 0) c->coop_cap = vilcoop::capacity((const void*)k_vgicp_align, VGA_THREADS, 0, c->device);
    if (c->coop_cap < 1) return vgicp_align_host(c, guess, o, T_out, out);      // the kernel cannot be resident on this device: one launch per pass instead
    std::unique_lock<std::shared_mutex> coop_lock(vilcoop::gate(c->device));                     // held until the result record has arrived (vil_coop.hpp)
    int G = std::min(std::min(nblk, VG_MAXG), c->coop_cap);     // all workgroups resident: they wait for each other.  A pass is bound by the slots per thread (each a chain of dependent gathers), so as many workgroups as there are 256-slot blocks, up to 128 (VGICP_G sweeps it)
    if (const char* ev = VIL_TUNE_ENV("VGICP_G")) G = std::max(1, std::min(VG_MAXG, atoi(ev)));
    const int epoch_of_call = c->coop_epoch;
    ho->seq = -1;
    hipLaunchKernelGGL(k_vgicp_align, dim3(G), dim3(VGA_THREADS), 0, c->stream, c->n, mode, c->d_sxyz, c->d_scov, c->res, tab(c), c->d_cvox, c->d_cM, c->d_cvox2, c->d_cM2, *o, (VgCoop*)c->d_coop, c->coop_epoch, gs, (VgAlignOut*)c->d_aout);
    c->coop_epoch += 4 * (o->max_iterations * (o->lm_max_iterations + 1) + 4);      // epochs only grow: nothing to reset between calls
    if (c->coop_epoch > (1 << 30)) { VGCHK(hipMemsetAsync(c->d_coop, 0, sizeof(VgCoop), c->stream)); c->coop_epoch = 0; }
    {   // the kernel's last act is the record's sequence number in pinned memory: polling it returns a few microseconds before the
        // stream's completion signal would (the launch queue stays in order either way); a kernel that never gets there is left to
        // hipStreamSynchronize, which reports the fault
        volatile int* seq = &ho->seq;
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (long spin = 0;; ++spin) {
            if (*seq == epoch_of_call + 1) { seen = true; break; }
            if ((spin & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
        }
        if (!seen) VGCHK(hipStreamSynchronize(c->stream));
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    VGCHK(hipGetLastError());
    if (ho->pad) { std::swap(c->d_cvox, c->d_cvox2); std::swap(c->d_cM, c->d_cM2); std::swap(c->slots_cap, c->slots_cap2); }      // vgicp_error() after an alignment sees its last linearisation
    c->noff = mode; c->slots = slots; c->linearized = true;
    std::memset(out, 0, sizeof *out);
    out->iterations = ho->iterations; out->converged = ho->converged; out->n_correspondences = ho->n_corr; out->lm_failed = ho->lm_failed; out->final_error = ho->err;
    std::memcpy(out->final_hessian, ho->H, sizeof ho->H);
    std::memcpy(T_out, ho->T, sizeof ho->T);
    return ho->status;
}

static int vgicp_align_host(vgicp_ctx* c, const double* guess, const vgicp_options* o, double* T_out, vgicp_summary* out) {
    double x0[16]; std::memcpy(x0, guess, sizeof x0);
    double lambda = -1.0;
    bool converged = false;
    std::memset(out, 0, sizeof *out);
    for (int k = 0; k < 36; ++k) out->final_hessian[k] = (k % 7 == 0) ? 1.0 : 0.0;
    int it = 0;
    for (; it < o->max_iterations && !converged; ++it) {
        double H[36], b[6], d[6] = {0, 0, 0, 0, 0, 0}, nb[6], y0;
        int32_t nc = 0;
        int st = vgicp_linearize(c, x0, o->neighbor_mode, &y0, H, b, &nc);
        if (st != VG_OK) return st;
        out->n_correspondences = nc; out->final_error = y0;
        for (int k = 0; k < 6; ++k) nb[k] = -b[k];
        bool stepped = false;
        if (o->optimizer == VGICP_GN) {                                        // step_gn
            if (!solve6(H, nb, d)) { out->lm_failed = 1; break; }
            double xi[16]; compose(d, x0, xi); std::memcpy(x0, xi, sizeof x0); std::memcpy(out->final_hessian, H, sizeof H); stepped = true;
        } else {                                                               // step_lm
            if (lambda < 0.0) { double m = 0; for (int k = 0; k < 6; ++k) m = std::fmax(m, std::fabs(H[7 * k])); lambda = o->lm_init_lambda_factor * m; }
            double nu = 2.0;
            for (int i = 0; i < o->lm_max_iterations; ++i) {
                double Hl[36]; std::memcpy(Hl, H, sizeof H);
                for (int k = 0; k < 6; ++k) Hl[7 * k] += lambda;
                if (!solve6(Hl, nb, d)) { lambda = nu * lambda; nu = 2 * nu; continue; }
                double xi[16], yi; compose(d, x0, xi);
                st = vgicp_compute_error(c, xi, &yi);
                if (st != VG_OK) return st;
                double den = 0; for (int k = 0; k < 6; ++k) den += d[k] * (lambda * d[k] - b[k]);
                const double rho = (y0 - yi) / den;
                if (rho < 0) {
                    if (is_converged(d, o->rotation_epsilon, o->transformation_epsilon)) { stepped = true; break; }
                    lambda = nu * lambda; nu = 2 * nu; continue;
                }
                std::memcpy(x0, xi, sizeof x0);
                lambda = lambda * std::fmax(1.0 / 3.0, 1 - std::pow(2 * rho - 1, 3));
                std::memcpy(out->final_hessian, H, sizeof H);
                stepped = true;
                break;
            }
        }
        if (!stepped) { out->lm_failed = 1; ++it; break; }
        converged = is_converged(d, o->rotation_epsilon, o->transformation_epsilon);
    }
    out->iterations = it; out->converged = converged ? 1 : 0;
    std::memcpy(T_out, x0, sizeof x0);
    return VG_OK;
}

}  // extern "C"
