// vilmap.hip -- LiDAR scan-to-map registration on gfx950 behind include/vilmap.h (SURVEY 8(f) row 2,
// lidar_mapping/src/localMapping.cpp:590-791).
//
// Per scan point the reference does a kd-tree query in the local map and a tiny dense fit (3x3 eigen-decomposition for
// corner points, 5x3 least squares for surf points), then hands the accepted points to Ceres as edge / plane factors.
// Here: the two map clouds are gridded once per scan (vil_knn.hpp), ONE kernel per feature class does query + fit with a
// thread per scan point (everything in registers), the accepted correspondences are compacted on the host in scan order,
// and the 7-parameter solve is vil_solve on a one-pose window that holds exactly these factors (k_sweep's LiDAR roles).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/vilmap.h"
#include "vil_internal.h"
#include "vil_coop.hpp"
#include "vil_knn.hpp"
#include "vil_pose1.hpp"

#define VM_OK 0
#define VM_ERR_INVALID -1
#define VM_ERR_DEVICE -2
#define VM_THREADS 256
#define VMCHK(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) { if (getenv("VIL_DEBUG")) fprintf(stderr, "vilmap.hip:%d: %s\n", __LINE__, hipGetErrorString(e_)); return VM_ERR_DEVICE; } } while (0)

namespace {
using namespace vknn;

using PoseD = vp1::PoseRT;      // rotation matrix + translation of the scan pose (double)

// eigen-decomposition of a symmetric 3x3 (cyclic Jacobi): A -> eigenvalues on its diagonal, V columns = eigenvectors
__device__ __forceinline__ void jacobi3(double* A, double* V) {
    V[0] = 1; V[1] = 0; V[2] = 0; V[3] = 0; V[4] = 1; V[5] = 0; V[6] = 0; V[7] = 0; V[8] = 1;
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off <= 1e-40 * (A[0] * A[0] + A[4] * A[4] + A[8] * A[8]) || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[3 * p + q];
                if (apq != 0.0) {
                    // the rotation of the textbook form  tau = d / (2 apq), t = sgn(tau) / (|tau| + sqrt(1 + tau^2)), c = 1 / sqrt(1 + t^2), s = t c
                    // written without its three divisions: u = |d| + sqrt(d^2 + 4 apq^2), c = u / sqrt(u^2 + 4 apq^2), s = sgn(tau) 2 |apq| / sqrt(...)
                    // (a lone thread's divide / sqrt latency is what this kernel costs: 2 reciprocal square roots per rotation instead of 2 sqrt + 3 div)
                    const double d = A[3 * q + q] - A[3 * p + p], a2 = 4.0 * apq * apq, h2 = d * d + a2;
                    const double u = fabs(d) + h2 * vd::rsqrt_nr(h2), r = vd::rsqrt_nr(u * u + a2);
                    const bool pos = (d >= 0.0 && apq > 0.0) || (d <= 0.0 && apq < 0.0);
                    const double c = u * r, sn = (pos ? 2.0 : -2.0) * fabs(apq) * r;
#pragma unroll
                    for (int r = 0; r < 3; ++r) { const double akp = A[3 * r + p], akq = A[3 * r + q]; A[3 * r + p] = c * akp - sn * akq; A[3 * r + q] = sn * akp + c * akq; }
#pragma unroll
                    for (int r = 0; r < 3; ++r) { const double apk = A[3 * p + r], aqk = A[3 * q + r]; A[3 * p + r] = c * apk - sn * aqk; A[3 * q + r] = sn * apk + c * aqk; }
#pragma unroll
                    for (int r = 0; r < 3; ++r) { const double vkp = V[3 * r + p], vkq = V[3 * r + q]; V[3 * r + p] = c * vkp - sn * vkq; V[3 * r + q] = sn * vkp + c * vkq; }
                }
            }
    }
}

__device__ __forceinline__ void to_map(const PoseD& T, const float* p, float& sx, float& sy, float& sz) {   // pointAssociateToMap: double, stored as float
    const double x = p[0], y = p[1], z = p[2];
    sx = (float)(T.R[0] * x + T.R[1] * y + T.R[2] * z + T.t[0]);
    sy = (float)(T.R[3] * x + T.R[4] * y + T.R[5] * z + T.t[1]);
    sz = (float)(T.R[6] * x + T.R[7] * y + T.R[8] * z + T.t[2]);
}

// min |A x - b| for a 5 x 3 A (row-major in P, overwritten) by column-pivoted Householder QR -- what
// matA0.colPivHouseholderQr().solve(matB0) does (localMapping.cpp:716)
__device__ __forceinline__ void qr_solve_5x3(double* P, double* b, double* x) {
    int perm[3] = {0, 1, 2};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int pv = k; double best = -1.0;
#pragma unroll
        for (int c = k; c < 3; ++c) { double s = 0; for (int r = k; r < 5; ++r) s += P[3 * r + c] * P[3 * r + c]; if (s > best) { best = s; pv = c; } }
        if (pv != k) { for (int r = 0; r < 5; ++r) { const double tmp = P[3 * r + k]; P[3 * r + k] = P[3 * r + pv]; P[3 * r + pv] = tmp; } const int tp = perm[k]; perm[k] = perm[pv]; perm[pv] = tp; }
        const double nrm = best > 0.0 ? best * vd::rsqrt_nr(best) : 0.0, akk = P[3 * k + k], alpha = akk > 0 ? -nrm : nrm;   // reciprocal estimates + Newton steps: this thread's divide / sqrt chain is the kernel's critical path
        const double v0 = akk - alpha;
        double vv = v0 * v0;
        for (int r = k + 1; r < 5; ++r) vv += P[3 * r + k] * P[3 * r + k];
        if (vv > 0) {
            const double beta = 2.0 * vd::rcp_nr(vv);
#pragma unroll
            for (int c = k + 1; c < 3; ++c) {
                double s = v0 * P[3 * k + c];
                for (int r = k + 1; r < 5; ++r) s += P[3 * r + k] * P[3 * r + c];
                s *= beta;
                P[3 * k + c] -= s * v0;
                for (int r = k + 1; r < 5; ++r) P[3 * r + c] -= s * P[3 * r + k];
            }
            double s = v0 * b[k];
            for (int r = k + 1; r < 5; ++r) s += P[3 * r + k] * b[r];
            s *= beta;
            b[k] -= s * v0;
            for (int r = k + 1; r < 5; ++r) b[r] -= s * P[3 * r + k];
        }
        P[3 * k + k] = alpha;
    }
    double y[3];
    const double r0 = vd::rcp_nr(P[0]), r1 = vd::rcp_nr(P[4]), r2 = vd::rcp_nr(P[8]);      // independent of each other (a zero pivot gives a NaN the caller rejects)
    y[2] = b[2] * r2;
    y[1] = (b[1] - P[5] * y[2]) * r1;
    y[0] = (b[0] - P[1] * y[1] - P[2] * y[2]) * r0;
#pragma unroll
    for (int k = 0; k < 3; ++k) x[perm[k]] = y[k];
}

// ---- search: one WAVE per scan point (vil_knn.hpp); queries [0, nqc) are corner points against the corner map (5-NN),
//      [nqc, nqc + nqs) surf points against the surf map (10-NN).  nn: 10 ints per query, nd5: the 5th squared distance
//      (or a value >= 1 when the query is rejected: fewer than five map points within 1 m, localMapping.cpp:613,:705)
#define VM_QPB 4
__global__ __launch_bounds__(64 * VM_QPB) void k_map_search(int nqc, int nqs, const float* __restrict__ scan, PoseD T, const PoseD* __restrict__ Tdev, int ncm, GridTab Gc, const int* __restrict__ oc, const float* __restrict__ xc,
                                                            int nsm, GridTab Gs, const int* __restrict__ os, const float* __restrict__ xs, int* __restrict__ nn, float* __restrict__ nd5,
                                                            const float* __restrict__ smap4, float* __restrict__ nint) {
    __shared__ int wl_all[VM_QPB * KNN_WL_CAP];
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qi = blockIdx.x * VM_QPB + wave;
    if (qi >= nqc + nqs) return;
    if (Tdev) T = *Tdev;                 // second round of the single-submission registration: the pose the first solve left on the device
    const bool surf = qi >= nqc;
    const int kk = surf ? 10 : 5, nmap = surf ? nsm : ncm;
    float d5 = 3.0e38f;
    unsigned long long best = ~0ull;
    if (nmap >= kk) {
        float sx, sy, sz;
        to_map(T, scan + 4 * (size_t)qi, sx, sy, sz);
        const bool ok = surf ? knn_wave_query(best, sx, sy, sz, 10, nsm, Gs, os, xs, wl_all + wave * KNN_WL_CAP, 1.0f, 5)
                             : knn_wave_query(best, sx, sy, sz, 5, ncm, Gc, oc, xc, wl_all + wave * KNN_WL_CAP, 1.0f, 5);
        if (ok) d5 = knn_key_d(readlane_u64(best, 4));
    }
    if (lane < 10) {
        const unsigned idx = (unsigned)best;
        nn[10 * (size_t)qi + lane] = (int)idx;
        // the fit re-ranks a surf point's ten neighbours by intensity: fetch the ten intensities here, ten lanes side by side with
        // thousands of other waves to hide the gather behind, instead of ten dependent loads in the fit's lone thread
        if (surf) nint[10 * (size_t)qi + lane] = idx < (unsigned)nsm ? smap4[4 * (size_t)idx + 3] : 0.0f;
    }
    if (lane == 0) nd5[qi] = d5;
}

// ---- fit: one thread per scan point; slot = 10 doubles: corner [valid, cp(3), a(3), b(3)], surf [valid, cp(3), n(3), d, -]
__device__ __forceinline__ bool map_fit_one(int i, int nqc, const float* __restrict__ scan, const float* __restrict__ cmap, const float* __restrict__ smap,
                                            const int* __restrict__ nn, const float* __restrict__ nd5, const float* __restrict__ nint, double* __restrict__ slot) {
    double* o = slot + (size_t)10 * i;
    o[0] = 0.0;
    if (!(nd5[i] < 1.0f)) return false;
    const int* nb = nn + 10 * (size_t)i;
    if (i < nqc) {                                                           // corner points (localMapping.cpp:607-660)
        double cx = 0, cy = 0, cz = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) { cx += (double)cmap[4 * nb[j]]; cy += (double)cmap[4 * nb[j] + 1]; cz += (double)cmap[4 * nb[j] + 2]; }
        cx /= 5.0; cy /= 5.0; cz /= 5.0;
        double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, V[9];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const double d0 = (double)cmap[4 * nb[j]] - cx, d1 = (double)cmap[4 * nb[j] + 1] - cy, d2 = (double)cmap[4 * nb[j] + 2] - cz;
            A[0] += d0 * d0; A[1] += d0 * d1; A[2] += d0 * d2; A[3] += d1 * d0; A[4] += d1 * d1; A[5] += d1 * d2; A[6] += d2 * d0; A[7] += d2 * d1; A[8] += d2 * d2;
        }
        jacobi3(A, V);
        int m2 = 0; if (A[4] > A[0]) m2 = 1; if (A[8] > A[4 * m2]) m2 = 2;      // largest and middle eigenvalue
        const int ma = (m2 + 1) % 3, mb = (m2 + 2) % 3;
        const double l2 = A[4 * m2], l1 = fmax(A[4 * ma], A[4 * mb]);
        if (!(l2 > 3.0 * l1)) return false;
        const double ux = m2 == 0 ? V[0] : (m2 == 1 ? V[1] : V[2]), uy = m2 == 0 ? V[3] : (m2 == 1 ? V[4] : V[5]), uz = m2 == 0 ? V[6] : (m2 == 1 ? V[7] : V[8]);
        o[0] = 1.0; o[1] = scan[4 * i]; o[2] = scan[4 * i + 1]; o[3] = scan[4 * i + 2];
        o[4] = 0.1 * ux + cx; o[5] = 0.1 * uy + cy; o[6] = 0.1 * uz + cz;
        o[7] = -0.1 * ux + cx; o[8] = -0.1 * uy + cy; o[9] = -0.1 * uz + cz;
        return true;
    }
    // surf points (localMapping.cpp:683-741): re-rank the ten by |intensity difference| (ties: smaller map index), keep five
    const float qi = scan[4 * i + 3];
    KnnList S; knn_init(S);
#pragma unroll
    for (int m = 0; m < 10; ++m) knn_insert(S, fabsf(nint[10 * (size_t)i + m] - qi), nb[m]);      // intensities gathered by the search kernel
    double P[15], Q[15], rhs[5] = {-1, -1, -1, -1, -1}, nv[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 5; ++j) { P[3 * j] = (double)smap[4 * S.bi[j]]; P[3 * j + 1] = (double)smap[4 * S.bi[j] + 1]; P[3 * j + 2] = (double)smap[4 * S.bi[j] + 2]; }
#pragma unroll
    for (int j = 0; j < 15; ++j) Q[j] = P[j];
    qr_solve_5x3(Q, rhs, nv);
    double nx = nv[0], ny = nv[1], nz = nv[2];
    const double d = vd::rsqrt_nr(nx * nx + ny * ny + nz * nz);          // 1 / |n|
    nx *= d; ny *= d; nz *= d;
    bool ok = isfinite(d) && isfinite(nx);
#pragma unroll
    for (int j = 0; j < 5; ++j) ok = ok && !(fabs(nx * P[3 * j] + ny * P[3 * j + 1] + nz * P[3 * j + 2] + d) > 0.2);
    if (!ok) return false;
    o[0] = 1.0; o[1] = scan[4 * i]; o[2] = scan[4 * i + 1]; o[3] = scan[4 * i + 2];
    o[4] = nx; o[5] = ny; o[6] = nz; o[7] = d;
    return true;
}
// blk[2 b], blk[2 b + 1] = accepted corner / surf slots of workgroup b: the compaction's offsets without a second pass over the slots
__global__ __launch_bounds__(VM_THREADS) void k_map_fit(int nqc, int nqs, const float* __restrict__ scan, const float* __restrict__ cmap, const float* __restrict__ smap,
                                                        const int* __restrict__ nn, const float* __restrict__ nd5, const float* __restrict__ nint, double* __restrict__ slot, int* __restrict__ blk) {
    __shared__ int s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blockIdx.x * VM_THREADS + threadIdx.x;
    const bool ok = i < nqc + nqs && map_fit_one(i, nqc, scan, cmap, smap, nn, nd5, nint, slot);
    const unsigned long long be = __ballot(ok && i < nqc), bp = __ballot(ok && i >= nqc);
    if ((threadIdx.x & 63) == 0) { if (be) atomicAdd(&s_cnt[0], __popcll(be)); if (bp) atomicAdd(&s_cnt[1], __popcll(bp)); }
    __syncthreads();
    if (threadIdx.x < 2) blk[2 * blockIdx.x + threadIdx.x] = s_cnt[threadIdx.x];
}


// ---- ordered compaction of the accepted slots into the solver's structure-of-arrays factor tables (vil_internal.h): the same
//      workgroup partition as k_map_fit; a workgroup's output offset is the sum of the preceding workgroups' counts (blk), a
//      slot's position inside it comes from wave ballots, so the tables keep the scan order (the order in which the reference
//      adds the residual blocks).  cnt[0] = edges, cnt[1] = planes.
__global__ __launch_bounds__(VM_THREADS) void k_map_compact(int nqc, int nqs, const double* __restrict__ slot, const int* __restrict__ blk, double* __restrict__ edge_soa, int es,
                                                            double* __restrict__ plane_soa, int ps, int* __restrict__ cnt) {
    __shared__ int s_off[2], s_w[2][VM_THREADS / 64];
    const int t = threadIdx.x, b = blockIdx.x, lane = t & 63, w = t >> 6;
    if (t < 2) s_off[t] = 0;
    __syncthreads();
    int pe = 0, pp = 0;
    for (int j = t; j < b; j += VM_THREADS) { pe += blk[2 * j]; pp += blk[2 * j + 1]; }
    for (int o = 32; o; o >>= 1) { pe += __shfl_xor(pe, o); pp += __shfl_xor(pp, o); }
    if (lane == 0 && (pe | pp)) { atomicAdd(&s_off[0], pe); atomicAdd(&s_off[1], pp); }
    const int i = b * VM_THREADS + t;
    const double* o = slot + (size_t)10 * i;
    const bool ok = i < nqc + nqs && o[0] != 0.0, edge = i < nqc;
    const unsigned long long be = __ballot(ok && edge), bp = __ballot(ok && !edge), below = (1ull << lane) - 1ull;
    if (lane == 0) { s_w[0][w] = __popcll(be); s_w[1][w] = __popcll(bp); }
    __syncthreads();
    if (ok) {
        int pos = s_off[edge ? 0 : 1] + __popcll((edge ? be : bp) & below);
        for (int v = 0; v < w; ++v) pos += s_w[edge ? 0 : 1][v];
        if (edge) { for (int q = 0; q < 9; ++q) edge_soa[(size_t)q * es + pos] = o[1 + q]; }
        else { for (int q = 0; q < 7; ++q) plane_soa[(size_t)q * ps + pos] = o[1 + q]; }
    }
    if (b == (int)gridDim.x - 1 && t < 2) cnt[t] = s_off[t] + blk[2 * b + t];
}

void quat_to_R(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

}  // namespace

struct vmap_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int nc = 0, ns = 0;
    float* d_cmap = nullptr; float* d_smap = nullptr; size_t cmap_cap = 0, smap_cap = 0;
    vknn::GridBuild gc, gs;
    float hc = 1.0f, hs = 1.0f;                      // cell sizes, adapted to the maps' densities
    // scan staging: corner points then surf points
    float* d_scan = nullptr; size_t scan_cap = 0; float* h_scan = nullptr; size_t h_scan_cap = 0; char* d_work = nullptr; size_t work_cap = 0;
    const float* up_corner = nullptr; const float* up_surf = nullptr; int up_nc = -1, up_ns = -1; bool scan_valid = false;
    double* h_slot = nullptr; size_t h_slot_cap = 0;       // pinned: the slot read-back is a plain DMA
    double* d_soa = nullptr; size_t soa_cap = 0; int* d_cnt = nullptr; int* h_cnt = nullptr;   // device-resident factor tables of vmap_align
    vp1::Pose1Coop* d_coop = nullptr; int reg_epoch = 0;   // meeting point of the pose solve's workgroups; epoch numbers are never reused
    char* d_reg = nullptr; char* h_reg = nullptr;     // single-submission registration: pose (7 doubles) | PoseRT | 2 x Pose1Out, and its pinned mirror
    char* h_res = nullptr; void* d_res = nullptr;     // pinned + mapped: the second round's Pose1Out | sequence word, written by k_pose_solve, polled by the host
    int coop_cap = -1;                                 // workgroups of k_pose_solve the device holds at once (vil_coop.hpp)
    int fused_max = 1 << 20;                           // scans up to this many points take the one-launch pose solve (vmap_set_fused_max; 0 = always the window solver)
    bool profiling = false; hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr}; long long prof_n[2] = {0, 0}; double prof_ms[2] = {0.0, 0.0};
};

namespace {
#define VM_OCC 6.0              // points per occupied grid cell the cell sizes are steered to
int upload_scan(vmap_ctx* c, int n_corner, const float* corner, int n_surf, const float* surf) {
    const int nq = n_corner + n_surf;
    const size_t need_scan = 16 * (size_t)nq + 16, need_work = (size_t)nq * (40 + 4 + 80 + 40) + 8 * ((size_t)nq / VM_THREADS + 1) + 256;
    if (need_scan > c->scan_cap) { hipFree(c->d_scan); c->d_scan = nullptr; c->scan_cap = 0; VMCHK(hipMalloc(&c->d_scan, 2 * need_scan)); c->scan_cap = 2 * need_scan; }
    if (need_work > c->work_cap) { hipFree(c->d_work); c->d_work = nullptr; c->work_cap = 0; VMCHK(hipMalloc(&c->d_work, 2 * need_work)); c->work_cap = 2 * need_work; }
    // through a pinned image: one DMA instead of two staged pageable copies (every entry point ends with a stream
    // synchronisation, so the image is free again when the next call starts)
    if (need_scan > c->h_scan_cap) { if (c->h_scan) hipHostFree(c->h_scan); c->h_scan = nullptr; c->h_scan_cap = 0; VMCHK(hipHostMalloc((void**)&c->h_scan, 2 * need_scan, hipHostMallocDefault)); c->h_scan_cap = 2 * need_scan; }
    if (n_corner) memcpy(c->h_scan, corner, 16 * (size_t)n_corner);
    if (n_surf) memcpy(c->h_scan + 4 * (size_t)n_corner, surf, 16 * (size_t)n_surf);
    if (nq) VMCHK(hipMemcpyAsync(c->d_scan, c->h_scan, 16 * (size_t)nq, hipMemcpyHostToDevice, c->stream));
    return VM_OK;
}
// association of the uploaded scan at pose (q, t); compacted on the host in scan order
int associate_uploaded(vmap_ctx* c, int n_corner, int n_surf, const double* q, const double* t, int32_t* n_edge, double* edge9, int32_t* n_plane, double* plane7) {
    PoseD T; quat_to_R(q, T.R); T.t[0] = t[0]; T.t[1] = t[1]; T.t[2] = t[2];
    *n_edge = 0; *n_plane = 0;
    const int nqc = c->nc ? n_corner : 0, nqs = c->ns ? n_surf : 0;       // an empty map yields no factors of that class
    const int nq = n_corner + n_surf;
    if (nq == 0 || nqc + nqs == 0) return VM_OK;
    // queries are addressed in the uploaded layout (corner block, then surf block); a class without a map is skipped by nmap < k
    double* d_slot = (double*)c->d_work; int* d_nn = (int*)(c->d_work + 80 * (size_t)nq); float* d_nd5 = (float*)(d_nn + 10 * (size_t)nq); float* d_nint = d_nd5 + nq; int* d_blk = (int*)(c->d_work + (((size_t)164 * nq + 7) & ~(size_t)7));
    if (c->profiling) hipEventRecord(c->ev[0], c->stream);
    hipLaunchKernelGGL(k_map_search, dim3((nq + VM_QPB - 1) / VM_QPB), dim3(64 * VM_QPB), 0, c->stream, n_corner, n_surf, c->d_scan, T, (const PoseD*)nullptr, c->nc, c->gc.G, c->gc.order, c->gc.cxyz,
                       c->ns, c->gs.G, c->gs.order, c->gs.cxyz, d_nn, d_nd5, c->d_smap, d_nint);
    if (c->profiling) { hipEventRecord(c->ev[1], c->stream); hipEventRecord(c->ev[2], c->stream); }
    hipLaunchKernelGGL(k_map_fit, dim3((nq + VM_THREADS - 1) / VM_THREADS), dim3(VM_THREADS), 0, c->stream, n_corner, n_surf, c->d_scan, c->d_cmap, c->d_smap, d_nn, d_nd5, d_nint, d_slot, d_blk);
    if (c->profiling) hipEventRecord(c->ev[3], c->stream);
    if (10 * (size_t)nq > c->h_slot_cap) {
        if (c->h_slot) hipHostFree(c->h_slot);
        c->h_slot = nullptr; c->h_slot_cap = 0;
        VMCHK(hipHostMalloc((void**)&c->h_slot, 8 * 20 * (size_t)nq, hipHostMallocDefault)); c->h_slot_cap = 20 * (size_t)nq;
    }
    VMCHK(hipMemcpyAsync(c->h_slot, d_slot, 80 * (size_t)nq, hipMemcpyDeviceToHost, c->stream));
    VMCHK(hipStreamSynchronize(c->stream));
    if (c->profiling) for (int k = 0; k < 2; ++k) { float ms = 0.f; if (hipEventElapsedTime(&ms, c->ev[2 * k], c->ev[2 * k + 1]) == hipSuccess) { c->prof_ms[k] += ms; c->prof_n[k]++; } }
    for (int i = 0; i < n_corner; ++i) if (c->h_slot[10 * (size_t)i] != 0.0) { memcpy(edge9 + 9 * (size_t)(*n_edge), &c->h_slot[10 * (size_t)i + 1], 72); ++*n_edge; }
    for (int i = n_corner; i < nq; ++i) if (c->h_slot[10 * (size_t)i] != 0.0) { memcpy(plane7 + 7 * (size_t)(*n_plane), &c->h_slot[10 * (size_t)i + 1], 56); ++*n_plane; }
    VMCHK(hipGetLastError());
    return VM_OK;
}
// association of the uploaded scan with the factor tables left ON THE DEVICE in the solver's layout; only the two counts come back
int associate_device(vmap_ctx* c, int n_corner, int n_surf, const double* q, const double* t, int32_t* n_edge, int32_t* n_plane, vil_device_lidar* dl) {
    PoseD T; quat_to_R(q, T.R); T.t[0] = t[0]; T.t[1] = t[1]; T.t[2] = t[2];
    *n_edge = 0; *n_plane = 0;
    const int nq = n_corner + n_surf;
    const int es = (std::max(n_corner, 1) + 31) & ~31, ps = (std::max(n_surf, 1) + 31) & ~31;
    const size_t need = 8 * ((size_t)9 * es + (size_t)7 * ps);
    if (need > c->soa_cap) { hipFree(c->d_soa); c->d_soa = nullptr; c->soa_cap = 0; VMCHK(hipMalloc(&c->d_soa, 2 * need)); c->soa_cap = 2 * need; }
    if (!c->d_cnt) { VMCHK(hipMalloc(&c->d_cnt, 16)); VMCHK(hipHostMalloc((void**)&c->h_cnt, 16, hipHostMallocDefault)); }
    dl->edge_soa = c->d_soa; dl->edge_stride = es; dl->plane_soa = c->d_soa + (size_t)9 * es; dl->plane_stride = ps;
    if (nq == 0) return VM_OK;
    double* d_slot = (double*)c->d_work; int* d_nn = (int*)(c->d_work + 80 * (size_t)nq); float* d_nd5 = (float*)(d_nn + 10 * (size_t)nq); float* d_nint = d_nd5 + nq; int* d_blk = (int*)(c->d_work + (((size_t)164 * nq + 7) & ~(size_t)7));
    hipLaunchKernelGGL(k_map_search, dim3((nq + VM_QPB - 1) / VM_QPB), dim3(64 * VM_QPB), 0, c->stream, n_corner, n_surf, c->d_scan, T, (const PoseD*)nullptr, c->nc, c->gc.G, c->gc.order, c->gc.cxyz,
                       c->ns, c->gs.G, c->gs.order, c->gs.cxyz, d_nn, d_nd5, c->d_smap, d_nint);
    hipLaunchKernelGGL(k_map_fit, dim3((nq + VM_THREADS - 1) / VM_THREADS), dim3(VM_THREADS), 0, c->stream, n_corner, n_surf, c->d_scan, c->d_cmap, c->d_smap, d_nn, d_nd5, d_nint, d_slot, d_blk);
    hipLaunchKernelGGL(k_map_compact, dim3((nq + VM_THREADS - 1) / VM_THREADS), dim3(VM_THREADS), 0, c->stream, n_corner, n_surf, d_slot, d_blk, c->d_soa, es, c->d_soa + (size_t)9 * es, ps, c->d_cnt);
    VMCHK(hipMemcpyAsync(c->h_cnt, c->d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    VMCHK(hipStreamSynchronize(c->stream));
    VMCHK(hipGetLastError());
    *n_edge = c->h_cnt[0]; *n_plane = c->h_cnt[1];
    return VM_OK;
}

// The whole registration in one submission: both rounds of {search, fit, compact, one-launch pose solve} are enqueued back
// to back -- the counts, the factor tables and the pose between the rounds never leave the device -- and one read-back of the
// two result records ends the call.
#define REG_POSE 0
#define REG_RT 64
#define REG_OUT 192
#define REG_BYTES (REG_OUT + 2 * sizeof(vp1::Pose1Out))
int align_fused(vmap_ctx* c, int n_corner, int n_surf, double* q, double* t, const vil_options* opts, vmap_summary* out) {
    const auto t0 = std::chrono::steady_clock::now();
    const int nq = n_corner + n_surf;
    const int es = (std::max(n_corner, 1) + 31) & ~31, ps = (std::max(n_surf, 1) + 31) & ~31;
    const size_t need = 8 * ((size_t)9 * es + (size_t)7 * ps);
    if (need > c->soa_cap) { hipFree(c->d_soa); c->d_soa = nullptr; c->soa_cap = 0; VMCHK(hipMalloc(&c->d_soa, 2 * need)); c->soa_cap = 2 * need; }
    if (!c->d_cnt) { VMCHK(hipMalloc(&c->d_cnt, 16)); VMCHK(hipHostMalloc((void**)&c->h_cnt, 16, hipHostMallocDefault)); }
    if (!c->d_reg) { VMCHK(hipMalloc(&c->d_reg, REG_BYTES)); VMCHK(hipHostMalloc((void**)&c->h_reg, REG_BYTES, hipHostMallocDefault)); }
    if (!c->d_coop) { VMCHK(hipMalloc(&c->d_coop, sizeof(vp1::Pose1Coop))); VMCHK(hipMemsetAsync(c->d_coop, 0, sizeof(vp1::Pose1Coop), c->stream)); }
    if (!c->h_res && !getenv("VIL_NO_POLL")) {
        if (hipHostMalloc((void**)&c->h_res, sizeof(vp1::Pose1Out) + 64, hipHostMallocMapped) == hipSuccess) {
            memset(c->h_res, 0, sizeof(vp1::Pose1Out) + 64);
            if (hipHostGetDevicePointer(&c->d_res, c->h_res, 0) != hipSuccess) { hipHostFree(c->h_res); c->h_res = nullptr; c->d_res = nullptr; }
        } else c->h_res = nullptr;
    }
    std::unique_lock<std::shared_mutex> coop_lock(vilcoop::gate(c->device));                     // k_pose_solve's workgroups wait for one another (vil_coop.hpp)
    const int G = std::min(std::min(VP1_MAXG, c->coop_cap), std::max(1, (nq + VP1_THREADS - 1) / VP1_THREADS));
    double* edge_soa = c->d_soa; double* plane_soa = c->d_soa + (size_t)9 * es;
    double* h_pose = (double*)(c->h_reg + REG_POSE);
    h_pose[0] = t[0]; h_pose[1] = t[1]; h_pose[2] = t[2]; h_pose[3] = q[0]; h_pose[4] = q[1]; h_pose[5] = q[2]; h_pose[6] = q[3];
    vp1::Pose1In pin0, pin1; memset(&pin0, 0, sizeof pin0); memset(&pin1, 0, sizeof pin1);
    for (int k = 0; k < 7; ++k) pin0.pose[k] = h_pose[k];
    pin0.use = 1;                                        // the starting pose rides in the first solve's arguments (no 56-byte upload)
    int seq_expect = 0;
    volatile int* hseq = c->h_res ? (volatile int*)(c->h_res + sizeof(vp1::Pose1Out)) : nullptr;
    if (hseq) *hseq = -1;
    PoseD T; quat_to_R(q, T.R); T.t[0] = t[0]; T.t[1] = t[1]; T.t[2] = t[2];
    double* d_slot = (double*)c->d_work; int* d_nn = (int*)(c->d_work + 80 * (size_t)nq); float* d_nd5 = (float*)(d_nn + 10 * (size_t)nq); float* d_nint = d_nd5 + nq; int* d_blk = (int*)(c->d_work + (((size_t)164 * nq + 7) & ~(size_t)7));
    vp1::Pose1Out* d_out = (vp1::Pose1Out*)(c->d_reg + REG_OUT);
    for (int round = 0; round < 2; ++round) {
        if (nq > 0) {
            if (c->profiling && round == 0) VMCHK(hipEventRecord(c->ev[0], c->stream));
            hipLaunchKernelGGL(k_map_search, dim3((nq + VM_QPB - 1) / VM_QPB), dim3(64 * VM_QPB), 0, c->stream, n_corner, n_surf, c->d_scan, T, round ? (const PoseD*)(c->d_reg + REG_RT) : (const PoseD*)nullptr,
                               c->nc, c->gc.G, c->gc.order, c->gc.cxyz, c->ns, c->gs.G, c->gs.order, c->gs.cxyz, d_nn, d_nd5, c->d_smap, d_nint);
            if (c->profiling && round == 0) VMCHK(hipEventRecord(c->ev[1], c->stream));
            hipLaunchKernelGGL(k_map_fit, dim3((nq + VM_THREADS - 1) / VM_THREADS), dim3(VM_THREADS), 0, c->stream, n_corner, n_surf, c->d_scan, c->d_cmap, c->d_smap, d_nn, d_nd5, d_nint, d_slot, d_blk);
            if (c->profiling && round == 0) VMCHK(hipEventRecord(c->ev[2], c->stream));
            hipLaunchKernelGGL(k_map_compact, dim3((nq + VM_THREADS - 1) / VM_THREADS), dim3(VM_THREADS), 0, c->stream, n_corner, n_surf, d_slot, d_blk, edge_soa, es, plane_soa, ps, c->d_cnt);
        } else VMCHK(hipMemsetAsync(c->d_cnt, 0, 8, c->stream));
        hipLaunchKernelGGL(vp1::k_pose_solve, dim3(G), dim3(VP1_THREADS), 0, c->stream, c->d_cnt, edge_soa, es, plane_soa, ps, (double*)(c->d_reg + REG_POSE), (PoseD*)(c->d_reg + REG_RT), *opts,
                           round ? d_out : (const vp1::Pose1Out*)nullptr, d_out + round, c->d_coop, c->reg_epoch, round ? pin1 : pin0,
                           (round && c->d_res) ? (vp1::Pose1Out*)c->d_res : (vp1::Pose1Out*)nullptr, (round && c->d_res) ? (int*)((char*)c->d_res + sizeof(vp1::Pose1Out)) : (int*)nullptr);
        if (round) seq_expect = c->reg_epoch + 1;
        c->reg_epoch += std::min(std::max(opts->max_iterations, 0), VP1_MAX_ITER) + 8;     // one epoch per evaluation: at most max_iterations + 1
        if (c->reg_epoch > (1 << 30)) { c->reg_epoch = 0; VMCHK(hipMemsetAsync(c->d_coop, 0, sizeof(vp1::Pose1Coop), c->stream)); }   // flags are compared by order: start over from zeroed flags
    }
    bool polled = false;
    if (hseq && !c->profiling) {                         // the second solve's last act is the record + its sequence word in pinned memory
        const auto tp0 = std::chrono::steady_clock::now();
        for (long spin = 1;; ++spin) {
            if (*hseq == seq_expect) { polled = true; break; }
            if ((spin & 0xfff) == 0 && std::chrono::steady_clock::now() - tp0 > std::chrono::milliseconds(500)) break;
        }
        if (polled) { std::atomic_thread_fence(std::memory_order_acquire); memcpy(c->h_reg + REG_OUT + sizeof(vp1::Pose1Out), c->h_res, sizeof(vp1::Pose1Out)); }
    }
    if (!polled) {
        VMCHK(hipMemcpyAsync(c->h_reg + REG_OUT, d_out, 2 * sizeof(vp1::Pose1Out), hipMemcpyDeviceToHost, c->stream));
        VMCHK(hipStreamSynchronize(c->stream));
    }
    VMCHK(hipGetLastError());
    if (c->profiling && nq > 0) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) { c->prof_ms[0] += ms; c->prof_n[0]++; }
        if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) { c->prof_ms[1] += ms; c->prof_n[1]++; }
    }
    const vp1::Pose1Out* r = (const vp1::Pose1Out*)(c->h_reg + REG_OUT);
    if (r[1].status != 0) return r[1].status;
    t[0] = r[1].pose[0]; t[1] = r[1].pose[1]; t[2] = r[1].pose[2]; q[0] = r[1].pose[3]; q[1] = r[1].pose[4]; q[2] = r[1].pose[5]; q[3] = r[1].pose[6];
    out->rounds = 2; out->n_edge = r[1].n_edge; out->n_plane = r[1].n_plane; out->iterations = r[1].iterations; out->initial_cost = r[1].initial_cost; out->final_cost = r[1].final_cost;
    out->t_solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();   // one submission: association and solve are not separable on the host clock
    return VM_OK;
}
}  // namespace

extern "C" {

int vmap_create(int32_t device, vmap_ctx** out) {
    if (!out) return VM_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VM_ERR_DEVICE;      // no CPU fallback
    VMCHK(hipSetDevice(device));
    vmap_ctx* c = new vmap_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return VM_ERR_DEVICE; }
    *out = c;
    return VM_OK;
}
void vmap_destroy(vmap_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipFree(c->d_cmap); hipFree(c->d_smap); hipFree(c->gc.ws); hipFree(c->gs.ws); hipFree(c->d_scan); hipFree(c->d_work); if (c->h_slot) hipHostFree(c->h_slot); hipFree(c->d_soa); hipFree(c->d_cnt); if (c->h_cnt) hipHostFree(c->h_cnt); hipFree(c->d_reg); if (c->h_reg) hipHostFree(c->h_reg); if (c->h_res) hipHostFree(c->h_res); hipFree(c->d_coop); if (c->h_scan) hipHostFree(c->h_scan);
    for (hipEvent_t e : c->ev) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int vmap_set_map(vmap_ctx* c, int32_t nc, const float* corner, int32_t ns, const float* surf) {
    if (!c || nc < 0 || ns < 0 || (nc && !corner) || (ns && !surf)) return VM_ERR_INVALID;
    VMCHK(hipSetDevice(c->device));
    c->nc = 0; c->ns = 0;
    if (16 * (size_t)nc > c->cmap_cap) { hipFree(c->d_cmap); c->d_cmap = nullptr; c->cmap_cap = 0; VMCHK(hipMalloc(&c->d_cmap, 24 * (size_t)nc)); c->cmap_cap = 24 * (size_t)nc; }
    if (16 * (size_t)ns > c->smap_cap) { hipFree(c->d_smap); c->d_smap = nullptr; c->smap_cap = 0; VMCHK(hipMalloc(&c->d_smap, 24 * (size_t)ns)); c->smap_cap = 24 * (size_t)ns; }
    if (nc) VMCHK(hipMemcpyAsync(c->d_cmap, corner, 16 * (size_t)nc, hipMemcpyHostToDevice, c->stream));
    if (ns) VMCHK(hipMemcpyAsync(c->d_smap, surf, 16 * (size_t)ns, hipMemcpyHostToDevice, c->stream));
    if (nc) VMCHK(vknn::grid_build_adaptive(c->gc, nc, c->d_cmap, 4, c->hc, VM_OCC, c->stream));
    if (ns) VMCHK(vknn::grid_build_adaptive(c->gs, ns, c->d_smap, 4, c->hs, VM_OCC, c->stream));
    VMCHK(hipStreamSynchronize(c->stream));
    c->nc = nc; c->ns = ns;
    return VM_OK;
}

int vmap_associate(vmap_ctx* c, int32_t n_corner, const float* corner, int32_t n_surf, const float* surf, const double* q, const double* t,
                   int32_t* n_edge, double* edge9, int32_t* n_plane, double* plane7) {
    if (!c || !q || !t || !n_edge || !n_plane || n_corner < 0 || n_surf < 0 || (n_corner && (!corner || !edge9)) || (n_surf && (!surf || !plane7))) return VM_ERR_INVALID;
    VMCHK(hipSetDevice(c->device));
    const int st = upload_scan(c, n_corner, corner, n_surf, surf);
    if (st != VM_OK) return st;
    return associate_uploaded(c, n_corner, n_surf, q, t, n_edge, edge9, n_plane, plane7);
}

int vmap_set_fused_max(vmap_ctx* c, int32_t max_points) { if (!c || max_points < 0) return VM_ERR_INVALID; c->fused_max = max_points; return VM_OK; }

int vmap_profile_enable(vmap_ctx* c, int32_t enable) {
    if (!c) return VM_ERR_INVALID;
    VMCHK(hipSetDevice(c->device));
    if (enable && !c->ev[0]) for (hipEvent_t& e : c->ev) VMCHK(hipEventCreate(&e));
    c->profiling = enable != 0;
    return VM_OK;
}
int vmap_profile_read(vmap_ctx* c, int64_t* launches2, double* total_ms2) {
    if (!c || !launches2 || !total_ms2) return VM_ERR_INVALID;
    for (int k = 0; k < 2; ++k) { launches2[k] = c->prof_n[k]; total_ms2[k] = c->prof_ms[k]; c->prof_n[k] = 0; c->prof_ms[k] = 0.0; }
    return VM_OK;
}

int vmap_align(vmap_ctx* c, vil_ctx* solver, int32_t n_corner, const float* corner, int32_t n_surf, const float* surf,
               double* q, double* t, const vil_options* opts, vmap_summary* out) {
    if (!c || !solver || !q || !t || !opts || !out) return VM_ERR_INVALID;
    memset(out, 0, sizeof *out);
    if (!(c->nc > 10 && c->ns > 50)) return VM_OK;                      // localMapping.cpp:586 "corner and surf num are not enough"
    if (n_corner < 0 || n_surf < 0 || (n_corner && !corner) || (n_surf && !surf)) return VM_ERR_INVALID;
    VMCHK(hipSetDevice(c->device));
    int st = upload_scan(c, n_corner, corner, n_surf, surf);             // the scan is uploaded once for both rounds
    if (st != VM_OK) return st;
    if (c->coop_cap < 0) c->coop_cap = vilcoop::capacity((const void*)vp1::k_pose_solve, VP1_THREADS, 0, c->device);
    if (n_corner + n_surf <= c->fused_max && c->coop_cap >= 1) return align_fused(c, n_corner, n_surf, q, t, opts, out);
    for (int round = 0; round < 2; ++round) {
        int32_t ne = 0, np = 0;
        const auto ta = std::chrono::steady_clock::now();
        vil_device_lidar dl;
        st = associate_device(c, n_corner, n_surf, q, t, &ne, &np, &dl);   // the factor tables stay on the device, in the solver's layout
        if (st != VM_OK) return st;
        out->t_associate_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count();
        // one-pose window: pose free, everything else constant, identity LiDAR extrinsic, only the point factors
        vil_problem p; memset(&p, 0, sizeof p);
        uint8_t pose_const = 0, sb_const = 1;
        p.K = 1; p.L = 0; p.pose_const = &pose_const; p.sb_const = &sb_const; p.ex_const = 1; p.td_const = 1; p.use_td = 0;
        p.n_edge = ne; p.n_plane = np;
        p.q_lb[3] = 1.0; p.sqrt_info_px = 230.0; p.G[2] = 9.8;
        double pose[7] = {t[0], t[1], t[2], q[0], q[1], q[2], q[3]}, sb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ex[7] = {0, 0, 0, 0, 0, 0, 1}, td = 0.0, lam = 0.0;
        vil_state s; memset(&s, 0, sizeof s);
        s.K = 1; s.L = 0; s.pose = pose; s.speedbias = sb; s.ex_pose = ex; s.td = &td; s.inv_depth = &lam;
        vil_summary sum;
        st = vil_solve_device_lidar(solver, &p, &dl, c->stream, &s, opts, &sum);
        if (st != VIL_OK) return st;
        t[0] = pose[0]; t[1] = pose[1]; t[2] = pose[2]; q[0] = pose[3]; q[1] = pose[4]; q[2] = pose[5]; q[3] = pose[6];
        out->t_prepare_ms += sum.t_prepare_ms; out->t_solve_ms += sum.t_solve_ms + sum.t_readback_ms;
        out->rounds = round + 1; out->n_edge = ne; out->n_plane = np; out->iterations = sum.iterations; out->initial_cost = sum.initial_cost; out->final_cost = sum.final_cost;
    }
    return VM_OK;
}

}  // extern "C"
