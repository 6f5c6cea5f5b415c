// vil_knn.hpp -- exact k-nearest-neighbour building blocks shared by the point-cloud rows (vilvgicp.hip, vilmap.hip):
// a register-resident sorted candidate list and a uniform grid built with atomics only.
#pragma once
#include <hip/hip_runtime.h>

namespace vknn {

__host__ __device__ inline long long pack_key(int x, int y, int z) { return ((long long)(x & 0x1FFFFF) << 42) | ((long long)(y & 0x1FFFFF) << 21) | (long long)(z & 0x1FFFFF); }
__host__ __device__ inline unsigned hash_key(long long k) { unsigned long long h = (unsigned long long)k * 0x9E3779B97F4A7C15ull; return (unsigned)(h >> 32); }

#define KNN_MAX 20
struct KnnList { float bd[KNN_MAX]; int bi[KNN_MAX]; };
__device__ __forceinline__ void knn_init(KnnList& L) {
#pragma unroll
    for (int m = 0; m < KNN_MAX; ++m) { L.bd[m] = 3.0e38f; L.bi[m] = 0x7fffffff; }
}
// insert candidate (d, j) into the list sorted by (distance, index): unrolled, branch-free, static register indices.
// The lexicographic order makes the result independent of the order in which candidates arrive.
__device__ __forceinline__ void knn_insert(KnnList& L, float d, int j) {
    if (d < L.bd[KNN_MAX - 1] || (d == L.bd[KNN_MAX - 1] && j < L.bi[KNN_MAX - 1])) {
#pragma unroll
        for (int m = KNN_MAX - 1; m >= 1; --m) {
            const bool up = d < L.bd[m - 1] || (d == L.bd[m - 1] && j < L.bi[m - 1]);
            const bool here = !up && (d < L.bd[m] || (d == L.bd[m] && j < L.bi[m]));
            L.bd[m] = up ? L.bd[m - 1] : (here ? d : L.bd[m]);
            L.bi[m] = up ? L.bi[m - 1] : (here ? j : L.bi[m]);
        }
        if (d < L.bd[0] || (d == L.bd[0] && j < L.bi[0])) { L.bd[0] = d; L.bi[0] = j; }
    }
}
__device__ __forceinline__ float sqdist_nofma(float qx, float qy, float qz, float x, float y, float z) {
    const float dx = __fsub_rn(qx, x), dy = __fsub_rn(qy, y), dz = __fsub_rn(qz, z);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));        // no fma: bit-equal to the CPU sum
}
// ---- uniform grid for larger clouds: count points per cell (hash table of packed cell keys), one atomic range claim per
//      occupied cell, scatter into cell order; a query (knn_wave_query below) then walks Chebyshev rings of cells around its
//      own cell until the k-th best distance is provably final (everything unvisited is at least ring * h away).  Exact.
struct GridTab { long long* keys; int* cnt; int* start; int* cur; int* nocc; int mask; float h; };
__device__ __forceinline__ int grid_slot(const GridTab& G, long long key, bool insert) {
    for (unsigned hh = hash_key(key) & G.mask;; hh = (hh + 1) & G.mask) {
        long long k = G.keys[hh];
        if (k == key) return (int)hh;
        if (k < 0) {
            if (!insert) return -1;
            k = (long long)atomicCAS((unsigned long long*)(G.keys + hh), (unsigned long long)-1LL, (unsigned long long)key);
            if (k < 0 || k == key) return (int)hh;
        }
    }
}
__device__ __forceinline__ long long cell_key(float x, float y, float z, float h, int dx, int dy, int dz) {
    return pack_key((int)floorf(x / h) + dx, (int)floorf(y / h) + dy, (int)floorf(z / h) + dz);
}
static __global__ void k_grid_count(int n, const float* __restrict__ xyz, int stride, GridTab G, int* __restrict__ pslot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = grid_slot(G, cell_key(xyz[stride * i], xyz[stride * i + 1], xyz[stride * i + 2], G.h, 0, 0, 0), true);
    pslot[i] = s;
    atomicAdd(G.cnt + s, 1);
}
// cell offsets: every occupied slot claims a contiguous range of the cell-ordered arrays with one atomic (the order of the
// cells in memory is irrelevant -- only the points of one cell must be contiguous -- so no scan is needed); G.nocc[0] counts
// the occupied cells (the host adapts the cell size to it), G.nocc[1] is the running total
static __global__ void k_grid_offsets(GridTab G) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e > G.mask) return;
    const int c = G.cnt[e];
    if (c) { G.start[e] = atomicAdd(G.nocc + 1, c); atomicAdd(G.nocc, 1); }
}
static __global__ void k_grid_fill(int n, const float* __restrict__ xyz, int stride, GridTab G, const int* __restrict__ pslot, int* __restrict__ order, float* __restrict__ cxyz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = pslot[i];
    const int pos = G.start[s] + atomicAdd(G.cur + s, 1);
    order[pos] = i; cxyz[3 * pos] = xyz[stride * i]; cxyz[3 * pos + 1] = xyz[stride * i + 1]; cxyz[3 * pos + 2] = xyz[stride * i + 2];
}

#define KNN_RMAX 6     // rings of cells searched before the exhaustive fallback
// ---- wave-per-query search: the 64 lanes of a wave hold the best candidates as ONE sorted list (lane i = i-th smallest
//      64-bit key (float distance bits << 32 | index): unsigned order == the lexicographic (distance, index) order of KnnList),
//      candidates are gathered 64 at a time with coalesced loads, and a candidate that beats the current k-th is inserted by
//      a single DPP shift + select across the wave.  Control flow is wave-uniform.
#define KNN_WL_CAP 512          // per-wave LDS work list (ints): candidate positions of one pass over up to 64 cells
__device__ __forceinline__ unsigned long long knn_key(float d, int j) { return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)j; }
__device__ __forceinline__ float knn_key_d(unsigned long long k) { return __uint_as_float((unsigned)(k >> 32)); }
__device__ __forceinline__ unsigned long long wave_shr1_u64(unsigned long long v) {     // lane i <- lane i-1, lane 0 <- 0
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)v, 0x138, 0xf, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(v >> 32), 0x138, 0xf, 0xf, false);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
// every lane offers one candidate key (~0 = none); all 64 lanes must be active
__device__ __forceinline__ void knn_wave_offer(unsigned long long& best, unsigned long long key, int kk) {
    unsigned long long thr = readlane_u64(best, kk - 1);
    unsigned long long m = __ballot(key < thr);
    if (thr == ~0ull && __builtin_popcountll(m) > kk) {
        // The list is not full yet (first batch of a query): every candidate passes the k-th best test and the serial loop below would take
        // k (1 + ln(n / k)) insertions to find the k smallest of n.  Find instead the smallest 16-bit prefix P of the distance bits
        // (sign, exponent, seven mantissa bits; distances are >= 0, so bit order = value order) with at least k candidates at or below
        // it -- sixteen ballots -- and offer only those: a candidate above P is beaten by k others of this very batch, so the result is
        // unchanged, and the loop runs k (+ the few that share the prefix) times.
        const unsigned hi = (unsigned)(key >> 48);
        unsigned lo = 0u, up = 0xFFFFu;
        while (lo < up) {
            const unsigned mid = (lo + up) >> 1;
            if (__builtin_popcountll(__ballot(hi <= mid)) >= kk) up = mid; else lo = mid + 1u;
        }
        m &= __ballot(hi <= lo);
    }
    while (m) {
        const int l = __builtin_ctzll(m);
        const unsigned long long c = readlane_u64(key, l);
        const unsigned long long prev = wave_shr1_u64(best);
        best = c < best ? (c > prev ? c : prev) : best;
        thr = readlane_u64(best, kk - 1);
        m &= m - 1;
        m &= __ballot(key < thr);
    }
}
__device__ __forceinline__ unsigned long long knn_wave_cand(int t, int total, float qx, float qy, float qz, const int* pos, const int* __restrict__ order, const float* __restrict__ cxyz) {
    if (t >= total) return ~0ull;
    const int j = pos ? pos[t] : t;
    return knn_key(sqdist_nofma(qx, qy, qz, cxyz[3 * j], cxyz[3 * j + 1], cxyz[3 * j + 2]), order[j]);
}
// Exact kk nearest neighbours (kk <= 64) of a WAVE-UNIFORM query; result: lane i < kk holds the i-th best key in `best`.
// Returns false when the query is rejected early: everything unvisited is at least sqrt(reject_d2) away and the reject_k-th
// best is not below reject_d2 (the callers discard such queries anyway).  wl: KNN_WL_CAP ints of LDS owned by this wave.
__device__ __forceinline__ bool knn_wave_query(unsigned long long& best, float qx, float qy, float qz, int kk, int n, const GridTab& G, const int* __restrict__ order,
                                               const float* __restrict__ cxyz, int* wl, float reject_d2, int reject_k) {
    const int lane = threadIdx.x & 63;
    best = ~0ull;
    const int cx = (int)floorf(qx / G.h), cy = (int)floorf(qy / G.h), cz = (int)floorf(qz / G.h);
    for (int r = 1; r <= KNN_RMAX; ++r) {
        const int side = 2 * r + 1, ncell = side * side * side;
        for (int base = 0; base < ncell; base += 64) {
            const int e = base + lane;
            int b = 0, cnt = 0;
            if (e < ncell) {
                int dz = e % side - r, dy = (e / side) % side - r, dx = e / (side * side) - r;
                if (r == 1) {
                    // first ring: the query's own cell first, then its 6 face, 12 edge and 8 corner neighbours (two bits per offset in three
                    // words).  The candidates reach the sorted list in lane order, so the near cells set the k-th best distance and most
                    // of what the far cells hold fails the threshold test before the serial insertion (the result does not depend on order).
                    dx = (int)((0x2a802a95402551ull >> (2 * e)) & 3) - 1; dy = (int)((0x28282528251945ull >> (2 * e)) & 3) - 1; dz = (int)((0x22221862185615ull >> (2 * e)) & 3) - 1;
                }
                if (r == 1 || max(max(abs(dx), abs(dy)), abs(dz)) == r) {
                    const int s = grid_slot(G, pack_key(cx + dx, cy + dy, cz + dz), false);
                    if (s >= 0) { b = G.start[s]; cnt = G.cnt[s]; }
                }
            }
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); incl += lane >= o ? v : 0; }
            const int total = __builtin_amdgcn_readlane(incl, 63);
            if (total == 0) continue;
            if (total <= KNN_WL_CAP) {
                const int excl = incl - cnt;
                for (int u = 0; u < cnt; ++u) wl[excl + u] = b + u;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int t0 = 0; t0 < total; t0 += 64) knn_wave_offer(best, knn_wave_cand(t0 + lane, total, qx, qy, qz, wl, order, cxyz), kk);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else {
                for (int c = 0; c < 64; ++c) {
                    const int cb = __builtin_amdgcn_readlane(b, c), cc = __builtin_amdgcn_readlane(cnt, c);
                    for (int t0 = 0; t0 < cc; t0 += 64) knn_wave_offer(best, knn_wave_cand(cb + t0 + lane, cb + cc, qx, qy, qz, nullptr, order, cxyz), kk);
                }
            }
        }
        const float bound = (float)r * G.h, b2 = bound * bound;
        if (knn_key_d(readlane_u64(best, kk - 1)) < b2) return true;
        if (b2 >= reject_d2 && !(knn_key_d(readlane_u64(best, reject_k - 1)) < reject_d2)) return false;
    }
    best = ~0ull;
    for (int t0 = 0; t0 < n; t0 += 64) knn_wave_offer(best, knn_wave_cand(t0 + lane, n, qx, qy, qz, nullptr, order, cxyz), kk);
    return true;
}

// host: device work space + launches that build the grid of a device-resident cloud (stride floats per point)
struct GridBuild { char* ws = nullptr; size_t ws_bytes = 0; GridTab G; int* order = nullptr; float* cxyz = nullptr; int n = 0; };
inline hipError_t grid_build(GridBuild& gb, int n, const float* d_xyz, int stride, float h, hipStream_t stream) {
    int cap = 1024; while (cap < 2 * n) cap <<= 1;
    const size_t bytes = 8 * (size_t)cap + 3 * 4 * (size_t)cap + 2 * 4 * (size_t)n + 12 * (size_t)n + 512;
    if (bytes > gb.ws_bytes) {
        if (gb.ws) { hipFree(gb.ws); gb.ws = nullptr; gb.ws_bytes = 0; }
        const size_t want = bytes + bytes / 2;                              // head room: the next scans' maps are about this size
        hipError_t e = hipMalloc(&gb.ws, want);
        if (e != hipSuccess) return e;
        gb.ws_bytes = want;
    }
    GridTab& G = gb.G;
    G.keys = (long long*)gb.ws; G.cnt = (int*)(gb.ws + 8 * (size_t)cap); G.start = G.cnt + cap; G.cur = G.start + cap; G.mask = cap - 1; G.h = h;
    G.nocc = G.cur + cap;
    int* pslot = G.nocc + 64; gb.order = pslot + n; gb.cxyz = (float*)(gb.order + n); gb.n = n;
    hipMemsetAsync(G.keys, 0xFF, 8 * (size_t)cap, stream);                  // every key = -1 (empty)
    hipMemsetAsync(G.cnt, 0, 3 * 4 * (size_t)cap + 256, stream);
    const int nblk = (n + 255) / 256;
    hipLaunchKernelGGL(k_grid_count, dim3(nblk), dim3(256), 0, stream, n, d_xyz, stride, G, pslot);
    hipLaunchKernelGGL(k_grid_offsets, dim3((cap + 255) / 256), dim3(256), 0, stream, G);
    hipLaunchKernelGGL(k_grid_fill, dim3(nblk), dim3(256), 0, stream, n, d_xyz, stride, G, pslot, gb.order, gb.cxyz);
    return hipGetLastError();
}

// grid with a cell size that keeps about `target` points per occupied cell (27 cells -> a few batches of 64 candidates per
// wave query); the searches are exact for any cell size, this only bounds their cost.  h is updated in place, so a caller that
// keeps it across scans converges after the first build.
inline hipError_t grid_build_adaptive(GridBuild& gb, int n, const float* d_xyz, int stride, float& h, double target, hipStream_t stream) {
    for (int pass = 0; pass < 3; ++pass) {
        hipError_t e = grid_build(gb, n, d_xyz, stride, h, stream);
        if (e != hipSuccess) return e;
        int nocc = 0;
        if ((e = hipMemcpyAsync(&nocc, gb.G.nocc, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        const double occ = (double)n / (nocc > 0 ? nocc : 1);
        if (occ <= 2.0 * target && (occ >= 0.4 * target || h >= 1.0f)) break;
        float hn = (float)(h * sqrt(target / occ));
        hn = fminf(1.0f, fmaxf(0.125f, hn));
        if (hn == h) break;
        h = hn;
    }
    return hipSuccess;
}

}  // namespace vknn
