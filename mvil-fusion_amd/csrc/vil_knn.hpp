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
// ---- uniform-grid search for larger clouds: count points per cell (hash table of packed cell keys), exclusive scan,
//      scatter into cell order, then every query walks Chebyshev rings of cells around its own cell until the k-th best
//      distance is provably final (everything unvisited is at least ring * h away).  Exact, like the tiled search.
struct GridTab { long long* keys; int* cnt; int* start; int* cur; int mask; float h; };
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
__global__ void k_grid_count(int n, const float* __restrict__ xyz, int stride, GridTab G, int* __restrict__ pslot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = grid_slot(G, cell_key(xyz[stride * i], xyz[stride * i + 1], xyz[stride * i + 2], G.h, 0, 0, 0), true);
    pslot[i] = s;
    atomicAdd(G.cnt + s, 1);
}
// single-workgroup exclusive scan of the per-slot counts (the table has at most a few 100 k slots)
__global__ __launch_bounds__(1024) void k_grid_scan(GridTab G) {
    __shared__ int part[1024];
    const int t = threadIdx.x, cap = G.mask + 1, per = (cap + 1023) / 1024;
    int s = 0;
    for (int e = t * per; e < min(cap, (t + 1) * per); ++e) s += G.cnt[e];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { const int v = t >= o ? part[t - o] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    int run = part[t] - s;
    for (int e = t * per; e < min(cap, (t + 1) * per); ++e) { G.start[e] = run; run += G.cnt[e]; }
}
__global__ void k_grid_fill(int n, const float* __restrict__ xyz, int stride, GridTab G, const int* __restrict__ pslot, int* __restrict__ order, float* __restrict__ cxyz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = pslot[i];
    const int pos = G.start[s] + atomicAdd(G.cur + s, 1);
    order[pos] = i; cxyz[3 * pos] = xyz[stride * i]; cxyz[3 * pos + 1] = xyz[stride * i + 1]; cxyz[3 * pos + 2] = xyz[stride * i + 2];
}

// the k-th best distance of a list (select chain: no dynamic register index)
__device__ __forceinline__ float knn_kth(const KnnList& L, int kk) {
    float bk = 3.0e38f;
#pragma unroll
    for (int m = 0; m < KNN_MAX; ++m) bk = (m == kk - 1) ? L.bd[m] : bk;
    return bk;
}

#define KNN_RMAX 6
// Exact k nearest neighbours of (qx, qy, qz) in a gridded cloud: Chebyshev rings of cells around the query's cell until the
// k-th best distance is below ring * h (everything unvisited is farther); isolated queries fall back to an exhaustive scan.
__device__ __forceinline__ void knn_grid_query(KnnList& L, float qx, float qy, float qz, int kk, int n, const GridTab& G, const int* __restrict__ order, const float* __restrict__ cxyz) {
    knn_init(L);
    bool done = false;
    for (int r = 0; r <= KNN_RMAX && !done; ++r) {
        for (int dx = -r; dx <= r; ++dx) for (int dy = -r; dy <= r; ++dy) for (int dz = -r; dz <= r; ++dz) {
            if (max(max(abs(dx), abs(dy)), abs(dz)) != r) continue;
            const int s = grid_slot(G, cell_key(qx, qy, qz, G.h, dx, dy, dz), false);
            if (s < 0) continue;
            const int b = G.start[s], e = b + G.cnt[s];
            for (int j = b; j < e; ++j) knn_insert(L, sqdist_nofma(qx, qy, qz, cxyz[3 * j], cxyz[3 * j + 1], cxyz[3 * j + 2]), order[j]);
        }
        const float bound = (float)r * G.h;
        done = r >= 1 && knn_kth(L, kk) < bound * bound;
    }
    if (!done) {
        knn_init(L);
        for (int j = 0; j < n; ++j) knn_insert(L, sqdist_nofma(qx, qy, qz, cxyz[3 * j], cxyz[3 * j + 1], cxyz[3 * j + 2]), order[j]);
    }
}

// host: device work space + launches that build the grid of a device-resident cloud (stride floats per point)
struct GridBuild { char* ws = nullptr; GridTab G; int* order = nullptr; float* cxyz = nullptr; int n = 0; };
inline hipError_t grid_build(GridBuild& gb, int n, const float* d_xyz, int stride, float h, hipStream_t stream) {
    if (gb.ws) { hipFree(gb.ws); gb.ws = nullptr; }
    int cap = 1024; while (cap < 2 * n) cap <<= 1;
    const size_t bytes = 8 * (size_t)cap + 3 * 4 * (size_t)cap + 2 * 4 * (size_t)n + 12 * (size_t)n + 256;
    hipError_t e = hipMalloc(&gb.ws, bytes);
    if (e != hipSuccess) return e;
    GridTab& G = gb.G;
    G.keys = (long long*)gb.ws; G.cnt = (int*)(gb.ws + 8 * (size_t)cap); G.start = G.cnt + cap; G.cur = G.start + cap; G.mask = cap - 1; G.h = h;
    int* pslot = G.cur + cap; gb.order = pslot + n; gb.cxyz = (float*)(gb.order + n); gb.n = n;
    hipMemsetAsync(G.keys, 0xFF, 8 * (size_t)cap, stream);                  // every key = -1 (empty)
    hipMemsetAsync(G.cnt, 0, 3 * 4 * (size_t)cap, stream);
    const int nblk = (n + 255) / 256;
    hipLaunchKernelGGL(k_grid_count, dim3(nblk), dim3(256), 0, stream, n, d_xyz, stride, G, pslot);
    hipLaunchKernelGGL(k_grid_scan, dim3(1), dim3(1024), 0, stream, G);
    hipLaunchKernelGGL(k_grid_fill, dim3(nblk), dim3(256), 0, stream, n, d_xyz, stride, G, pslot, gb.order, gb.cxyz);
    return hipGetLastError();
}

}  // namespace vknn
