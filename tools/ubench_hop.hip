// Flag hand-off between two workgroups of one launch (gfx950): round-trip time of a ping-pong through memory, (a) at agent scope -- what every hand-off of the one-launch
// iteration pays (sc1: the level the eight XCDs' L2s share) --, (b) with group-scope accesses (sc0) + an L1 invalidate per poll, which stay inside ONE XCD's L2 and are
// only coherent between workgroups of the same XCD.  Workgroups of a launch go to the XCDs round robin (block b -> XCD b % 8; each workgroup reports HW_REG_XCC_ID), so
// the pair (0, 8) shares an XCD and (0, 1) does not.  Every wait is bounded: a variant that is not coherent for its pair reports -1 instead of hanging the device.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_hop tools/ubench_hop.hip && /tmp/ubench_hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ unsigned long long wall() { unsigned long long t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
template <int V> __device__ __forceinline__ int ld(const int* p) {
    int v;
    if (V == 0) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (V == 1) asm volatile("buffer_inv sc0\n global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (V == 2) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (V == 3) asm volatile("buffer_inv sc1\n global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (V == 5) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_atomic_or %0, %1, %2, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(0) : "memory");      // V == 4: a returning read-modify-write (performed where atomics are performed)
    return v;
}
template <int V> __device__ __forceinline__ void st(int* p, int v) {
    if (V == 0 || V == 5) asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (V == 4) asm volatile("global_atomic_swap %0, %1, off" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dword %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
template <int V> __device__ __forceinline__ bool wait_eq(const int* p, int v, int* ab) {
    for (int sp = 0; sp < (1 << 15); ++sp) { if (ld<V>(p) == v) return true; __builtin_amdgcn_s_sleep(1); }
    __hip_atomic_store(ab, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}
// payload: the receiver also loads `nd` doubles per lane the sender stored in front of the flag (0: the bare flag)
template <int V> __global__ __launch_bounds__(64) void k_hop(int* flags, double* data, int a, int b, int rounds, int nd, long long* out, int* xcc, int* ab) {
    const int blk = blockIdx.x, t = threadIdx.x;
    if (t == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[blk] = (int)(id & 15); }
    if (blk != a && blk != b) return;
    int* fa = flags; int* fb = flags + 64;      // separate cache lines
    double acc = 0.0;
    const unsigned long long t0 = wall();
    bool ok = true;
    for (int r = 1; r <= rounds && ok; ++r) {
        if (blk == a) {
            for (int q = 0; q < nd; ++q) { const double v = r + q; if (V == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(data + q * 64 + t), "v"(v) : "memory"); else if (V == 5) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(data + q * 64 + t), "v"(v) : "memory"); else asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(data + q * 64 + t), "v"(v) : "memory"); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t == 0) st<V>(fa, r);
            ok = wait_eq<V>(fb, r, ab);
        } else {
            ok = wait_eq<V>(fa, r, ab);
            for (int q = 0; q < nd; ++q) { double v; if (V == 0 || V == 5) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(data + q * 64 + t) : "memory"); else asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(data + q * 64 + t) : "memory"); acc += v - (r + q); }
            if (t == 0) st<V>(fb, r);
        }
    }
    const unsigned long long t1 = wall();
    if (t == 0 && blk == a) { out[0] = ok ? (long long)(t1 - t0) : -1; }
    if (blk == b && acc != 0.0 && t == 0) out[1] = 1;      // stale payload seen
}
int main(int argc, char** argv) {
    // argv[1]: offset (kB) of the flags / payload inside a 64 MB buffer -- where the lines' home channel lies relative to the XCDs differs with the address
    int* flags; double* data; long long* out; int* xcc; int* ab;
    const size_t off = argc > 1 ? (size_t)atol(argv[1]) * 1024 : 0;
    char* big; hipMalloc(&big, 64u << 20); flags = (int*)(big + off); data = (double*)(big + off + 4096); printf("offset %zu kB\n", off >> 10); hipMalloc(&out, 64); hipMalloc(&xcc, 4 * 256); hipMalloc(&ab, 4);
    const int rounds = 400;
    std::vector<int> hx(256);
    const int pairs[4][2] = {{0, 8}, {0, 1}, {0, 4}, {8, 16}};
    for (int nd : {0, 4}) for (int v : {0, 3, 5}) for (auto& pr : pairs) {
        hipMemset(flags, 0, 4096); hipMemset(out, 0, 64); hipMemset(ab, 0, 4); hipMemset(data, 0, 8 * 64 * 64);
        if (v == 0) hipLaunchKernelGGL(k_hop<0>, dim3(64), dim3(64), 0, 0, flags, data, pr[0], pr[1], rounds, nd, out, xcc, ab);
        if (v == 1) hipLaunchKernelGGL(k_hop<1>, dim3(64), dim3(64), 0, 0, flags, data, pr[0], pr[1], rounds, nd, out, xcc, ab);
        if (v == 3) hipLaunchKernelGGL(k_hop<3>, dim3(64), dim3(64), 0, 0, flags, data, pr[0], pr[1], rounds, nd, out, xcc, ab);
        if (v == 5) hipLaunchKernelGGL(k_hop<5>, dim3(64), dim3(64), 0, 0, flags, data, pr[0], pr[1], rounds, nd, out, xcc, ab);
        if (v == 4) hipLaunchKernelGGL(k_hop<4>, dim3(64), dim3(64), 0, 0, flags, data, pr[0], pr[1], rounds, nd, out, xcc, ab);
        if (v == 2) hipLaunchKernelGGL(k_hop<2>, dim3(64), dim3(64), 0, 0, flags, data, pr[0], pr[1], rounds, nd, out, xcc, ab);
        if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 1; }
        long long ho[2]; hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), xcc, 4 * 64, hipMemcpyDeviceToHost);
        printf("payload %d doubles/lane  variant %d (%s)  blocks (%d, %d) on XCDs (%d, %d): %s", nd, v, v == 0 ? "agent scope sc1" : (v == 1 ? "sc0 + buffer_inv sc0 per poll" : (v == 2 ? "sc0, no invalidate" : (v == 3 ? "sc0 + buffer_inv sc1 per poll" : (v == 5 ? "plain payload stores, sc1 flag and loads" : "flag by atomic or / swap, payload sc0")))),
               pr[0], pr[1], hx[pr[0]], hx[pr[1]], ho[0] < 0 ? "NOT coherent (gave up)" : "");
        if (ho[0] >= 0) printf("%.0f ns per round trip (two hand-offs)%s", 10.0 * ho[0] / rounds, ho[1] ? "  STALE PAYLOAD" : "");
        printf("\n");
    }
    printf("XCD of blocks 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", hx[i]); printf("\n");
    return 0;
}
