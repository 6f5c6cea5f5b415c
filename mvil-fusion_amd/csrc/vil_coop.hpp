// Persistent kernels whose workgroups wait for one another (k_vgicp_align, k_pose_solve; in k_step the master and its helper workgroups and, in
// the merged gather + step launch, the W W^T tile workgroups) only finish if ALL of their waiting workgroups are resident at once -- k_step's finite
// roles (chain, gather) carry the lowest block indices and are dispatched before the workgroups that wait for them.
// Two guards, shared by the three translation units of the library:
//   capacity(): how many workgroups of a kernel the device can hold at the same time (occupancy x compute units THIS PROCESS has: compute_units() below) -- the grid of a
//               persistent launch is clamped to it, and a kernel that cannot be resident at all takes its multi-launch fallback;
//   gate(dev):  one mutex PER DEVICE, held from the launch of a spinning kernel until its result has arrived, so that two of them
//               (two contexts, two host threads on one device) never sit half-resident waiting for CUs the other one holds; contexts on
//               different devices do not contend.  vil_solve_resident takes it for the duration of an un-sharded solve.  The ranks of ANY
//               communicator (in-process, peer-buffer, RCCL) do not take it: a rank waits for its peers' launches inside the per-iteration
//               collective, and a peer that is a thread of the same process (or shares the device in a test) would be blocked on the gate
//               the waiting rank holds.  Sharded solves never use the merged launch, so what spins there is the master + its helpers only.
// Every other kernel of the library is finite: it can delay a persistent launch, never starve it.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <shared_mutex>
#include <algorithm>

namespace vilcoop {
// How many compute units does this PROCESS really have?  hipDeviceAttributeMultiprocessorCount and the occupancy query do not see a CU mask (HSA_CU_MASK /
// ROC_GLOBAL_CU_MASK, a partitioned or shared device: measured, profiles/r06_cu_mask.txt -- 256 reported under a 32-unit mask), and a persistent launch sized for units
// that are not there waits for workgroups that never run.  Probed once per device and process: n one-wave workgroups that each own a compute unit (100 kB of dynamic
// LDS) count themselves in, wait until all n are in or 1 ms has passed, and count themselves out -- the largest number inside at once is the number of units.
// An unmasked device answers in microseconds (everybody arrives at once); a masked one in (n / units) ms.
template <int LDS_KB>
__global__ __launch_bounds__(64) void k_cu_probe(int* live, int* most, int n) {
    extern __shared__ char probe_lds[];
    if (threadIdx.x != 0) return;
    probe_lds[0] = 1;
    const int me = __hip_atomic_fetch_add(live, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    __hip_atomic_fetch_max(most, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(most, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n && wall_clock64() - t0 < 100000ull) __builtin_amdgcn_s_sleep(8);      // 1 ms of the 100 MHz clock
    __hip_atomic_fetch_add(live, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
inline int compute_units(int device) {
    static std::mutex mu; static int cache[64]; static bool have[64];
    std::lock_guard<std::mutex> lk(mu);
    const unsigned d = (unsigned)device & 63u;
    if (have[d]) return cache[d];
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int* buf = nullptr;
    int got = cus;
    if (cus > 0 && hipMalloc((void**)&buf, 8) == hipSuccess) {
        constexpr int KB = 100;
        if (hipMemset(buf, 0, 8) == hipSuccess && hipFuncSetAttribute((const void*)k_cu_probe<KB>, hipFuncAttributeMaxDynamicSharedMemorySize, KB * 1024) == hipSuccess) {
            hipLaunchKernelGGL(k_cu_probe<KB>, dim3(cus), dim3(64), KB * 1024, 0, buf, buf + 1, cus);
            int h[2] = {0, 0};
            if (hipMemcpy(h, buf, 8, hipMemcpyDeviceToHost) == hipSuccess && h[1] > 0) got = std::min(cus, h[1]);
        }
        (void)hipGetLastError();
        hipFree(buf);
    } else (void)hipGetLastError();
    cache[d] = got; have[d] = true;
    return got;
}
// gate(dev): EXCLUSIVE (std::unique_lock) for a launch whose whole grid must be resident (k_solve, k_pose_solve, k_vgicp_align); SHARED (std::shared_lock) for the
// solves of a vil_solve_batch -- one-launch iterations, a handful of waiting workgroups each, whose sum the batch checks against the device (vilsolve.hip)
inline std::shared_mutex& gate(int device) { static std::shared_mutex m[64]; return m[(unsigned)device & 63u]; }          // (inline function: one instance per shared object)
inline int capacity(const void* func, int threads, size_t dyn_lds, int device) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, func, threads, dyn_lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return per_cu * compute_units(device);
}
}  // namespace vilcoop
