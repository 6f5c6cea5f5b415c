// Persistent kernels whose workgroups wait for one another (k_vgicp_align, k_pose_solve; in k_step the master and its helper workgroups and, in
// the merged gather + step launch, the W W^T tile workgroups) only finish if ALL of their waiting workgroups are resident at once -- k_step's finite
// roles (chain, gather) carry the lowest block indices and are dispatched before the workgroups that wait for them.
// Two guards, shared by the three translation units of the library:
//   capacity(): how many workgroups of a kernel the device can hold at the same time (occupancy x compute units) -- the grid of a
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

namespace vilcoop {
inline std::mutex& gate(int device) { static std::mutex m[64]; return m[(unsigned)device & 63u]; }          // (inline function: one instance per shared object)
inline int capacity(const void* func, int threads, size_t dyn_lds, int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, func, threads, dyn_lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return per_cu * cus;
}
}  // namespace vilcoop
