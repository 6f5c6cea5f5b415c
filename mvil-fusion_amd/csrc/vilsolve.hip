// libvilsolve.so -- C-ABI (include/vilsolve.h) over the gfx950 kernels.
// Replaces, for mVIL-Fusion's Estimator::optimization() (estimator.cpp:1124-1687), the ceres::Problem
// construction + ceres::Solve + MarginalizationInfo machinery.  No CPU fallback: every compute entry
// point needs a HIP device and returns VIL_ERR_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <mutex>
#include <thread>
#include <array>
#include <memory>
#include <pthread.h>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types only: RCCL is bound at run time (dlopen) so that a process which already
                         // carries an RCCL (PyTorch bundles one) never ends up with two copies

#include "../../include/vilsolve.h"
#include "vil_internal.h"
#include "vil_tuning.hpp"
#include "vil_coop.hpp"
#include "vil_dev.hpp"
#include "vil_finish.hpp"
#include "vil_sweep.hpp"
#include "vil_eval.hpp"
#include "vil_step.hpp"
#include "vil_iter.hpp"
#include "vil_marg.hpp"
#include "vil_window.hpp"

// vilpersist.hip: the persistent solve kernel k_solve<2 | 5> (compiled in its own translation unit: vil_math.hpp, VIL_OPAQUE_TID)
const void* vil_k_solve_fn(int vis_ts);
void vil_k_solve_launch(int vis_ts, unsigned grid, size_t lds, hipStream_t stream, const DevP& P, const SolveOpts& O, long long budget_ticks);

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[vilsolve] HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return VIL_ERR_DEVICE; } } while (0)

namespace {

struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    bool load() {
        if (h) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }     // reuse a copy that is already mapped
        if (!h) for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
        if (!h) return false;
        GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
        AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
        AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        return GetUniqueId && CommInitRank && AllReduce && AllGather && CommDestroy;
    }
};
RcclApi g_rccl;

// One device allocation per uploaded window: [tables | work space].  The tables (everything the caller hands over) are
// assembled in a PINNED host image and go up in one DMA; the work space (partial records, systems, step vectors) exists on the
// device only and is cleared by a memset -- it used to travel as zeros through a pageable staging vector.
struct Arena {
    char* h = nullptr; size_t hcap = 0, hsize = 0;     // pinned host image of the tables
    char* d = nullptr; size_t cap = 0;                 // device
    size_t ssize = 0;                                  // work-space bytes
    bool grow(size_t need) {
        if (need <= hcap) return true;
        size_t ncap = hcap ? hcap : (size_t)1 << 20;
        while (ncap < need) ncap *= 2;
        char* nh = nullptr;
        if (hipHostMalloc((void**)&nh, ncap, hipHostMallocDefault) != hipSuccess) return false;
        if (hsize) memcpy(nh, h, hsize);
        if (h) hipHostFree(h);
        h = nh; hcap = ncap;
        return true;
    }
    // returns (size_t)-1 when the pinned image cannot grow
    size_t take(size_t bytes) { const size_t o = (hsize + 255) & ~size_t(255); if (!grow(o + bytes)) return (size_t)-1; if (o > hsize) memset(h + hsize, 0, o - hsize); hsize = o + bytes; return o; }
    size_t take_scratch(size_t bytes) { const size_t o = (ssize + 255) & ~size_t(255); ssize = o + bytes; return o; }
    void reset() { hsize = 0; ssize = 0; }
};

}  // namespace

// In-process communicator: the contexts of ONE process (one host thread each) sum their buffers through device
// memory.  Same call sites and the same deterministic result on every rank as the RCCL path; used where the ranks share
// a process, and by the tests to run a sharded solve on a single device.
struct LocalComm {
    int n = 0;
    pthread_barrier_t bar;
    double* ptr[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int err[8] = {0, 0, 0, 0, 0, 0, 0, 0};        // per-rank status of the current collective: an error on one rank is an error on all (nobody is left at a barrier)
    ~LocalComm() { if (n) pthread_barrier_destroy(&bar); }
    // every rank posts its status, waits, and reads the worst one: collective error exits
    int agree(int rank, int st) { err[rank] = st; pthread_barrier_wait(&bar); int w = 0; for (int r = 0; r < n; ++r) w = std::min(w, err[r]); pthread_barrier_wait(&bar); return w; }
};
struct PeerPtrs { const double* p[8]; int n; };

// Multi-process exchange without RCCL (the per-iteration message is latency-bound; on the fully connected xGMI mesh a one-shot
// "everybody writes to everybody, then sums locally" beats a ring): every rank owns an INBOX in its device memory -- world x 2 (parity of
// the collective's sequence number) x cap doubles, then one flag per (source rank, parity), 64 bytes apart -- exported with hipIpcGetMemHandle
// and opened by every peer.  A collective = k_ipc_push (my message into every inbox, a system-scope fence, then the flags) ->
// k_ipc_wait (one workgroup spins until every source's flag carries the sequence number) -> k_ipc_sum (rank order: identical bits on every
// rank).  Everything is in stream order and the sequence number lives in device memory: capturable in a hipGraph, nothing for the host to wait for.
// Two parities suffice: a peer can only push collective i + 2 after it has summed i + 1, which needs MY push of i + 1, which I do after my sum of i.
struct IpcComm {
    int world = 0, rank = 0; size_t cap = 0;
    bool ready = false;                       // every peer's handle is open (vil_comm_ipc_init): before that a collective would write through null inbox pointers
    char* base = nullptr;                     // my allocation: [inbox | flags | seq | count]
    void* peer_base[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // opened handles (my own: base)
    size_t off_flags = 0, off_seq = 0;
    double* inbox(int r) const { return (double*)peer_base[r]; }
    int* flags(int r) const { return (int*)((char*)peer_base[r] + off_flags); }
    int* seq() const { return (int*)(base + off_seq); }
};
struct IpcPtrs { double* inbox[8]; int* flags[8]; int world, rank; size_t cap; int* seq; int* count; };
// sym = D > 0: the message starts with a symmetric D x D block (the reduced system S', both triangles filled with the same bits by the gather): only
// its lower triangle travels, the sum writes both mirror images -- identical bits, D (D - 1) / 2 doubles less per rank pair (98 of 484 kB at K = 10)
__device__ __forceinline__ bool sym_skip(size_t e, int sym) { if (!sym || e >= (size_t)sym * sym) return false; const int i = (int)(e / sym), j = (int)(e - (size_t)i * sym); return j > i; }
__device__ __forceinline__ void sym_store(double* out, size_t e, int sym, double v) {
    out[e] = v;
    if (sym && e < (size_t)sym * sym) { const int i = (int)(e / sym), j = (int)(e - (size_t)i * sym); if (j < i) out[(size_t)j * sym + i] = v; }
}
// Owned segments of the per-iteration message [camera part | h_ll, b_l, 1/pivot, scale (L each) | e_A (13 L) | e_O (6 F)]: a landmark's entries are
// non-zero on ONE rank, its owner (contiguous landmark / factor ranges, vil_shard_ranges) -- "sum over the ranks" of that part is "take the owner's value"
// (x + 0 + ... + 0 = x: the same bits).  The library's own exchanges therefore move the camera part of every rank but only the OWNED slice of the landmark
// arrays: per peer 103 + 380 / world kB instead of 390 kB at K = 10 / 1000 landmarks (150 kB at world = 8), and the summing kernel reads one inbox for them.
struct OwnSeg { size_t cam; int Lp, n; int lb[9], fb[9]; };      // n = 0: a plain sum of everything (agreements; RCCL beyond eight ranks)
__device__ __forceinline__ int seg_owner(const OwnSeg& S, size_t e) {          // -1: summed over all ranks
    if (S.n == 0 || e < S.cam) return -1;
    const size_t a = e - S.cam, Lp = (size_t)S.Lp;
    const bool lm = a < 17 * Lp;
    const int v = a < 4 * Lp ? (int)(a % Lp) : (lm ? (int)((a - 4 * Lp) / 13) : (int)((a - 17 * Lp) / 6));
    const int* b = lm ? S.lb : S.fb;
    int r = 0;
    while (r + 1 < S.n && v >= b[r + 1]) ++r;
    return r;
}
__global__ void k_ipc_push(const double* send, size_t cnt, IpcPtrs I, int sym, OwnSeg S) {
    const int sq = *I.seq + 1, par = sq & 1;
    const size_t off = ((size_t)I.rank * 2 + par) * I.cap;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += (size_t)gridDim.x * blockDim.x) {
        if (sym_skip(e, sym)) continue;
        const int own = seg_owner(S, e);
        if (own >= 0 && own != I.rank) continue;          // somebody else's landmark: zero here, nothing to send
        const double v = send[e];
        for (int r = 0; r < I.world; ++r) I.inbox[r][off + e] = v;
    }
    __threadfence_system();                   // this thread's stores have reached every peer ...
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) last = atomicAdd(I.count, 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (last && threadIdx.x == 0) {           // ... all blocks' have: the flags go out, the sequence number advances
        __threadfence_system();
        *I.count = 0;
        for (int r = 0; r < I.world; ++r) __hip_atomic_store(I.flags[r] + 16 * (I.rank * 2 + par), sq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        *I.seq = sq;
    }
}
__global__ void k_ipc_wait(IpcPtrs I) {
    const int sq = *I.seq, par = sq & 1, t = threadIdx.x;       // (k_ipc_push of this collective has advanced it: same stream)
    if (t < I.world) while (__hip_atomic_load(I.flags[I.rank] + 16 * (t * 2 + par), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != sq) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_ipc_sum(double* out, size_t cnt, IpcPtrs I, int sym, OwnSeg S) {
    const int par = *I.seq & 1;
    const double* in = I.inbox[I.rank];
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += (size_t)gridDim.x * blockDim.x) {
        if (sym_skip(e, sym)) continue;
        const int own = seg_owner(S, e);
        double s = 0;
        if (own >= 0) s = __builtin_nontemporal_load(in + ((size_t)own * 2 + par) * I.cap + e);                        // the owner's entry (everybody else's is zero and did not travel)
        else for (int r = 0; r < I.world; ++r) s += __builtin_nontemporal_load(in + ((size_t)r * 2 + par) * I.cap + e);      // written by peers: not through a stale cache line
        sym_store(out, e, sym, s);
    }
}
__global__ void k_sum_peers(double* out, PeerPtrs pp, size_t cnt, int sym, OwnSeg S) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += (size_t)gridDim.x * blockDim.x) {
        if (sym_skip(e, sym)) continue;
        const int own = seg_owner(S, e);
        double s = 0;
        if (own >= 0) s = pp.p[own][e];
        else for (int r = 0; r < pp.n; ++r) s += pp.p[r][e];     // rank order: identical bits on every rank
        sym_store(out, e, sym, s);
    }
}

// The same slim message for RCCL ("[upper(S), g, cost]", ~100 kB): the per-iteration set is PACKED into M = [lower triangle of S' | g, b_c, diag | cost]
// (D (D + 1) / 2 + 3 D + 4 doubles, summed by ONE ncclAllReduce) and G = this rank's slice of the landmark arrays [h_ll, b_l, 1/pivot, scale | e_A | e_O], padded
// to the largest slice (ONE ncclAllGather: a landmark's entries live on its owner only, so the sum over ranks is the owner's value -- nothing to add), and
// UNPACKED into set 1 with both mirror images of S'.  At K = 10 / 1000 landmarks / world 8 RCCL moves 103 + 48 kB per rank instead of all-reducing 484 kB.
// staging (doubles): [M camS | G gmax | Msum camS | Gall world x gmax]; pack / unpack do not know the transport -- the tests run them over the in-process
// communicator with k_slim_emul standing in for the two RCCL calls (vil_debug_set_slim_emul), on 2 / 3 / 8 ranks of one device.
struct SlimLay { int D, Lp, n, rank; size_t cam, camS, gmax, span; int lb[9], fb[9]; };
__device__ __forceinline__ size_t slim_tri(int i, int j) { return (size_t)i * (i + 1) / 2 + j; }
// position of set entry e (e >= cam) inside its owner's slice; *owner receives the rank
__device__ __forceinline__ size_t slim_slot(const SlimLay& Y, size_t e, int* owner) {
    const size_t a = e - Y.cam, Lp = (size_t)Y.Lp;
    const bool lm = a < 17 * Lp;
    const int v = a < 4 * Lp ? (int)(a % Lp) : (lm ? (int)((a - 4 * Lp) / 13) : (int)((a - 17 * Lp) / 6));
    const int* b = lm ? Y.lb : Y.fb;
    if (v >= b[Y.n]) { *owner = -1; return 0; }      // padding of an empty array (Lp = max(L, 1), Fp = max(n_vis, 1) with L or n_vis = 0): nobody owns it, it stays zero
    int r = 0;
    while (r + 1 < Y.n && v >= b[r + 1]) ++r;
    *owner = r;
    const size_t nL = (size_t)(Y.lb[r + 1] - Y.lb[r]);
    if (a < 4 * Lp) return (a / Lp) * nL + (size_t)(v - Y.lb[r]);
    if (lm) return 4 * nL + (a - 4 * Lp) - 13 * (size_t)Y.lb[r];
    return 17 * nL + (a - 17 * Lp) - 6 * (size_t)Y.fb[r];
}
__global__ void k_slim_pack(const double* set0, double* M, double* G, SlimLay Y) {
    const size_t DD = (size_t)Y.D * Y.D, trin = (size_t)Y.D * (Y.D + 1) / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < Y.span; e += (size_t)gridDim.x * blockDim.x) {
        if (e < DD) { const int i = (int)(e / Y.D), j = (int)(e - (size_t)i * Y.D); if (j <= i) M[slim_tri(i, j)] = set0[e]; }
        else if (e < Y.cam) M[trin + (e - DD)] = set0[e];
        else { int r; const size_t g = slim_slot(Y, e, &r); if (r == Y.rank) G[g] = set0[e]; }
    }
}
__global__ void k_slim_unpack(double* set1, const double* Msum, const double* Gall, SlimLay Y) {
    const size_t DD = (size_t)Y.D * Y.D, trin = (size_t)Y.D * (Y.D + 1) / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < Y.span; e += (size_t)gridDim.x * blockDim.x) {
        double v;
        if (e < DD) { const int i = (int)(e / Y.D), j = (int)(e - (size_t)i * Y.D); v = Msum[slim_tri(max(i, j), min(i, j))]; }
        else if (e < Y.cam) v = Msum[trin + (e - DD)];
        else { int r; const size_t g = slim_slot(Y, e, &r); v = r >= 0 ? Gall[(size_t)r * Y.gmax + g] : 0.0; }
        set1[e] = v;
    }
}
// stand-in for ncclAllReduce(M -> Msum) + ncclAllGather(G -> Gall) over the in-process communicator (tests): pp.p[r] = rank r's staging buffer
__global__ void k_slim_emul(double* Msum, double* Gall, PeerPtrs pp, SlimLay Y) {
    const size_t tot = Y.camS + (size_t)pp.n * Y.gmax;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        if (e < Y.camS) { double a = 0; for (int r = 0; r < pp.n; ++r) a += pp.p[r][e]; Msum[e] = a; }
        else { const size_t q = e - Y.camS; const int r = (int)(q / Y.gmax); Gall[q] = pp.p[r][Y.camS + (q - (size_t)r * Y.gmax)]; }
    }
}

#define VIL_MAX_CHUNK 24       // iterations enqueued without a host round trip (the first solve of an upload: vil_solve_resident); vil_profile_enable sizes its events for it
#define VIL_CHC_MAX 16384      // entries of the chain workgroup's gather table (K = 20: ~7000)
#define VIL_SFLAG_MAX 4096      // sweep workgroups a one-launch iteration may have (configs[2]: ~600)
struct vil_ctx {
    int device = 0, rank = 0, world = 1;
    hipStream_t stream = nullptr;
    Arena ar;
    hipEvent_t dep_ev = nullptr;                           // orders the solve after a producer stream (vil_solve_device_lidar)
    hipEvent_t up_ev = nullptr; bool up_pending = false;   // the DMA out of the pinned image must finish before the image is rewritten
    DevP P;                        // device pointers
    bool uploaded = false;
    int resident_kind = 0;         // 1: a window handed over through vil_upload / vil_solve (the only kind vil_solve_resident accepts);
                                   // 2: a derived problem of vil_marginalize / vil_eval_factors / vil_linearize (they replace the resident window)
    int K = 0, L = 0, D = 0, NS = 0;
    size_t off_x0 = 0;             // backup of the uploaded state (device)
    double* d_x0 = nullptr;
    double* d_xsave = nullptr;        // the state the solve at hand started from (written by its init launch)
    int drop_role = -1, drop_launch = -1;      // vil_debug_drop_flag: armed for the next solve
    bool in_batch = false;         // this solve is one of a vil_solve_batch (shared gate, no persistent solve)
    int rung_fail_run[2] = {0, 0}, rung_cooldown[2] = {0, 0};      // launch-structure ladder (vil_solve_resident): consecutive give-ups of the persistent solve / the one-launch iteration, solves they sit out
    int64_t n_recovered = 0, n_aborted = 0;    // solves whose one-launch attempt gave up and were re-run with two launches per iteration / that failed on both
    bool reset_pending = false;       // vil_reset_state called, the copy not launched yet
    std::vector<int> plane_perm, edge_perm;   // sorted index -> caller index
    int n_blocks_sweep = 0, n_blocks_reduce = 0, n_blocks_reduce_po = 0, n_gather_m = 0, n_ww = 0;      // n_ww: tiles of W W^T formed by extra workgroups of k_reduce (vil_prechain.hpp)
    size_t lds_sweep = 0, lds_step = 0, lds_reduce = 0;
    bool persist = false; size_t lds_solve = 0; int cap_solve[2] = {-1, -1}; size_t cap_solve_lds[2] = {0, 0}; int attr_solve[2] = {0, 0};      // the whole solve in one resident launch (k_solve<2 | 5>, vil_iter.hpp)
    bool fused = false; size_t lds_iter = 0; int cap_iter[2] = {-1, -1}; size_t cap_iter_lds[2] = {0, 0}; int attr_iter[2] = {0, 0};      // the one-launch iteration (k_iter<2 | 5>, vil_iter.hpp)
    int cap_step3 = -1; size_t cap_step3_lds = 0;      // workgroups of the merged gather + step launch the device holds at once AT THAT dynamic-LDS size (vil_coop.hpp)
    int vis_gm = 0;                      // doubles of operand rows the largest visual chunk of the uploaded window needs (vil_sweep.hpp)
    size_t span = 0;               // doubles of one linear-system set (SysBuf::ar): the multi-GPU all-reduce message
    bool step_lds = false;
    int* d_status = nullptr;
    Ctl* h_ctl = nullptr;          // pinned
    int* h_word = nullptr;         // pinned: status words read back from the device
    char* h_mirror = nullptr; Ctl* d_hctl = nullptr; int* d_hseq = nullptr; double* d_hstate = nullptr; size_t mirror_ns = 0;      // pinned + mapped: Ctl | sequence word | final state, written by solve_finish (vil_finish.hpp)
    bool no_poll = false;          // VIL_NO_POLL=1: copy + synchronise instead of polling the mirror
    bool mirror_state = false;     // the mirror holds the final state of the last solve (vil_download_state needs no device operation)
    int attr_sweep[2] = {0, 0}, attr_step[5] = {0, 0, 0, 0, 0}, attr_marg[2] = {0, 0}, attr_commit = 0;      // dynamic-LDS sizes already granted to the kernels (hipFuncSetAttribute is not free)
    double* h_pin = nullptr;       // pinned scratch
    char* marg_ws = nullptr;       // device work space of vil_marginalize (grow-only)
    unsigned long long* d_marg_ts = nullptr;      // phase stamps of the last marginalisation's kernels (vil_debug_marg_stamps)
    size_t marg_ws_bytes = 0;
    size_t h_pin_bytes = 0;
    std::vector<int> prior_joff;
    ncclComm_t comm = nullptr;     // RCCL communicator over xGMI (world > 1)
    std::shared_ptr<struct LocalComm> lcomm;   // in-process communicator (vil_comm_init_local)
    std::shared_ptr<struct IpcComm> ipc;       // multi-process peer-buffer exchange (vil_comm_ipc_export / vil_comm_ipc_init)
    bool has_comm() const { return comm != nullptr || lcomm != nullptr || ipc != nullptr; }
    double* lc_tmp = nullptr; size_t lc_cap = 0;
    double* ipc_tmp = nullptr;
    double* slim_buf = nullptr; size_t slim_cap = 0; bool slim_emul = false;      // staging of the packed per-iteration message (RCCL; vil_debug_set_slim_emul)
    bool slim() const { return own.n > 0 && world > 1 && (comm != nullptr || (slim_emul && lcomm != nullptr)); }
    bool sharded = false;          // the resident problem is this rank's shard of the factor set
    // hipGraph of a chunk of iterations, reused by repeated solves of one upload (key: chunk length, options)
    struct ChunkGraph { int n; SolveOpts so; hipGraphExec_t exec; };
    std::vector<ChunkGraph> graphs;
    int use_graph = -1;            // VIL_GRAPH=0 disables (tuning build)
    int* d_imu_perm = nullptr;      // (vil_sweep.hpp, sweep_imu: entry order of the IMU roles)
    // (a few tables by key: a tracker alternates between the prior structures of its two marginalisation kinds, and a rebuild -- entries, sort, copy -- is ~80 us of host time)
    struct ChTab { std::vector<int> key; int* d = nullptr; int* h = nullptr; size_t cap = 0; int n = 0; hipEvent_t ev = nullptr; bool ev_pending = false; unsigned long long used = 0; };
    ChTab chtabs[4]; int chtab_cur = 0; unsigned long long chtab_clock = 0;      // gather table of the chain workgroup (vil_prechain.hpp)
    bool graph_failed = false;     // a chunk could not be captured / instantiated: direct launches from then on
    int fail_capture = 0;          // vil_debug_fail_graph_capture: that many captures are treated as failed
    int solve_gen = 0;             // generation counter of the helper-workgroup flags (Ctl::gen)
    int solves_since_upload = 0;
    bool split = false;            // sweep + gather fill set 0, the collective sums it into set 1, the step kernel reads set 1
    bool force_split = false;      // vil_debug_set_split: that plumbing on a single rank
    int launch_mode = 0;           // vil_debug_set_launch_mode: 0 = the library's choice, 1 = no merged launch, 2 = no chain workgroup, 3 = sweep + merged gather / step launch (no one-launch iteration)
    int last_live = 5;             // live sweep launches of the previous solve (sizes the first launch chunk)
    int recent_live[8] = {5, 5, 5, 5, 5, 5, 5, 5}; int recent_at = 0;      // ... of the last eight solves: the first solve of an upload (a tracker's image) is sized by their maximum
    int lm_b = 0, lm_e = 0;        // owned landmark range
    OwnSeg own = {0, 0, 0, {0}, {0}};      // every rank's landmark / factor range in the per-iteration message (sharded windows)
    // ---- window residency across frames (vil_lidar_*, vil_set_gauge_fix, vil_marginalize_resident) --------------------------------
    struct Slab { int np, ne, slot; int np_all, ne_all; };      // np_all / ne_all: the frame's points BEFORE a communicator's slicing (the same on every rank)
    std::vector<Slab> slabs;       // window order: slab i <-> pose K - count + i
    std::vector<int> free_slots;
    int cap_p = 0, cap_e = 0, nslot = 0;      // points per slab (capacity), physical slabs
    double* d_pl = nullptr; double* d_ed = nullptr;       // component-major: row q of the plane table at d_pl + q * nslot * cap_p
    double* d_lstage = nullptr; double* h_lstage = nullptr; size_t lstage_cap = 0;     // AoS staging of one pushed frame (device / pinned)
    hipEvent_t lpush_ev = nullptr; bool lpush_pending = false;
    bool lidar_resident = false;   // the resident problem takes its point factors from the slabs
    bool gauge_on = false;
    struct MargMeta {              // what vil_marginalize_resident needs to know about the resident window (captured at upload)
        std::vector<int> prior_kind, prior_index; bool has_prior = false;
        std::vector<char> obs0;    // frames observing a landmark anchored in frame 0
        int n_lm0 = 0; bool imu01 = false; bool lidar0 = false; bool use_td = false;
        std::vector<int> icp_ids, lps_ids;
    } mm;
    // ---- the fully resident window (vil_win_*, vil_window.hpp) ---------------------------------------------------------------------
    struct WinStore {
        bool open = false; vil_win_cfg cfg;
        int K = 0, T = 0, S = 0, NF = 0, nmax = 0, x0max = 0;
        std::vector<int> fslot, islot, free_f, free_i;           // window frame -> observation slot / IMU slot; free lists
        std::vector<int> ns; std::vector<double> sum_dt;          // per IMU slot: samples held, duration of the interval
        double* d_store = nullptr; double* d_samp = nullptr; double* d_hdr = nullptr; double* d_rec = nullptr; double* d_U = nullptr;
        double* d_prior[2] = {nullptr, nullptr}; int cur = 0;     // prior slots: [J0 nmax^2 | r0 nmax | x0 x0max | pH nmax^2 | pg0 nmax | pc0 8]
        int pn = 0, pm = 0; std::vector<int> pkind, pindex, pcol;  // the current prior's block structure (host)
        int* d_wstat = nullptr;                                   // sticky status of the asynchronous work (a covariance that is not positive definite, a prior that is not finite)
        char* h_stage = nullptr; char* d_stage = nullptr; size_t stage_cap = 0; hipEvent_t ev = nullptr; bool pending = false;
        double* dt(int q) const { return d_samp + (size_t)q * 7 * S; }
        double* acc(int q) const { return dt(q) + S; }
        double* gyr(int q) const { return dt(q) + 4 * (size_t)S; }
        double* pJ0(int w) const { return d_prior[w]; }
        double* pr0(int w) const { return d_prior[w] + (size_t)nmax * nmax; }
        double* px0(int w) const { return pr0(w) + nmax; }
        double* pH(int w) const { return px0(w) + x0max; }
        double* pg0(int w) const { return pH(w) + (size_t)nmax * nmax; }
        double* pc0(int w) const { return pg0(w) + nmax; }
        size_t prior_doubles() const { return 2 * (size_t)nmax * nmax + 2 * (size_t)nmax + x0max + 8; }
    } win;
    bool profiling = false, stamps = false; long long period_n = 0;
    std::vector<unsigned long long> last_stamps;      // raw stamps of the last profiled solve (vil_debug_read_stamps)
    int wg_launch = -1;      // (vil_profile_workgroups)
    long long* d_prof = nullptr; double phase_us[VIL_PROF_SLOTS] = {0}; long long phase_n = 0;      // phase stamps of the one-launch iterations (vil_profile_phases)
    std::vector<hipEvent_t> ev, ev_mid, ev_coll;
    vil_profile prof = {0, 0.0, 0, 0.0, 0.0};
};

// pinned, device-mapped mirror [Ctl | sequence word (64 B) | final state (ns doubles)]; grow-only
static int ensure_mirror(vil_ctx* c, size_t ns) {
    if (c->h_mirror && ns <= c->mirror_ns) return VIL_OK;
    if (c->stream) HIPCHK(hipStreamSynchronize(c->stream));       // nobody is writing the old one
    if (c->h_mirror) hipHostFree(c->h_mirror);
    c->h_mirror = nullptr; c->d_hctl = nullptr; c->d_hseq = nullptr; c->d_hstate = nullptr; c->mirror_ns = 0; c->mirror_state = false;
    const size_t cap = ns + ns / 2 + 256, bytes = sizeof(Ctl) + 64 + 8 * cap;
    HIPCHK(hipHostMalloc((void**)&c->h_mirror, bytes, hipHostMallocMapped));
    memset(c->h_mirror, 0, bytes);
    void* dp = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dp, c->h_mirror, 0));
    c->d_hctl = (Ctl*)dp; c->d_hseq = (int*)((char*)dp + sizeof(Ctl)); c->d_hstate = (double*)((char*)dp + sizeof(Ctl) + 64);
    c->mirror_ns = cap;
    return VIL_OK;
}

// static LDS of a kernel as the loaded code object reports it (StepShared is only part of what k_step carries: the gather and tile roles keep scratch arrays there)
static size_t step_static_lds(const void* fn) {
    // (a property of the code object: asked once per kernel and process -- hipFuncGetAttributes is a few microseconds of driver time, and every upload, i.e. every
    //  image of a tracker, asked five times)
    static std::mutex mu; static std::vector<std::pair<const void*, size_t>> known;
    std::lock_guard<std::mutex> lk(mu);
    for (const auto& e : known) if (e.first == fn) return e.second;
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess) { (void)hipGetLastError(); return (size_t)64 * 1024; }
    known.emplace_back(fn, (size_t)fa.sharedSizeBytes);
    return (size_t)fa.sharedSizeBytes;
}
static SolveOpts to_dev_opts(const vil_options* o) {
    SolveOpts s;
    memset(&s, 0, sizeof s);            // compared bytewise as the key of the cached iteration graphs
    s.max_iterations = o->max_iterations; s.jacobi_scaling = o->jacobi_scaling;
    s.visual_loss = o->visual_loss; s.lidar_loss = o->lidar_loss; s.rel_loss = o->rel_loss; s.autodiff_quirk = o->autodiff_quirk;
    s.precision = o->precision == 1 ? 1 : 0;
    s.function_tolerance = o->function_tolerance; s.gradient_tolerance = o->gradient_tolerance; s.parameter_tolerance = o->parameter_tolerance;
    s.max_radius = o->max_radius; s.min_relative_decrease = o->min_relative_decrease; s.min_mu = o->min_mu; s.max_mu = o->max_mu;
    s.visual_loss_scale = o->visual_loss_scale; s.lidar_loss_scale = o->lidar_loss_scale; s.rel_loss_scale = o->rel_loss_scale;
    return s;
}

static void quat_to_R_host(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}

// test hook (vil_debug_dense_solve): the step's dense factorisation + back substitution on a matrix of the caller's -- one workgroup, the tiled LDS layout of vil_step.hpp.
// RW: what the one-launch iteration runs (chol_dense / back_subst_cols); otherwise the look-ahead factorisation and the back substitution with inverted diagonal tiles.
template <int SLOTS, bool RW>
__global__ __launch_bounds__(VIL_STEP_THREADS) void k_debug_dense(const double* Ain /* (D+1) x (D+1) row major: lower triangle, last row = right-hand side */, double* Lout, double* xout, int* okout, int D) {
    extern __shared__ double dbgA[];
    __shared__ vd::StepShared s;
    const int t = threadIdx.x, R = D + 1, T = (R + 15) >> 4, NTL = ((T * (T + 1)) >> 1) * TILE_SZ;
    if (t == 0) { int g = 0; for (int I = 0; I < T; ++I) for (int J = 0; J <= I; ++J) { s.tI[g] = (unsigned char)I; s.tJ[g] = (unsigned char)J; ++g; } }
    for (int e = t; e < NTL; e += blockDim.x) dbgA[e] = 0.0;
    __syncthreads();
    for (int e = t; e < R * R; e += blockDim.x) { const int i = e / R, j = e - i * R; if (j <= i && j < D) dbgA[vd::tl_idx(i, j)] = Ain[e]; }
    __syncthreads();
    bool ok;
    if constexpr (RW) ok = vd::chol_dense<SLOTS>(dbgA, D, s); else ok = vd::chol_lookahead<SLOTS, false>(dbgA, D, s);
    __syncthreads();
    for (int e = t; e < R * R; e += blockDim.x) { const int i = e / R, j = e - i * R; Lout[e] = (j < i && j < D) ? dbgA[vd::tl_idx(i, j)] : ((j == i && i < D) ? 1.0 / s.dinv[i] : 0.0); }
    __syncthreads();
    if (ok) { if constexpr (RW) vd::back_subst_cols(dbgA, D, s); else vd::back_subst(dbgA, D, s); }
    __syncthreads();
    for (int e = t; e < D; e += blockDim.x) xout[e] = ok ? s.y[e] : 0.0;
    if (t == 0) okout[0] = ok ? 1 : 0;
}
__global__ void k_aos2soa(const double* aos, int n, int ncomp, double* dst, size_t stride) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n) for (int q = 0; q < ncomp; ++q) dst[(size_t)q * stride + f] = aos[(size_t)f * ncomp + q];
}
extern "C" {

int vil_abi_version(void) { return VIL_ABI_VERSION; }

const char* vil_strerror(int st) {
    switch (st) {
        case VIL_OK: return "ok";
        case VIL_ERR_INVALID_ARGUMENT: return "invalid argument";
        case VIL_ERR_DEVICE: return "HIP device error (no device / runtime failure)";
        case VIL_ERR_NON_FINITE: return "non-finite cost or state";
        case VIL_ERR_NOT_POSITIVE_DEFINITE: return "reduced system not positive definite";
        case VIL_ERR_COMM: return "RCCL error";
        case VIL_ERR_UNSUPPORTED: return "unsupported configuration";
    }
    return "unknown";
}

int vil_reduced_dim(int K) { return 15 * K + 7; }
void vil_prior_capacity(int K, int* n_max, int* nblk_max, int* x0_max) {
    if (n_max) *n_max = 6 * K + 16;
    if (nblk_max) *nblk_max = K + 4;
    if (x0_max) *x0_max = 7 * K + 9 + 7 + 1 + 16;
}
void vil_default_options(vil_options* o) {
    o->max_iterations = 30; o->max_time_s = 0.05;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_relative_decrease = 1e-3;
    o->min_mu = 1e-8; o->max_mu = 1.0; o->jacobi_scaling = 1;
    o->visual_loss = VIL_LOSS_CAUCHY; o->visual_loss_scale = 1.0;
    o->lidar_loss = VIL_LOSS_HUBER; o->lidar_loss_scale = 0.1;
    o->rel_loss = VIL_LOSS_CAUCHY; o->rel_loss_scale = 1.0;
    o->autodiff_quirk = 1; o->precision = 0;
}

int vil_create(const vil_device_cfg* cfg, vil_ctx** out) {
    if (!cfg || !out) return VIL_ERR_INVALID_ARGUMENT;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "[vilsolve] no HIP device: the hot path has no CPU fallback\n"); return VIL_ERR_DEVICE; }
    if (cfg->device < 0 || cfg->device >= ndev) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(cfg->device));
    vil_ctx* c = new vil_ctx();
    c->device = cfg->device; c->rank = cfg->rank; c->world = cfg->world > 0 ? cfg->world : 1;
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&c->d_status, sizeof(int)));
    HIPCHK(hipMemset(c->d_status, 0, sizeof(int)));
    HIPCHK(hipHostMalloc(&c->h_ctl, sizeof(Ctl), hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&c->h_word, 64, hipHostMallocDefault));      // small status reads land in PINNED memory (an asynchronous copy into pageable memory goes through the runtime's staging path)
    c->no_poll = getenv("VIL_NO_POLL") != nullptr;
    memset(&c->P, 0, sizeof c->P);
    *out = c;
    return VIL_OK;
}

void vil_destroy(vil_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto& e : c->ev) hipEventDestroy(e);
    for (auto& e : c->ev_mid) hipEventDestroy(e);
    for (auto& e : c->ev_coll) hipEventDestroy(e);
    for (auto& g : c->graphs) hipGraphExecDestroy(g.exec);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->ar.d) hipFree(c->ar.d);
    if (c->ar.h) hipHostFree(c->ar.h);
    if (c->up_ev) hipEventDestroy(c->up_ev);
    if (c->dep_ev) hipEventDestroy(c->dep_ev);
    if (c->d_status) hipFree(c->d_status);
    if (c->h_ctl) hipHostFree(c->h_ctl);
    if (c->h_word) hipHostFree(c->h_word);
    if (c->h_mirror) hipHostFree(c->h_mirror);
    if (c->h_pin) hipHostFree(c->h_pin);
    if (c->marg_ws) hipFree(c->marg_ws);
    if (c->lc_tmp) hipFree(c->lc_tmp);
    if (c->d_imu_perm) hipFree(c->d_imu_perm);
    for (auto& ct : c->chtabs) { if (ct.d) hipFree(ct.d); if (ct.h) hipHostFree(ct.h); if (ct.ev) hipEventDestroy(ct.ev); }
    if (c->ipc_tmp) hipFree(c->ipc_tmp);
    if (c->slim_buf) hipFree(c->slim_buf);
    if (c->d_prof) hipFree(c->d_prof);
    if (c->ipc) { for (int r = 0; r < c->ipc->world; ++r) if (r != c->ipc->rank && c->ipc->peer_base[r]) hipIpcCloseMemHandle(c->ipc->peer_base[r]); if (c->ipc->base) hipFree(c->ipc->base); c->ipc.reset(); }
    if (c->d_pl) hipFree(c->d_pl);
    if (c->d_ed) hipFree(c->d_ed);
    if (c->d_lstage) hipFree(c->d_lstage);
    if (c->h_lstage) hipHostFree(c->h_lstage);
    if (c->lpush_ev) hipEventDestroy(c->lpush_ev);
    { auto& w = c->win; hipFree(w.d_store); hipFree(w.d_samp); hipFree(w.d_hdr); hipFree(w.d_rec); hipFree(w.d_U); hipFree(w.d_prior[0]); hipFree(w.d_prior[1]); hipFree(w.d_wstat); hipFree(w.d_stage);
      if (w.h_stage) hipHostFree(w.h_stage); if (w.ev) hipEventDestroy(w.ev); }
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

static int validate(const vil_problem* p, const vil_state* s, bool device_lidar = false, bool win = false) {
    if (!p || !s) return VIL_ERR_INVALID_ARGUMENT;
    if (p->K < 1 || p->L < 0 || s->K != p->K || s->L != p->L) return VIL_ERR_INVALID_ARGUMENT;
    if (!s->pose || !s->speedbias || !s->ex_pose || !s->td || (p->L > 0 && !s->inv_depth)) return VIL_ERR_INVALID_ARGUMENT;
    if (15 * p->K + 7 > 320) return VIL_ERR_UNSUPPORTED;   // K <= 20 (step kernel work space)
    const bool res_lidar = p->n_plane == VIL_LIDAR_RESIDENT && p->n_edge == VIL_LIDAR_RESIDENT;
    if (res_lidar) device_lidar = true;          // the tables are the context's slabs: nothing of the caller's to check
    if (p->n_vis < 0 || p->n_imu < 0 || p->n_icp < 0 || p->n_lps < 0 || (!res_lidar && (p->n_plane < 0 || p->n_edge < 0)) || p->prior.n < 0) return VIL_ERR_INVALID_ARGUMENT;
    if (p->n_icp + p->n_lps > 12) return VIL_ERR_UNSUPPORTED;   // reference trims to 5 + 7 (estimator.cpp:1283-1286,1345-1348)
    if (p->prior.n > 512 || p->prior.nblk > 256) return VIL_ERR_UNSUPPORTED;
    // a table may be NULL only when its count is zero
    if (p->n_vis > 0 && (!p->vis_i || !p->vis_j || !p->vis_l || !p->vis_const)) return VIL_ERR_INVALID_ARGUMENT;
    if (p->n_imu > 0 && (!p->imu_i || !p->imu_j || (!p->imu_const && !win))) return VIL_ERR_INVALID_ARGUMENT;
    if (p->n_icp > 0 && (!p->icp_ids || !p->icp_const)) return VIL_ERR_INVALID_ARGUMENT;
    if (p->n_lps > 0 && (!p->lps_ids || !p->lps_const)) return VIL_ERR_INVALID_ARGUMENT;
    if (!device_lidar && p->n_plane > 0 && (!p->plane_pose || !p->plane_const)) return VIL_ERR_INVALID_ARGUMENT;
    if (!device_lidar && p->n_edge > 0 && (!p->edge_pose || !p->edge_const)) return VIL_ERR_INVALID_ARGUMENT;
    for (int f = 0; f < p->n_vis; ++f) {
        if (p->vis_l[f] < 0 || p->vis_l[f] >= p->L || p->vis_i[f] < 0 || p->vis_i[f] >= p->K || p->vis_j[f] < 0 || p->vis_j[f] >= p->K) return VIL_ERR_INVALID_ARGUMENT;
        if (p->vis_i[f] == p->vis_j[f]) return VIL_ERR_INVALID_ARGUMENT;     // the reference never builds such a factor (estimator.cpp:1205: `if (imu_i == imu_j) continue;`)
        if (f && p->vis_l[f] < p->vis_l[f - 1]) return VIL_ERR_INVALID_ARGUMENT;
        if (f && p->vis_l[f] == p->vis_l[f - 1] && p->vis_i[f] != p->vis_i[f - 1]) return VIL_ERR_INVALID_ARGUMENT;
    }
    if (!device_lidar) {
        for (int f = 0; f < p->n_plane; ++f) if (p->plane_pose[f] < 0 || p->plane_pose[f] >= p->K) return VIL_ERR_INVALID_ARGUMENT;
        for (int f = 0; f < p->n_edge; ++f) if (p->edge_pose[f] < 0 || p->edge_pose[f] >= p->K) return VIL_ERR_INVALID_ARGUMENT;
    }
    for (int f = 0; f < p->n_imu; ++f) if (p->imu_i[f] < 0 || p->imu_i[f] >= p->K || p->imu_j[f] < 0 || p->imu_j[f] >= p->K) return VIL_ERR_INVALID_ARGUMENT;
    for (int q = 0; q < 4 * p->n_icp; ++q) if (p->icp_ids[q] < 0 || p->icp_ids[q] >= p->K) return VIL_ERR_INVALID_ARGUMENT;
    for (int q = 0; q < 2 * p->n_lps; ++q) if (p->lps_ids[q] < 0 || p->lps_ids[q] >= p->K) return VIL_ERR_INVALID_ARGUMENT;
    if (p->prior.n > 0) {
        const vil_prior& pr = p->prior;
        if (pr.nblk <= 0 || !pr.blk_kind || !pr.blk_index || !pr.blk_col || (!win && (!pr.x0 || !pr.J0 || !pr.r0))) return VIL_ERR_INVALID_ARGUMENT;
        for (int b = 0; b < pr.nblk; ++b) {
            const int kind = pr.blk_kind[b], idx = pr.blk_index[b];
            if (kind < VIL_BLK_POSE || kind > VIL_BLK_TD) return VIL_ERR_INVALID_ARGUMENT;
            if ((kind == VIL_BLK_POSE || kind == VIL_BLK_SPEEDBIAS) && (idx < 0 || idx >= p->K)) return VIL_ERR_INVALID_ARGUMENT;
            const int ls = (kind == VIL_BLK_POSE || kind == VIL_BLK_EX) ? 6 : (kind == VIL_BLK_SPEEDBIAS ? 9 : 1);
            if (pr.blk_col[b] < 0 || pr.blk_col[b] + ls > pr.n) return VIL_ERR_INVALID_ARGUMENT;
        }
    }
    return VIL_OK;
}

// pose-sort LiDAR points, transpose to SoA (straight into `dst`, ncomp x stride doubles), build (start,count,pose) chunks of <= 256 points
static void lidar_order(int n, const int* pose, int K, std::vector<int>& perm, std::vector<int>& cnt) {
    perm.resize(n);
    cnt.assign(K + 1, 0);
    for (int f = 0; f < n; ++f) cnt[pose[f] + 1]++;
    for (int k = 0; k < K; ++k) cnt[k + 1] += cnt[k];
    std::vector<int> pos(cnt.begin(), cnt.end() - 1);
    for (int f = 0; f < n; ++f) perm[pos[pose[f]]++] = f;
}
static void lidar_soa(int n, int ncomp, const double* c, const std::vector<int>& perm, int stride, double* dst) {
    for (int q = 0; q < ncomp; ++q) {
        double* row = dst + (size_t)q * stride;
        for (int sidx = 0; sidx < n; ++sidx) row[sidx] = c[(size_t)perm[sidx] * ncomp + q];
        for (int sidx = n; sidx < stride; ++sidx) row[sidx] = 0.0;
    }
}
static void lidar_chunks(const std::vector<int>& cnt, int K, std::vector<int>& chunks) {
    chunks.clear();
    for (int k = 0; k < K; ++k) for (int s = cnt[k]; s < cnt[k + 1]; s += VIL_THREADS) { chunks.push_back(s); chunks.push_back(std::min(VIL_THREADS, cnt[k + 1] - s)); chunks.push_back(k); }
}

// ---- plan of the visual role (vil_sweep.hpp: sweep_visual), host logic without a device -------------------------------------------------------------
// The sweep walks the landmarks sorted by (first frame, last frame) -- insertion order in a tracker's feature list is already close to that -- and the
// factor tables are STORED in that order (sorted position <-> caller's factor: vfac / vfinv), so that a chunk's factors are consecutive rows of the SoA
// tables.  The sorted list is cut into chunks, one per workgroup, so that (a) a chunk fits the role's LDS (VIS_LM landmarks, VIS_MF factors, VIS_GM doubles of
// operand rows at the row stride of ITS window), (b) the matrix-core work of a chunk, (rows / 4) x tiles of its window, stays under a cap found by bisection:
// the smallest one that gives every chunk a compute unit of its own in the first round of the launch (vwg_max).  A chunk below VIL_VCHUNK_FBAL factors is
// not closed for balance (a workgroup's fixed cost is ~8 us whatever it holds).
struct VisPlan {
    std::vector<int> order, fperm, finv;                 // sorted landmarks (those with factors); sorted position -> caller's factor and back
    std::vector<int> vwg, vrec, vlm, vfac, wend;         // the device tables (vil_dev.hpp)
    size_t rec_doubles = 0; int n_chunks = 0, tmax = 1, gm = 0, cap = 0;
};
static bool plan_visual(const int K, const int L, const int n_vis, const std::vector<int>& lms, const std::vector<int>& fmin, const std::vector<int>& fmax, const std::vector<int>& anch,
                        const int vwg_max, const int cap_forced, VisPlan& o) {
    o.order.clear(); o.fperm.assign(std::max(n_vis, 1), 0); o.finv.assign(std::max(n_vis, 1), 0);
    {   // by (first frame, last frame, index): a counting sort over the K^2 frame pairs, stable in the index (a comparison sort of a tracker's ~1000 landmarks was 40 us of
        // every image's 110 us of host preparation)
        std::vector<int> cnt((size_t)K * K + 1, 0);
        int nlive = 0;
        for (int l = 0; l < L; ++l) if (lms[l + 1] > lms[l]) { ++cnt[(size_t)fmin[l] * K + fmax[l] + 1]; ++nlive; }
        for (size_t q = 0; q < (size_t)K * K; ++q) cnt[q + 1] += cnt[q];
        o.order.assign(nlive, 0);
        for (int l = 0; l < L; ++l) if (lms[l + 1] > lms[l]) o.order[cnt[(size_t)fmin[l] * K + fmax[l]]++] = l;
    }
    { int pos = 0; for (int l : o.order) for (int f = lms[l]; f < lms[l + 1]; ++f) { o.fperm[pos] = f; o.finv[f] = pos; ++pos; } }
    const std::vector<int>& order = o.order;
    struct Chunk { int p0, nl, nf, fa, span; };
    std::vector<Chunk> ch;
    auto cost = [](int nf, int T) { return (vd::vis_rows(nf) / 4 + 4) * vd::vis_ntile(T); };
    auto cut = [&](int cap) -> bool {
        ch.clear();
        Chunk cc{0, 0, 0, 0, 0}; int wlo = K, whi = -1;
        for (int q = 0; q < (int)order.size(); ++q) {
            const int l = order[q], n = lms[l + 1] - lms[l];
            const int lo = std::min(wlo, fmin[l]), hi = std::max(whi, fmax[l]), T = vd::vis_tiles(hi - lo + 1);
            const bool fits = cc.nl < VIS_LM && cc.nf + n <= VIS_MF && vd::vis_gm_doubles(cc.nf + n, T) <= VIS_GM;
            if (cc.nl > 0 && (!fits || (cc.nf >= VIL_VCHUNK_FBAL && cost(cc.nf + n, T) > cap))) {
                cc.fa = wlo; cc.span = whi - wlo + 1; ch.push_back(cc);
                cc = Chunk{q, 0, 0, 0, 0}; wlo = K; whi = -1;
            }
            if (cc.nl == 0 && (n > VIS_MF || vd::vis_gm_doubles(n, vd::vis_tiles(fmax[l] - fmin[l] + 1)) > VIS_GM)) return false;      // a landmark no chunk can hold
            cc.nl++; cc.nf += n; wlo = std::min(wlo, fmin[l]); whi = std::max(whi, fmax[l]);
        }
        if (cc.nl > 0) { cc.fa = wlo; cc.span = whi - wlo + 1; ch.push_back(cc); }
        return true;
    };
    int lo = cost(VIL_VCHUNK_FBAL, 1), hi = cost(VIS_MF, VIS_TMAX);
    if (cap_forced > 0) lo = hi = cap_forced;
    if (!cut(lo)) return false;
    o.cap = lo;
    if ((int)ch.size() > vwg_max) {
        while (lo < hi) { const int mid = (lo + hi) / 2; cut(mid); if ((int)ch.size() <= vwg_max) hi = mid; else lo = mid + 1; }
        cut(hi); o.cap = hi;
    }
    const int nw = (int)ch.size();
    o.n_chunks = nw;
    o.vwg.assign(8 * (size_t)std::max(nw, 1), 0); o.vrec.assign(4 * (size_t)std::max(nw, 1), 0); o.vlm.assign(4 * std::max(order.size(), (size_t)1), 0); o.vfac.assign(2 * (size_t)std::max(n_vis, 1), 0);
    size_t roff = 0; int fpos = 0;
    o.tmax = 1; o.gm = 0;
    for (int w = 0; w < nw; ++w) {
        const Chunk& cc = ch[w];
        const int T = vd::vis_tiles(cc.span);
        int* d = &o.vwg[8 * (size_t)w];
        d[0] = cc.p0; d[1] = cc.nl; d[2] = fpos; d[3] = cc.nf; d[4] = cc.fa; d[5] = cc.span; d[6] = T; d[7] = (int)(roff / 16);
        int* r = &o.vrec[4 * (size_t)w]; r[0] = d[7]; r[1] = cc.fa; r[2] = cc.span; r[3] = T;
        int floc = 0;
        for (int q = 0; q < cc.nl; ++q) {
            const int l = order[cc.p0 + q], n = lms[l + 1] - lms[l];
            int* e = &o.vlm[4 * (size_t)(cc.p0 + q)]; e[0] = l; e[1] = floc; e[2] = n; e[3] = anch[l];
            for (int k = 0; k < n; ++k) { o.vfac[2 * (size_t)(fpos + floc + k)] = lms[l] + k; o.vfac[2 * (size_t)(fpos + floc + k) + 1] = q; }
            floc += n;
        }
        fpos += cc.nf; roff += (size_t)vd::vis_rec_doubles(T);
        o.tmax = std::max(o.tmax, T); o.gm = std::max(o.gm, vd::vis_gm_doubles(cc.nf, T));
    }
    o.rec_doubles = roff;
    o.wend.assign(std::max(K, 1), 0);
    for (int w = 0; w < nw; ++w) for (int f = ch[w].fa; f < K; ++f) o.wend[f]++;
    return true;
}
// frame range and anchor of every landmark from the caller's factor tables
static void landmark_frames(const vil_problem* p, std::vector<int>& lms, std::vector<int>& fmin, std::vector<int>& fmax, std::vector<int>& anch) {
    const int L = p->L, K = p->K;
    lms.assign(L + 1, 0); fmin.assign(std::max(L, 1), K); fmax.assign(std::max(L, 1), -1); anch.assign(std::max(L, 1), 0);
    for (int f = 0; f < p->n_vis; ++f) lms[p->vis_l[f] + 1]++;
    for (int l = 0; l < L; ++l) lms[l + 1] += lms[l];
    for (int f = 0; f < p->n_vis; ++f) {
        const int l = p->vis_l[f], lo = std::min(p->vis_i[f], p->vis_j[f]), hi = std::max(p->vis_i[f], p->vis_j[f]);
        anch[l] = p->vis_i[f]; fmin[l] = std::min(fmin[l], lo); fmax[l] = std::max(fmax[l], hi);
    }
}
// helper workgroups of the step kernel for L landmarks: none below one landmark per master thread, else one per 128 (quad of lanes per landmark) or 256 (pair) landmarks, at most 15
static int vil_helpers_for(int L) {
    const int quad = VIL_STEP_THREADS / 4, pair = VIL_STEP_THREADS / 2;
    if (L < VIL_STEP_THREADS) return 0;
    if (L <= 15 * quad) return std::min(15, (L + quad - 1) / quad);
    return std::min(15, (L + pair - 1) / pair);
}
// Co-residency is decided per XCD: the dispatcher deals the workgroups of a grid round-robin to the eight XCDs, each of which places ITS share on ITS compute units.
// n consecutive workgroups that wait for others therefore need ceil(n / 8) slots on every XCD, and one more for the workgroups they wait for to run through
// (measured under a 32-unit mask, 4 units per XCD: 25 waiting workgroups fit the device's 32 slots and still starved the gather workgroups of one XCD).
static bool fits_per_xcd(int n_resident, int capacity) { return (n_resident + 7) / 8 + 1 <= capacity / 8; }
static int visual_wg_budget(int n_imu, int n_plane, int n_edge) { return std::max(64, 256 - (n_imu + 3 + (n_plane + 511) / 512 + (n_edge + 511) / 512)); }
// diagnostic surface of the plan (no device needed: the CPU test suite checks its invariants; DESIGN.md quotes its record sizes)
int vil_visual_plan(const vil_problem* p, vil_visual_plan_info* info, int32_t max_chunks, int32_t* chunk_first_frame, int32_t* chunk_frames, int32_t* chunk_factors, int32_t* chunk_landmarks,
                    int32_t* factor_position) {
    if (!p || !info) return VIL_ERR_INVALID_ARGUMENT;
    for (int f = 0; f < p->n_vis; ++f) {
        if (!p->vis_l || !p->vis_i || !p->vis_j || p->vis_l[f] < 0 || p->vis_l[f] >= p->L || p->vis_i[f] < 0 || p->vis_i[f] >= p->K || p->vis_j[f] < 0 || p->vis_j[f] >= p->K) return VIL_ERR_INVALID_ARGUMENT;
        if (f && p->vis_l[f] < p->vis_l[f - 1]) return VIL_ERR_INVALID_ARGUMENT;
    }
    std::vector<int> lms, fmin, fmax, anch;
    landmark_frames(p, lms, fmin, fmax, anch);
    VisPlan vp;
    if (!plan_visual(p->K, p->L, p->n_vis, lms, fmin, fmax, anch, visual_wg_budget(p->n_imu, std::max(p->n_plane, 0), std::max(p->n_edge, 0)), 0, vp)) return VIL_ERR_UNSUPPORTED;
    memset(info, 0, sizeof *info);
    info->n_chunks = vp.n_chunks; info->max_tiles = vp.tmax; info->tiles_per_wave = vd::vis_slots(vp.tmax) <= 2 ? 2 : 5; info->cost_cap = vp.cap;
    info->record_bytes = (int64_t)(8 * vp.rec_doubles); info->lds_bytes = (int64_t)(8 * (size_t)(VIS_LDS_FIXED + vp.gm));
    info->dense_record_bytes = (int64_t)vp.n_chunks * 8 * ((6 * p->K + 7) * (6 * p->K + 8) / 2 + 3 * (6 * p->K + 7) + 1);
    for (int w = 0; w < vp.n_chunks && w < max_chunks; ++w) {
        if (chunk_first_frame) chunk_first_frame[w] = vp.vwg[8 * (size_t)w + 4];
        if (chunk_frames) chunk_frames[w] = vp.vwg[8 * (size_t)w + 5];
        if (chunk_factors) chunk_factors[w] = vp.vwg[8 * (size_t)w + 3];
        if (chunk_landmarks) chunk_landmarks[w] = vp.vwg[8 * (size_t)w + 1];
    }
    if (factor_position) for (int f = 0; f < p->n_vis; ++f) factor_position[f] = vp.finv[f];
    return VIL_OK;
}


// gp / vis_f0 (sharded): the whole window's problem and the index of this rank's first visual factor in it
// Resident sources of the big tables (vil_win_solve, vil_window.hpp): the visual factor tables are expanded on the device from the landmark
// list + observation store, IMU records / sqrt-information come from the IMU slots, the prior is the device prior slot.  The vil_problem
// handed to upload_impl then carries n_vis = 0, imu_const = NULL and the prior's block tables only.
struct WinSrc {
    const int* lm_track; const int* lm_startf; const int* lm_nobs;      // host, L each
    int n_vis, T;
    const double* d_store; const int* fslot; const int* islot;          // slots of the window frames (host, K each)
    const double* d_rec; const double* d_U; const double* sum_dt;       // IMU slots (device) and the intervals' durations (host, per physical slot)
    const double* pJ0; const double* pr0; const double* px0; double* pH; double* pg0; double* pc0;     // device prior slot (prior.n > 0)
    int lb, le, f0, n_vis_all;       // sharded (gp != null): this rank keeps the factors of landmarks [lb, le), the first of which is factor f0 of the n_vis_all of the window
};

// first landmark whose factor prefix reaches r / world of the total (lms: factor prefix per landmark, L + 1 entries) -- THE sharding rule of the visual factors
static int shard_cut(const std::vector<int>& lms, int L, int r, int world) {
    if (r <= 0) return 0;
    if (r >= world) return L;
    const long long target = (long long)lms[L] * r / world;
    return std::min((int)(std::lower_bound(lms.begin(), lms.end(), (int)target) - lms.begin()), L);
}

// check_setup: wait for k_setup's verdict (an IMU covariance that is not positive definite) and return it; false: nothing is waited for --
// the first step kernel of the solve ends it with that status (DevP::setup_stat), and the launches of the solve queue up behind the upload
static int upload_impl(vil_ctx* c, const vil_problem* p, const vil_state* s, bool sharded, const vil_device_lidar* dl = nullptr, const vil_problem* gp = nullptr, int vis_f0 = 0, bool check_setup = true, const WinSrc* ws = nullptr) {
    if (!c) return VIL_ERR_INVALID_ARGUMENT;
    int st = validate(p, s, dl != nullptr, ws != nullptr);
    if (st != VIL_OK) return st;                 // an invalid problem leaves the resident one untouched
    c->uploaded = false;                         // from here on the arena is rewritten: resident only again after a complete upload
    c->resident_kind = 0; c->reset_pending = false;
    HIPCHK(hipSetDevice(c->device));
    const int K = p->K, L = p->L, D = 15 * K + 7, NV = 6 * K + 7, NS = 16 * K + 8 + L;
#ifdef VIL_TUNING
    static const bool up_trace = getenv("VIL_UPLOAD_TRACE") != nullptr;
    auto up_t0 = std::chrono::steady_clock::now();
    #define UPTICK(name) do { if (up_trace) { const auto t_ = std::chrono::steady_clock::now(); fprintf(stderr, "[upload] %-10s %7.1f us\n", name, std::chrono::duration<double, std::micro>(t_ - up_t0).count()); up_t0 = t_; } } while (0)
#else
    #define UPTICK(name) do {} while (0)
#endif
    Arena& ar = c->ar;
    if (c->up_pending) { HIPCHK(hipEventSynchronize(c->up_ev)); c->up_pending = false; }
    ar.reset();
    DevP P; memset(&P, 0, sizeof P);
    P.K = K; P.L = L; P.D = D; P.NV = NV; P.NS = NS;
    P.ex_const = p->ex_const; P.use_td = p->use_td; P.td_free = (p->use_td && !p->td_const) ? 1 : 0;
    memcpy(P.G, p->G, sizeof P.G); P.sqrt_info = p->sqrt_info_px; P.k_tr = p->tr_over_row;
    { double R[9]; quat_to_R_host(p->q_lb, R); for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P.Rbl[3 * i + j] = R[3 * j + i]; }
      for (int i = 0; i < 3; ++i) P.tbl[i] = -(P.Rbl[3 * i] * p->t_lb[0] + P.Rbl[3 * i + 1] * p->t_lb[1] + P.Rbl[3 * i + 2] * p->t_lb[2]); }
    struct Fix { size_t off; void** slot; bool scratch; };
    std::vector<Fix> fix;
    bool oom = false;
    // src != null: a table (copied into the pinned image); src == null: zero-initialised device work space
    auto put = [&](const void* src, size_t bytes, void** slot) {
        if (!src || !bytes) { const size_t o = ar.take_scratch(bytes ? bytes : 8); fix.push_back({o, slot, true}); return o; }
        const size_t o = ar.take(bytes);
        if (o == (size_t)-1) { oom = true; return (size_t)0; }
        memcpy(ar.h + o, src, bytes); fix.push_back({o, slot, false});
        return o;
    };
    // a table that is produced in place (transpositions): space in the pinned image, to be filled before the next put()
    auto reserve = [&](size_t bytes, void** slot) -> char* {
        const size_t o = ar.take(bytes ? bytes : 8);
        if (o == (size_t)-1) { oom = true; return nullptr; }
        fix.push_back({o, slot, false});
        return ar.h + o;
    };
    // constancy
    if (p->pose_const) put(p->pose_const, K, (void**)&P.pose_const);
    if (p->sb_const) put(p->sb_const, K, (void**)&P.sb_const);
    if (p->lm_const && L) put(p->lm_const, L, (void**)&P.lm_const);
    // state x[0], x[1], backup
    std::vector<double> x(NS);
    memcpy(&x[0], s->pose, sizeof(double) * 7 * K); memcpy(&x[7 * K], s->speedbias, sizeof(double) * 9 * K);
    memcpy(&x[16 * K], s->ex_pose, sizeof(double) * 7); x[16 * K + 7] = s->td[0];
    if (L) memcpy(&x[16 * K + 8], s->inv_depth, sizeof(double) * L);
    put(x.data(), sizeof(double) * NS, (void**)&P.x[0]);
    if (ws) { put(nullptr, sizeof(double) * NS, (void**)&P.x[1]); put(nullptr, sizeof(double) * NS, (void**)&c->d_x0); }      // (k_win_pack copies them on the device)
    else { put(x.data(), sizeof(double) * NS, (void**)&P.x[1]); put(x.data(), sizeof(double) * NS, (void**)&c->d_x0); }
    put(nullptr, sizeof(double) * NS, (void**)&c->d_xsave);
    UPTICK("head");
    // visual
    const int n_vis = ws ? ws->n_vis : p->n_vis;
    P.n_vis = n_vis; P.vis_stride = (n_vis + 31) & ~31;
    int* wd_track = nullptr; int* wd_startf = nullptr;       // device copies of the landmark table (resident window)
    {
        std::vector<int> lms(L + 1, 0);
        const int ws_lb = (ws && gp) ? ws->lb : 0, ws_le = (ws && gp) ? ws->le : L;      // resident window under a communicator: the factors of the owned landmarks only
        if (ws) for (int l = 0; l < L; ++l) lms[l + 1] = lms[l] + ((l >= ws_lb && l < ws_le) ? ws->lm_nobs[l] - 1 : 0);
        else {
            for (int f = 0; f < p->n_vis; ++f) lms[p->vis_l[f] + 1]++;
            for (int l = 0; l < L; ++l) lms[l + 1] += lms[l];
        }
        // frame window of every landmark, then the plan of the visual role (plan_visual above: sorted order, chunks, device tables)
        std::vector<int> fmin(std::max(L, 1), K), fmax(std::max(L, 1), -1), anch(std::max(L, 1), 0);
        if (ws) { for (int l = ws_lb; l < ws_le; ++l) { anch[l] = fmin[l] = ws->lm_startf[l]; fmax[l] = ws->lm_startf[l] + ws->lm_nobs[l] - 1; } }
        else for (int f = 0; f < p->n_vis; ++f) {
            const int l = p->vis_l[f], lo = std::min(p->vis_i[f], p->vis_j[f]), hi = std::max(p->vis_i[f], p->vis_j[f]);
            anch[l] = p->vis_i[f]; fmin[l] = std::min(fmin[l], lo); fmax[l] = std::max(fmax[l], hi);
        }
        VisPlan vp;
        {
            int npt = std::max(p->n_plane, 0), net = std::max(p->n_edge, 0);           // LiDAR points of this upload (resident slabs: what they hold)
            if (p->n_plane == VIL_LIDAR_RESIDENT) for (auto& sl : c->slabs) { npt += sl.np; net += sl.ne; }
            int vwg_max = visual_wg_budget(p->n_imu, npt, net), cap_forced = 0;
            if (const char* ev = VIL_TUNE_ENV("VIL_VWG")) vwg_max = std::max(1, atoi(ev));
            if (const char* ev = VIL_TUNE_ENV("VIL_VCAP")) cap_forced = std::max(1, atoi(ev));
            if (!plan_visual(K, L, n_vis, lms, fmin, fmax, anch, vwg_max, cap_forced, vp)) return VIL_ERR_UNSUPPORTED;
        }
        const std::vector<int>& fperm = vp.fperm; const std::vector<int>& finv = vp.finv;
        put(finv.data(), 4 * finv.size(), (void**)&P.vfinv);
        if (ws) {
            // resident window: the factor tables are device work space, k_win_pack fills them from the observation store
            put(nullptr, 8 * (size_t)14 * std::max(P.vis_stride, 1), (void**)&P.vis_c);
            put(nullptr, 4 * (size_t)std::max(n_vis, 1), (void**)&P.vis_i); put(nullptr, 4 * (size_t)std::max(n_vis, 1), (void**)&P.vis_j); put(nullptr, 4 * (size_t)std::max(n_vis, 1), (void**)&P.vis_l);
            put(ws->lm_track, 4 * (size_t)std::max(L, 1), (void**)&wd_track); put(ws->lm_startf, 4 * (size_t)std::max(L, 1), (void**)&wd_startf);
        } else {
            if (double* soa = (double*)reserve(8 * (size_t)14 * std::max(P.vis_stride, 1), (void**)&P.vis_c)) {
                for (int q = 0; q < 14; ++q) {
                    double* row = soa + (size_t)q * P.vis_stride;
                    for (int pos = 0; pos < p->n_vis; ++pos) row[pos] = p->vis_const[(size_t)fperm[pos] * 14 + q];
                    for (int f = p->n_vis; f < P.vis_stride; ++f) row[f] = 0.0;
                }
            }
            std::vector<int> si(std::max(p->n_vis, 1)), sj(std::max(p->n_vis, 1)), sl(std::max(p->n_vis, 1));
            for (int pos = 0; pos < p->n_vis; ++pos) { si[pos] = p->vis_i[fperm[pos]]; sj[pos] = p->vis_j[fperm[pos]]; sl[pos] = p->vis_l[fperm[pos]]; }
            put(si.data(), 4 * (size_t)p->n_vis, (void**)&P.vis_i); put(sj.data(), 4 * (size_t)p->n_vis, (void**)&P.vis_j); put(sl.data(), 4 * (size_t)p->n_vis, (void**)&P.vis_l);
        }
        put(lms.data(), 4 * (size_t)(L + 1), (void**)&P.lm_start);
        {
            std::vector<int> acol(std::max(L, 1), -1);
            if (ws) {
                for (int l = 0; l < L; ++l) acol[l] = 6 * ws->lm_startf[l];
                put(acol.data(), 4 * acol.size(), (void**)&P.lm_acol); put(nullptr, 4 * (size_t)std::max(n_vis, 1), (void**)&P.fcol);
            } else {
                std::vector<int> fcol(std::max(p->n_vis, 1), 0);
                for (int f = p->n_vis - 1; f >= 0; --f) { acol[p->vis_l[f]] = 6 * p->vis_i[f]; fcol[f] = 6 * p->vis_j[f]; }
                put(acol.data(), 4 * acol.size(), (void**)&P.lm_acol); put(fcol.data(), 4 * fcol.size(), (void**)&P.fcol);
            }
        }
        P.vis_f0 = 0;
        if (gp) {                                        // tables of the whole window for the step kernel (every rank walks every landmark)
            const int gnv = ws ? ws->n_vis_all : gp->n_vis;
            std::vector<int> gl(L + 1, 0), gac(std::max(L, 1), -1), gfc(std::max(gnv, 1), 0);
            if (ws) {
                for (int l = 0; l < L; ++l) {
                    gl[l + 1] = gl[l] + ws->lm_nobs[l] - 1; gac[l] = 6 * ws->lm_startf[l];
                    for (int q = 1; q < ws->lm_nobs[l]; ++q) gfc[gl[l] + q - 1] = 6 * (ws->lm_startf[l] + q);
                }
            } else {
                for (int f = 0; f < gp->n_vis; ++f) gl[gp->vis_l[f] + 1]++;
                for (int l = 0; l < L; ++l) gl[l + 1] += gl[l];
                for (int f = gp->n_vis - 1; f >= 0; --f) { gac[gp->vis_l[f]] = 6 * gp->vis_i[f]; gfc[f] = 6 * gp->vis_j[f]; }
            }
            put(gl.data(), 4 * gl.size(), (void**)&P.glm_start); put(gac.data(), 4 * gac.size(), (void**)&P.glm_acol); put(gfc.data(), 4 * gfc.size(), (void**)&P.gfcol);
            P.vis_f0 = vis_f0;
        }
        P.n_vwg = vp.n_chunks;
        P.vis_ts = vd::vis_slots(vp.tmax) <= 2 ? 2 : 5;
        c->vis_gm = vp.gm;
        put(vp.wend.data(), 4 * vp.wend.size(), (void**)&P.vwend);
        put(vp.vwg.data(), 4 * vp.vwg.size(), (void**)&P.vwg); put(vp.vrec.data(), 4 * vp.vrec.size(), (void**)&P.vrec);
        put(vp.vlm.data(), 4 * vp.vlm.size(), (void**)&P.vlm); put(vp.vfac.data(), 4 * vp.vfac.size(), (void**)&P.vfac);
#ifdef VIL_TUNING
        put(nullptr, 8 * 2 * std::max(vp.rec_doubles, (size_t)16), (void**)&P.vpart);      // (x 2: the second half is the mirror of the record-traffic experiment, VIL_SKIP=1024 -- tuning build only)
        P.vmirror = (long long)std::max(vp.rec_doubles, (size_t)16);
#else
        put(nullptr, 8 * std::max(vp.rec_doubles, (size_t)16), (void**)&P.vpart);
#endif
    }
    UPTICK("visual");
    // LiDAR
    {
        std::vector<int> ch;
        std::vector<int> lcp(2 * (K + 1), 0);   // chunk ranges per pose (chunks are pose-ordered): plane [0..K], edge [K+1..2K+1]
        auto ranges = [&](const std::vector<int>& chunks, int base) {
            std::vector<int> cnt(K + 1, 0);
            for (size_t q = 0; q < chunks.size() / 3; ++q) cnt[chunks[3 * q + 2] + 1]++;
            for (int k = 0; k < K; ++k) cnt[k + 1] += cnt[k];
            for (int k = 0; k <= K; ++k) lcp[base + k] = cnt[k];
        };
        // device-resident tables (vil_internal.h): every factor sits on pose 0 in the caller's order -- only the chunk list is built here
        auto chunks_pose0 = [&](int n) { ch.clear(); for (int s0 = 0; s0 < n; s0 += VIL_THREADS) { ch.push_back(s0); ch.push_back(std::min(VIL_THREADS, n - s0)); ch.push_back(0); } };
        std::vector<int> cnt;
        const bool res_lidar = p->n_plane == VIL_LIDAR_RESIDENT && p->n_edge == VIL_LIDAR_RESIDENT;
        c->lidar_resident = res_lidar;
        // resident frame slabs (vil_lidar_push): slab i belongs to pose K - count + i; only the chunk list is built here
        auto chunks_slabs = [&](bool plane) {
            ch.clear();
            const int ns = (int)c->slabs.size();
            for (int i = 0; i < ns; ++i) {
                const int n = plane ? c->slabs[i].np : c->slabs[i].ne, cap = plane ? c->cap_p : c->cap_e, pose = K - ns + i;
                for (int s0 = 0; s0 < n; s0 += VIL_THREADS) { ch.push_back(c->slabs[i].slot * cap + s0); ch.push_back(std::min(VIL_THREADS, n - s0)); ch.push_back(pose); }
            }
        };
        int np_tot = p->n_plane, ne_tot = p->n_edge;
        if (res_lidar) {
            if ((int)c->slabs.size() > K) return VIL_ERR_UNSUPPORTED;
            np_tot = ne_tot = 0;
            for (auto& sl : c->slabs) { np_tot += sl.np; ne_tot += sl.ne; }
        }
        if (res_lidar) { chunks_slabs(true); c->plane_perm.clear(); put(nullptr, 8, (void**)&P.pl_c); }
        else if (dl) { chunks_pose0(p->n_plane); c->plane_perm.clear(); put(nullptr, 8, (void**)&P.pl_c); }
        else {
            lidar_order(p->n_plane, p->plane_pose, K, c->plane_perm, cnt); lidar_chunks(cnt, K, ch);
            P.pl_stride = (p->n_plane + 31) & ~31;
            if (double* dst = (double*)reserve(8 * (size_t)7 * std::max(P.pl_stride, 1), (void**)&P.pl_c)) lidar_soa(p->n_plane, 7, p->plane_const, c->plane_perm, P.pl_stride, dst);
        }
        P.n_plane = np_tot; P.n_pchunk = (int)ch.size() / 3; ranges(ch, 0);
        c->mm.lidar0 = false;
        for (size_t q = 0; q < ch.size() / 3; ++q) if (ch[3 * q + 2] == 0) c->mm.lidar0 = true;
        put(ch.data(), 4 * ch.size(), (void**)&P.pchunk);
        if (res_lidar) { chunks_slabs(false); c->edge_perm.clear(); put(nullptr, 8, (void**)&P.ed_c); }
        else if (dl) { chunks_pose0(p->n_edge); c->edge_perm.clear(); put(nullptr, 8, (void**)&P.ed_c); }
        else {
            lidar_order(p->n_edge, p->edge_pose, K, c->edge_perm, cnt); lidar_chunks(cnt, K, ch);
            P.ed_stride = (p->n_edge + 31) & ~31;
            if (double* dst = (double*)reserve(8 * (size_t)9 * std::max(P.ed_stride, 1), (void**)&P.ed_c)) lidar_soa(p->n_edge, 9, p->edge_const, c->edge_perm, P.ed_stride, dst);
        }
        P.n_edge = ne_tot; P.n_echunk = (int)ch.size() / 3; ranges(ch, K + 1);
        for (size_t q = 0; q < ch.size() / 3; ++q) if (ch[3 * q + 2] == 0) c->mm.lidar0 = true;
        put(ch.data(), 4 * ch.size(), (void**)&P.echunk);
        // (resident slabs under a communicator: a frame with fewer points than ranks leaves some ranks' slices empty -- whether pose 0 carries LiDAR factors, and with it the
        //  kept / dropped layout of the marginalisation every rank commits, is decided from the UNSLICED counts)
        if (res_lidar) c->mm.lidar0 = (int)c->slabs.size() == K && c->slabs[0].np_all + c->slabs[0].ne_all > 0;
        put(nullptr, 8 * (size_t)28 * std::max(P.n_pchunk + P.n_echunk, 1), (void**)&P.lpart);
        put(lcp.data(), 4 * lcp.size(), (void**)&P.lchunk_pose);
    }
    UPTICK("lidar");
    // IMU
    P.n_imu = p->n_imu;
    if (ws) put(nullptr, 8 * (size_t)287 * std::max(p->n_imu, 1), (void**)&P.imu_c);      // gathered from the IMU slots by k_win_pack, like U
    else put(p->imu_const, 8 * (size_t)287 * p->n_imu, (void**)&P.imu_c);
    put(nullptr, 8 * (size_t)225 * std::max(p->n_imu, 1), (void**)&P.imu_U);
    put(p->imu_i, 4 * (size_t)p->n_imu, (void**)&P.imu_i); put(p->imu_j, 4 * (size_t)p->n_imu, (void**)&P.imu_j);
    put(nullptr, 8 * (size_t)931 * std::max(p->n_imu, 1), (void**)&P.ipart);
    if (!c->d_imu_perm) {   // the order of an IMU role's record entries in a one-launch iteration: what the chain workgroup gathers first.  Constant: one device copy per context
        int perm[1024]; int n = VIL_CHAIN_REC;
        for (int e = 0; e < 931; ++e) { const int ce = chain_rec_index(e); if (ce >= 0) perm[ce] = e; else perm[n++] = e; }
        for (int e = 931; e < 1024; ++e) perm[e] = 930;
        HIPCHK(hipMalloc((void**)&c->d_imu_perm, sizeof(perm)));
        HIPCHK(hipMemcpy(c->d_imu_perm, perm, sizeof(perm), hipMemcpyHostToDevice));
    }
    P.imu_perm = c->d_imu_perm;
    put(nullptr, 4 * (size_t)(std::max(p->n_imu, 1) + 8), (void**)&P.cflag);
    put(nullptr, 8 * (size_t)VIL_CHAIN_REC * std::max(p->n_imu, 1), (void**)&P.irec);      // (compact IMU records for the chain workgroup of a one-launch iteration)
    // prior
    P.pn = p->prior.n > 0 ? p->prior.n : 0; P.pnblk = P.pn ? p->prior.nblk : 0;
    c->prior_joff.clear();
    std::vector<int> pinv(D, -1);
    if (P.pn) {
        const vil_prior& pr = p->prior;
        const int n = pr.n;
        std::vector<int> xoff(pr.nblk), pmap(n, -1);
        int xo = 0, jo = 0;
        for (int b = 0; b < pr.nblk; ++b) {
            const int kind = pr.blk_kind[b], gs = (kind == VIL_BLK_POSE || kind == VIL_BLK_EX) ? 7 : (kind == VIL_BLK_SPEEDBIAS ? 9 : 1), ls = gs == 7 ? 6 : gs;
            xoff[b] = xo; xo += gs; c->prior_joff.push_back(jo); jo += n * gs;
            int col = -1;
            const int idx = pr.blk_index[b];
            if (kind == VIL_BLK_POSE) { if (idx < 0 || idx >= K) return VIL_ERR_INVALID_ARGUMENT; col = (p->pose_const && p->pose_const[idx]) ? -1 : 6 * idx; }
            else if (kind == VIL_BLK_SPEEDBIAS) { if (idx < 0 || idx >= K) return VIL_ERR_INVALID_ARGUMENT; col = (p->sb_const && p->sb_const[idx]) ? -1 : 6 * K + 7 + 9 * idx; }
            else if (kind == VIL_BLK_EX) col = p->ex_const ? -1 : 6 * K;
            else col = P.td_free ? 6 * K + 6 : -1;
            if (pr.blk_col[b] < 0 || pr.blk_col[b] + ls > n) return VIL_ERR_INVALID_ARGUMENT;
            for (int q = 0; q < ls; ++q) pmap[pr.blk_col[b] + q] = col < 0 ? -1 : col + q;
        }
        put(pr.blk_kind, 4 * (size_t)pr.nblk, (void**)&P.pblk_kind); put(pr.blk_index, 4 * (size_t)pr.nblk, (void**)&P.pblk_index);
        put(pr.blk_col, 4 * (size_t)pr.nblk, (void**)&P.pblk_col); put(xoff.data(), 4 * (size_t)pr.nblk, (void**)&P.pblk_xoff);
        put(pmap.data(), 4 * (size_t)n, (void**)&P.pmap);
        for (int q = 0; q < n; ++q) if (pmap[q] >= 0) pinv[pmap[q]] = q;
        if (!ws) {
            put(pr.x0, 8 * (size_t)xo, (void**)&P.px0); put(pr.J0, 8 * (size_t)n * n, (void**)&P.pJ0); put(pr.r0, 8 * (size_t)n, (void**)&P.pr0);
            put(nullptr, 8 * (size_t)n * n, (void**)&P.pH); put(nullptr, 8 * (size_t)n, (void**)&P.pg0); put(nullptr, 8, (void**)&P.pc0);
        }
    }
    put(pinv.data(), 4 * (size_t)D, (void**)&P.pinv);
    put(nullptr, 8 * (size_t)((P.pn ? P.pn + 1 : 0) + 601 * (p->n_icp + p->n_lps) + 1), (void**)&P.mpart);
    // ICP / LPS
    P.n_icp = p->n_icp; P.n_lps = p->n_lps;
    put(p->icp_ids, 16 * (size_t)p->n_icp, (void**)&P.icp_ids); put(p->icp_const, 80 * (size_t)p->n_icp, (void**)&P.icp_c);
    put(p->lps_ids, 8 * (size_t)p->n_lps, (void**)&P.lps_ids); put(p->lps_const, 56 * (size_t)p->n_lps, (void**)&P.lps_c);
    UPTICK("imu+prior");
    // systems + work space (zero-initialised)
    // one contiguous block per set: [S | gred | bc | diag | cost | 2 spare | hll | bl | invp | sl | eA | eO]
    const size_t Lp = (size_t)std::max(L, 1), Fp = (size_t)std::max(gp ? (ws ? ws->n_vis_all : gp->n_vis) : n_vis, 1);
    const size_t ar_cam = ((size_t)D * D + 3 * (size_t)D + 3 + 1) & ~size_t(1);
    c->span = ar_cam + 4 * Lp + 13 * Lp + 6 * Fp;
    for (int q = 0; q < 2; ++q) put(nullptr, 8 * c->span, (void**)&P.sys[q].ar);
    P.rank = sharded ? c->rank : 0; P.world = sharded ? c->world : 1;
    for (auto& g : c->graphs) hipGraphExecDestroy(g.exec);
    c->graphs.clear(); c->solves_since_upload = 0;
    c->sharded = sharded && c->world > 1;
    // multi-GPU plumbing (set 0 = this rank's partial system, all-reduced into set 1, which the step kernel reads); vil_debug_set_split
    // runs it on a single rank (tests)
    c->split = c->sharded || c->force_split;
    put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.Sl); put(nullptr, 8 * (size_t)D, (void**)&P.Sc); put(nullptr, 8 * (size_t)D, (void**)&P.dc); put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.dl);
    put(nullptr, 8 * (size_t)D, (void**)&P.gradc); put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.gradl); put(nullptr, 8 * (size_t)D, (void**)&P.gnc); put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.gnl);
    { const size_t Tm = (size_t)(D + 16) / 16; put(nullptr, 8 * std::max((size_t)D * D, (size_t)TILE_SZ * (Tm * (Tm + 1) / 2)), (void**)&P.M); } put(nullptr, 16 * (size_t)D, (void**)&P.stepc);      // (two 64-bit words per value: half + launch epoch)
    put(nullptr, 8 * 4 * 16, (void**)&P.hpart); put(nullptr, 4 * 16, (void**)&P.hflag);
    put(nullptr, 8 * 8 * 16 * 8, (void**)&P.hpart2); put(nullptr, 4 * 16 * 8, (void**)&P.hflag2);      // one slot per helper WAVE
    put(nullptr, 16, (void**)&P.xflag); put(nullptr, 16, (void**)&P.xstat);
    put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.la); put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.lb); put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.stepl);
    put(nullptr, 8 * (size_t)D, (void**)&P.tmpc); put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.tmpl);
    put(nullptr, sizeof(Ctl), (void**)&P.ctl);
    put(nullptr, 8 * 64, (void**)&P.dbg);
    // ---- chain eliminated ahead of the step kernel (vil_prechain.hpp): every IMU factor joins frames (k, k+1), at most one per pair, single GPU.
    //      The gather table is built here once per upload: the chain workgroup then sums <= 3 sources per entry in a fixed order.
    bool pre_ok = !sharded && !c->force_split && c->launch_mode != 2 && L >= 0 && K >= 3 && VIL_TUNE_ENV("VIL_NO_PRECHAIN") == nullptr;
    std::vector<int> as_i(K, -1), as_j(K, -1);
    for (int f = 0; f < p->n_imu && pre_ok; ++f) {
        const int i = p->imu_i[f], j = p->imu_j[f];
        if (j != i + 1 || as_i[i] >= 0 || as_j[j] >= 0) pre_ok = false; else { as_i[i] = f; as_j[j] = f; }
    }
    {
        const int rs = vd::chain_rs(K);
        put(nullptr, 8 * (size_t)vd::chain_wcols(K) * rs, (void**)&P.chW);
        put(nullptr, 8 * (size_t)54 * K, (void**)&P.chLdg); put(nullptr, 8 * (size_t)82 * K, (void**)&P.chLsb); put(nullptr, 8 * (size_t)136 * K, (void**)&P.chLraw);
        put(nullptr, 8 * (size_t)9 * K, (void**)&P.chSc); put(nullptr, 8 * (size_t)9 * K, (void**)&P.chDc);
        put(nullptr, 8 * (size_t)2 * (NV + 1), (void**)&P.chZ); put(nullptr, 8 * 4, (void**)&P.chQ); put(nullptr, 16, (void**)&P.chOk);
        put(nullptr, 4 * (size_t)(gather_blocks(D, NV, RED_EPW, false) + 8), (void**)&P.gflag);      // (one flag per gather workgroup of the merged launch: never more than the gather kernel has)
        put(nullptr, 64, (void**)&P.chflag); put(nullptr, 4 * 64, (void**)&P.wwflag); put(nullptr, 4 * (size_t)(K + 8), (void**)&P.swflag);
        put(nullptr, 4 * 64, (void**)&P.sall);
        put(nullptr, 4 * (size_t)VIL_SFLAG_MAX, (void**)&P.sflag);      // one flag per sweep workgroup of a one-launch iteration (taken only when there are fewer: below)
        put(nullptr, 256, (void**)&P.ihdr); put(nullptr, 8 * 2 * (size_t)(16 * K + 8), (void**)&P.xtag);
        put(nullptr, 64, (void**)&P.abortf);      // (raised by a wait on another workgroup's flag that gives up: vil_math.hpp, spin_until_eq)
        { const size_t Tp = (size_t)(NV + 1 + 15) / 16; put(nullptr, 8 * (size_t)TILE_SZ * (Tp * (Tp + 1) / 2), (void**)&P.chWW); }
        put(nullptr, 8 * (size_t)VIL_CHC_MAX, (void**)&P.chc);
    }
    UPTICK("ws-puts");
    if (pre_ok) {
        // the table depends on K, the IMU factor layout and the prior's block structure only: consecutive windows of a tracker share it, so it
        // lives in its own device buffer and is rebuilt (and sent) only when that key changes
        std::vector<int> key; key.reserve(2 * K + D + 2);
        key.push_back(K); key.push_back(P.pn); key.insert(key.end(), as_i.begin(), as_i.end()); key.insert(key.end(), as_j.begin(), as_j.end()); key.insert(key.end(), pinv.begin(), pinv.end());
        int cs = -1;
        for (int q = 0; q < 4; ++q) if (c->chtabs[q].d && c->chtabs[q].key == key) cs = q;
        const bool ct_hit = cs >= 0;
        if (!ct_hit) { cs = 0; for (int q = 1; q < 4; ++q) if (c->chtabs[q].used < c->chtabs[cs].used) cs = q; }      // (least recently used)
        vil_ctx::ChTab& ct = c->chtabs[cs];
        ct.used = ++c->chtab_clock; c->chtab_cur = cs;
        if (!ct_hit) {
            const int NPs = vd::chain_slab_nps(K), pn = P.pn;
            const int o_dg = 0, o_sub = vd::even_up(45 * K), o_pbc = o_sub + vd::even_up(81 * K), o_pp = o_pbc + 162 * K, o_rhs = o_pp + CHAIN_NPC_MAX * NPs;
            std::vector<int> tab, pq(9 * K, -1);
            int npc = 0;
            for (int jc = 0; jc < 9 * K; ++jc) if (pn > 0 && pinv[NV + jc] >= 0) pq[jc] = npc++;
            if (npc > CHAIN_NPC_MAX) pre_ok = false;       // (cannot happen: the prior's speed-bias blocks are neighbours -- checked for the chain path)
            auto ip = [&](int f, int la, int lb) { return f < 0 ? -1 : f * 931 + la * 30 + lb; };
            auto pr = [&](int r, int col) { if (pn <= 0) return -1; const int pi = pinv[r], pj = pinv[col]; return (pi >= 0 && pj >= 0) ? pi * pn + pj : -1; };
            // (IMU sources carry two indices: (index into the 931-double records + 1) in the low half, (index into the compact records of a one-launch iteration + 1) in the high half; 0: none)
            auto both = [&](int a) { if (a < 0) return 0; const int f = a / 931, ce = chain_rec_index(a % 931); return (a + 1) | ((f * VIL_CHAIN_REC + ce + 1) << 16); };
            auto emit = [&](int dst, int a, int b, int cc) { if (a < 0 && b < 0 && cc == -1) return; tab.push_back(dst); tab.push_back(both(a)); tab.push_back(both(b)); tab.push_back(cc); };
            for (int k = 0; k < K && pre_ok; ++k) {
                const int fi = as_i[k], fj = as_j[k];
                for (int i = 0; i < 9; ++i) for (int j = 0; j <= i; ++j) emit(o_dg + 45 * k + i * (i + 1) / 2 + j, ip(fi, 6 + i, 6 + j), ip(fj, 21 + i, 21 + j), pr(NV + 9 * k + i, NV + 9 * k + j));
                if (k + 1 < K) for (int q = 0; q < 9; ++q) for (int cc = 0; cc < 9; ++cc) emit(o_sub + 81 * k + q * 9 + cc, ip(fi, 21 + q, 6 + cc), -1, pr(NV + 9 * (k + 1) + q, NV + 9 * k + cc));
                for (int cc = 0; cc < 9; ++cc) {
                    const int pj = pn > 0 ? pinv[NV + 9 * k + cc] : -1;
                    emit(o_rhs + 9 * k + cc, fi < 0 ? -1 : fi * 931 + 900 + 6 + cc, fj < 0 ? -1 : fj * 931 + 900 + 21 + cc, pj >= 0 ? -pj - 2 : -1);
                    for (int d = 0; d < 3; ++d) {              // pose rows of frames k-1, k, k+1: the IMU factors' share, compact
                        const int fr = k - 1 + d;
                        if (fr < 0 || fr >= K) continue;
                        for (int lr = 0; lr < 6; ++lr) {
                            const int a = (fi >= 0 && (fr == k || fr == k + 1)) ? ip(fi, fr == k ? lr : 15 + lr, 6 + cc) : -1;
                            const int b = (fj >= 0 && (fr == k - 1 || fr == k)) ? ip(fj, fr == k - 1 ? lr : 15 + lr, 21 + cc) : -1;
                            emit(o_pbc + ((k * 3 + d) * 6 + lr) * 9 + cc, a, b, -1);
                        }
                    }
                    if (pq[9 * k + cc] >= 0) for (int r = 0; r < NV; ++r) emit(o_pp + pq[9 * k + cc] * NPs + r, -1, -1, pr(r, NV + 9 * k + cc));      // the prior's share: every row
                }
            }
            {   // the chain workgroup deals consecutive entries to consecutive threads and every entry has its own target: sorted by source address, a wave's loads of
                // the IMU records (and of the prior) fall into a few cache lines each instead of one line per lane -- the gather is bound by the lines its loads touch
                const size_t ne = tab.size() / 4;
                std::vector<std::array<int, 4>> ent(ne);
                memcpy(ent.data(), tab.data(), 16 * ne);
                auto skey = [](const std::array<int, 4>& e) -> long long { return e[1] > 0 ? (e[1] >> 16) : e[2] > 0 ? (e[2] >> 16) : (1LL << 40) + (e[3] >= 0 ? (1LL << 32) + e[3] : -e[3]); };
                std::stable_sort(ent.begin(), ent.end(), [&](const std::array<int, 4>& x, const std::array<int, 4>& y) { return skey(x) < skey(y); });
                memcpy(tab.data(), ent.data(), 16 * ne);
            }
            tab.insert(tab.end(), pq.begin(), pq.end());       // behind the table: the chain column -> prior column map
            while (tab.size() & 3) tab.push_back(0);
            const size_t bytes = 4 * tab.size();
            if (ct.ev_pending) { HIPCHK(hipEventSynchronize(ct.ev)); ct.ev_pending = false; }      // the previous table's DMA has left the pinned copy
            if (bytes > ct.cap) {
                HIPCHK(hipStreamSynchronize(c->stream));      // (nobody reads the old one)
                if (ct.d) hipFree(ct.d); if (ct.h) hipHostFree(ct.h);
                ct.d = nullptr; ct.h = nullptr; ct.cap = 0;
                HIPCHK(hipMalloc((void**)&ct.d, 2 * bytes)); HIPCHK(hipHostMalloc((void**)&ct.h, 2 * bytes, hipHostMallocDefault));
                ct.cap = 2 * bytes;
            }
            memcpy(ct.h, tab.data(), bytes);
            HIPCHK(hipMemcpyAsync(ct.d, ct.h, bytes, hipMemcpyHostToDevice, c->stream));
            if (!ct.ev) HIPCHK(hipEventCreateWithFlags(&ct.ev, hipEventDisableTiming));
            HIPCHK(hipEventRecord(ct.ev, c->stream)); ct.ev_pending = true;
            ct.n = ((int)tab.size() - ((9 * K + 3) & ~3)) / 4; ct.key.swap(key);
            UPTICK("chtab-new");
        }
        P.n_chtab = ct.n;
        if (ct.n > VIL_CHC_MAX) pre_ok = false;
    }
    if (const char* ev = VIL_TUNE_ENV("VIL_SKIP")) P.skip_mask = atoi(ev);
    // helper workgroups of the step kernel: worth it once every master thread would own more than one landmark
    // (a helper keeps its landmarks' rows in registers between its two passes while it has at least one thread per landmark -- vil_step.hpp, lm_rows_quad:
    //  up to 1920 landmarks a QUAD of threads per landmark, 128 landmarks per helper; beyond that a PAIR, 256 per helper; at most 15 helpers, so past
    //  3840 landmarks a helper's threads loop over its slice)
    P.n_help = vil_helpers_for(L);
    if (const char* ev = VIL_TUNE_ENV("VIL_HELP")) P.n_help = std::max(0, std::min(15, atoi(ev)));
    // master and helpers wait for one another inside the launch: all of them must be resident at once (vil_coop.hpp).  With its
    // dynamic LDS a step workgroup owns a compute unit; a device with fewer units than 1 + n_help runs without helpers.
    {
        int cus = 0;
        cus = vilcoop::compute_units(c->device);      // (what this process really has: a CU mask is not in the device attribute)
        if (1 + P.n_help > cus / 2) P.n_help = 0;
    }
    UPTICK("workspace");
    // device allocation + single H2D copy
    if (oom) return VIL_ERR_DEVICE;
    const size_t tables = (ar.hsize + 255) & ~size_t(255), total = tables + ((ar.ssize + 255) & ~size_t(255));
    if (total > ar.cap) { if (ar.d) HIPCHK(hipFree(ar.d)); ar.d = nullptr; ar.cap = 0; HIPCHK(hipMalloc(&ar.d, total + total / 4)); ar.cap = total + total / 4; }
    for (const Fix& f : fix) *f.slot = ar.d + (f.scratch ? tables : 0) + f.off;
    if (dl) { P.pl_c = dl->plane_soa; P.pl_stride = dl->plane_stride; P.ed_c = dl->edge_soa; P.ed_stride = dl->edge_stride; }
    if (!gp) { P.glm_start = P.lm_start; P.glm_acol = P.lm_acol; P.gfcol = P.fcol; }
    if (pre_ok) { const vil_ctx::ChTab& ct = c->chtabs[c->chtab_cur]; P.chtab = ct.d; P.chpq = ct.d + 4 * (size_t)ct.n; }
    if (c->lidar_resident) { P.pl_c = c->d_pl; P.pl_stride = c->nslot * c->cap_p; P.ed_c = c->d_ed; P.ed_stride = c->nslot * c->cap_e; }
    if (ws && P.pn) { P.px0 = ws->px0; P.pJ0 = ws->pJ0; P.pr0 = ws->pr0; P.pH = ws->pH; P.pg0 = ws->pg0; P.pc0 = ws->pc0; }      // the device prior slot, contractions included
    {   // what vil_marginalize_resident will need (a few passes over int tables)
        vil_ctx::MargMeta& mm = c->mm;
        const vil_problem* const lp = p;                  // this rank's shard (LiDAR points of pose 0 of a resident window: from the unsliced slab counts, above)
        if (gp) p = gp;                                    // which blocks the collected factors touch is a property of the WHOLE window, the same on every rank
        mm.has_prior = p->prior.n > 0; mm.prior_kind.clear(); mm.prior_index.clear();
        if (mm.has_prior) { mm.prior_kind.assign(p->prior.blk_kind, p->prior.blk_kind + p->prior.nblk); mm.prior_index.assign(p->prior.blk_index, p->prior.blk_index + p->prior.nblk); }
        mm.obs0.assign(K, 0); mm.n_lm0 = 0; mm.imu01 = false; mm.use_td = p->use_td != 0;
        if (ws) {
            for (int l = 0; l < L; ++l) if (ws->lm_startf[l] == 0) { ++mm.n_lm0; for (int q = 1; q < ws->lm_nobs[l]; ++q) mm.obs0[q] = 1; }
            for (int f = 0; f < p->n_imu; ++f) if (p->imu_i[f] == 0 && p->imu_j[f] == 1 && ws->sum_dt[ws->islot[1]] < 10.0) mm.imu01 = true;
        } else {
            int last_l = -1;
            for (int f = 0; f < p->n_vis; ++f) if (p->vis_i[f] == 0) { mm.obs0[p->vis_j[f]] = 1; if (p->vis_l[f] != last_l) { ++mm.n_lm0; last_l = p->vis_l[f]; } }
            for (int f = 0; f < p->n_imu; ++f) if (p->imu_i[f] == 0 && p->imu_j[f] == 1 && p->imu_const[(size_t)f * 287 + 16] < 10.0) mm.imu01 = true;
        }
        mm.icp_ids.assign(p->icp_ids, p->icp_ids + 4 * (size_t)p->n_icp); mm.lps_ids.assign(p->lps_ids, p->lps_ids + 2 * (size_t)p->n_lps);
        if (!c->lidar_resident && !dl) {
            for (int f = 0; f < p->n_plane && !mm.lidar0; ++f) if (p->plane_pose[f] == 0) mm.lidar0 = true;
            for (int f = 0; f < p->n_edge && !mm.lidar0; ++f) if (p->edge_pose[f] == 0) mm.lidar0 = true;
        }
        p = lp;
    }
    for (int q = 0; q < 2; ++q) {
        SysBuf& sb = P.sys[q];
        sb.S = sb.ar; sb.gred = sb.S + (size_t)D * D; sb.bc = sb.gred + D; sb.diag = sb.bc + D; sb.cost = sb.diag + D;
        sb.hll = sb.ar + ar_cam; sb.bl = sb.hll + Lp; sb.invp = sb.bl + Lp; sb.sl = sb.invp + Lp; sb.eA = sb.sl + Lp; sb.eO = sb.eA + 13 * Lp;
    }
    if (ar.hsize) {
        HIPCHK(hipMemcpyAsync(ar.d, ar.h, ar.hsize, hipMemcpyHostToDevice, c->stream));
        if (!c->up_ev) HIPCHK(hipEventCreateWithFlags(&c->up_ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->up_ev, c->stream)); c->up_pending = true;
    }
    if (ar.ssize) HIPCHK(hipMemsetAsync(ar.d + tables, 0, ar.ssize, c->stream));
    P.drop_role = -1; P.drop_launch = -1;
    c->P = P; c->K = K; c->L = L; c->D = D; c->NS = NS;
    c->mirror_state = false;
    if (!c->no_poll) { const int ms = ensure_mirror(c, (size_t)NS); if (ms != VIL_OK) return ms; }
    c->P.hctl = c->d_hctl; c->P.hseq = c->d_hseq; c->P.hstate = c->d_hstate;
    c->P.xorig = c->d_x0; c->P.gauge_on = c->gauge_on ? 1 : 0; c->P.setup_stat = c->d_status;
    {
        const int per = VIL_SWEEP_THREADS / 256;
        auto nsw = [&](int R) { return P.n_imu + P.n_vwg + (P.n_pchunk + per * R - 1) / (per * R) + (P.n_echunk + per * R - 1) / (per * R) + 2; };
        // (LiDAR workgroups can make several passes of two chunks -- fewer, longer workgroups when a window has more sweep roles than the device has compute units.
        //  Measured at configs[2], 496 sweep roles: 1 pass 1068 us per 13-iteration solve, 2 passes 1085, 4 passes 1120, 8 passes 1282: the dispatcher's second round
        //  balances better than any static packing.  One pass; the knob stays in the tuning build.)
        int R = 1;
        if (const char* ev = VIL_TUNE_ENV("VIL_LIDAR_REP")) R = std::max(1, atoi(ev));
        P.lidar_rep = R; c->P.lidar_rep = R;
        c->n_blocks_sweep = nsw(R);
    }
    // visual workgroups: the staged factors, the landmark records, the dense operand rows of the largest chunk
    c->lds_sweep = sizeof(double) * (size_t)(VIS_LDS_FIXED + c->vis_gm);
    if (c->lds_sweep < 8 * 2048) c->lds_sweep = 8 * 2048;
    if (c->lds_sweep > 160 * 1024) return VIL_ERR_UNSUPPORTED;
    auto grant_sweep = [&]() -> int {
        const int v = P.vis_ts == 2 ? 0 : 1;
        if ((int)c->lds_sweep > c->attr_sweep[v]) { HIPCHK(hipFuncSetAttribute(v ? (const void*)k_sweep<5> : (const void*)k_sweep<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_sweep)); c->attr_sweep[v] = (int)c->lds_sweep; }
        return VIL_OK;
    };
    { const int gs = grant_sweep(); if (gs != VIL_OK) return gs; }
    c->n_blocks_reduce = gather_blocks(D, NV, RED_EPW, false);
    // ---- step kernel variant: the speed-bias part of the reduced matrix is a chain whenever every IMU factor couples (k, k+1)
    //      and the prior's speed-bias blocks are neighbours (VINS: exactly one) -> vil_chain.hpp; anything else: dense path
    {
        bool chain = VIL_TUNE_ENV("VIL_DENSE_STEP") == nullptr;
        for (int f = 0; f < p->n_imu && chain; ++f) if (std::abs(p->imu_i[f] - p->imu_j[f]) > 1) chain = false;
        if (P.pn) {
            std::vector<int> psb;
            for (int b = 0; b < p->prior.nblk; ++b) if (p->prior.blk_kind[b] == VIL_BLK_SPEEDBIAS) psb.push_back(p->prior.blk_index[b]);
            for (int a : psb) for (int b : psb) if (std::abs(a - b) > 1) chain = false;
        }
        P.chain = 0; P.chain_rs = vd::chain_rs(K);
        if (chain) {
            const size_t Tp = (size_t)(NV + 1 + 15) / 16, tiles = (size_t)TILE_SZ * (Tp * (Tp + 1) / 2), wt = (size_t)vd::chain_wcols(K) * P.chain_rs, scr = vd::chain_scratch_doubles(K);
            const size_t fixed = step_static_lds((const void*)k_step<true, 1>) + 256;      // (the kernel's whole static LDS: StepShared + the gather / tile roles' scratch)
            if (8 * (tiles + wt + scr) + fixed <= 160 * 1024) { P.chain = 1; c->lds_step = 8 * (tiles + wt + scr); }
            else if (8 * (tiles + scr) + fixed <= 160 * 1024) { P.chain = 2; c->lds_step = 8 * (tiles + scr); }
        }
        // one GPU, chain windows: gather + step in ONE launch with the chain eliminated beside the gather (prechain 1, vil_prechain.hpp).
        // Until round 4 only up to K = 12: with ~100 kB of dynamic LDS per workgroup the ~1500 gather workgroups of a K = 20 window needed six rounds
        // on 256 compute units.  Windows the launch cannot hold (capacity check below) keep the separate gather and eliminate the chain inside
        // k_sweep, behind the IMU / prior workgroups' flags, with the W W^T tiles on extra workgroups of k_reduce (prechain 2).
        const size_t Tp_ = (size_t)(NV + 1 + 15) / 16, tiles_ = (size_t)TILE_SZ * (Tp_ * (Tp_ + 1) / 2);
        const size_t lds3 = 8 * (tiles_ + 54 * (size_t)K + 82 * (size_t)K + vd::even_up(9 * K) + 16), ldsc = 8 * vd::prechain_lds_doubles(K);
        const bool can_pre = !c->split && pre_ok && P.chain != 0 && Tp_ * (Tp_ + 1) / 2 <= 64 && lds3 + step_static_lds((const void*)k_step<true, 3>) + 256 <= 160 * 1024 && ldsc <= 150 * 1024;
        // (round 4: every window size -- the gather of a prechain solve forms the visual sub-space only, 551 workgroups at K = 20 instead of 1500)
        int kmerge = 20;
        if (const char* ev = VIL_TUNE_ENV("VIL_MERGE_K")) kmerge = atoi(ev);
        bool merged = can_pre && (c->launch_mode == 0 || c->launch_mode == 3 || c->launch_mode == 4) && K <= kmerge && std::max(lds3, ldsc) + step_static_lds((const void*)k_step<true, 3>) + 256 <= 160 * 1024 && VIL_TUNE_ENV("VIL_NO_MERGE") == nullptr;
        if (merged) {
            // the merged launch holds workgroups that spin on flags (master, helpers, one per W W^T tile) next to the finite ones they wait for (chain,
            // gather: lower block indices, dispatched first).  It is only taken when the device can hold every spinning workgroup AND one more at the
            // same time -- otherwise the waiters could occupy every slot before the last gather workgroup has found one (vil_coop.hpp)
            // (keyed by the LDS size: 42 kB at K = 10 is three workgroups per compute unit, 95 kB at K = 20 one -- a context that uploads a small window first must not check a large one against the small one's capacity)
            const size_t ldsm = std::max(lds3, ldsc);
            if (c->cap_step3 < 0 || c->cap_step3_lds != ldsm) { c->cap_step3 = vilcoop::capacity((const void*)k_step<true, 3>, VIL_STEP_THREADS, ldsm, c->device); c->cap_step3_lds = ldsm; }
            const int Tw = (int)(Tp_ * (Tp_ + 1) / 2);
            if (!fits_per_xcd(1 + P.n_help + Tw + 1, c->cap_step3)) merged = false;
        }
        P.prechain = merged ? 1 : (can_pre ? 2 : 0);
        c->n_ww = 0;
        if (P.prechain) {
            P.chain = 3;
            c->n_ww = (int)(Tp_ * (Tp_ + 1) / 2);
            if (merged) c->lds_step = std::max(lds3, ldsc);       // (the chain workgroup is one of the merged launch's)
            else {
                c->lds_step = lds3;
                c->lds_sweep = std::max(c->lds_sweep, ldsc);      // the chain workgroup rides in k_sweep
                { const int gs = grant_sweep(); if (gs != VIL_OK) return gs; }
                c->n_blocks_sweep += 1;
            }
        }
        c->P.chain = P.chain; c->P.chain_rs = P.chain_rs; c->P.prechain = P.prechain;
        // (the merged launch gathers 64 entries per 512-thread workgroup: half as many workgroups as the gather kernel's)
        const bool g64 = VIL_TUNE_ENV("VIL_GATHER32") == nullptr;
        c->n_gather_m = g64 ? gather_blocks(D, NV, 64, true) : gather_blocks(D, NV, RED_EPW, true);
        c->n_blocks_reduce_po = gather_blocks(D, NV, RED_EPW, true);
        c->P.rs_merged = merged ? (g64 ? 2 : 1) : 0; c->P.n_ww = c->n_ww; c->P.n_gather = merged ? c->n_gather_m : 0;
        // ---- the whole iteration in ONE launch (k_iter, vil_iter.hpp): whenever the merged gather + step launch is taken, the kernel's single dynamic-LDS size
        //      (the larger of the sweep roles' and the step roles' needs -- StepShared and the gather / tile scratch are carved from it) fits a compute unit, and
        //      the device holds the workgroups that wait for one another (master, helpers, tiles) at once.  vil_debug_set_launch_mode(3) keeps the two launches.
        c->fused = false; c->persist = false; c->P.n_sw = 0;
        if (merged && g64 && (c->launch_mode == 0 || c->launch_mode == 4) && c->n_blocks_sweep <= VIL_SFLAG_MAX && VIL_TUNE_ENV("VIL_NO_FUSE") == nullptr) {
            const size_t scratch = 8 * (size_t)(2 * VIS_TAB + 2 * 8 * (VIL_STEP_THREADS / 8) + 160);      // gather role: descriptor table | part[2][512] | index tables | red
            const size_t step_need = 8 * (size_t)VIL_SS_DOUBLES + std::max(std::max(lds3, ldsc), scratch);
            const size_t li = std::max(c->lds_sweep, step_need);
            const int v = P.vis_ts == 2 ? 0 : 1;
            const void* fn = v ? (const void*)k_iter<5> : (const void*)k_iter<2>;
            if (li <= 160 * 1024) {
                if ((int)li > c->attr_iter[v]) { HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)li)); c->attr_iter[v] = (int)li; }
                if (c->cap_iter[v] < 0 || c->cap_iter_lds[v] != li) { c->cap_iter[v] = vilcoop::capacity(fn, VIL_STEP_THREADS, li, c->device); c->cap_iter_lds[v] = li; }
                const int Tw = (int)(Tp_ * (Tp_ + 1) / 2);
                if (fits_per_xcd(1 + P.n_help + Tw + 1, c->cap_iter[v])) { c->fused = true; c->lds_iter = li; c->P.n_sw = c->n_blocks_sweep; }      // (+ the chain workgroup)
                // ---- the whole SOLVE in one resident launch (k_solve): the grid [sweep roles | chain | master | helpers | tiles] must fit the device at once with two
                //      workgroups to spare, and every gather item must find a workgroup that takes it as a duty (tiles, helpers, sweep roles): configs[1]-sized
                //      windows.  Everything else keeps one launch per iteration.  vil_debug_set_launch_mode(4) keeps k_iter.
                if (c->fused && c->launch_mode == 0 && VIL_TUNE_ENV("VIL_NO_PERSIST") == nullptr) {
                    const size_t sweep_need = 8 * (size_t)(VIL_LC_DOUBLES + VIL_XL_DOUBLES) + std::max(c->lds_sweep, 8 * (size_t)2048 + scratch);      // [Ctl copy | state copy | role arrays (IMU roles: gather scratch behind them)]
                    const size_t ls = std::max(sweep_need, step_need);
                    const void* fs = vil_k_solve_fn(P.vis_ts);
                    if (ls <= 160 * 1024 && K <= 15) {
                        if ((int)ls > c->attr_solve[v]) { HIPCHK(hipFuncSetAttribute(fs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ls)); c->attr_solve[v] = (int)ls; }
                        if (c->cap_solve[v] < 0 || c->cap_solve_lds[v] != ls) { c->cap_solve[v] = vilcoop::capacity(fs, VIL_STEP_THREADS, ls, c->device); c->cap_solve_lds[v] = ls; }
                        const int grid = c->n_blocks_sweep + 2 + P.n_help + Tw, n_cap = c->n_blocks_sweep + P.n_help + Tw;
                        if (fits_per_xcd(grid, c->cap_solve[v]) && c->n_gather_m <= n_cap) { c->persist = true; c->lds_solve = ls; }
                    }
                }
            }
        }
    }
    if (P.chain) {
        c->step_lds = true;
        if (P.chain == 3) {
            if ((int)c->lds_step > c->attr_step[4]) { HIPCHK(hipFuncSetAttribute((const void*)k_step<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_step)); c->attr_step[4] = (int)c->lds_step; }
        } else if (P.chain == 1) {
            if ((int)c->lds_step > c->attr_step[1]) { HIPCHK(hipFuncSetAttribute((const void*)k_step<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_step)); c->attr_step[1] = (int)c->lds_step; }
        } else {
            if ((int)c->lds_step > c->attr_step[2]) { HIPCHK(hipFuncSetAttribute((const void*)k_step<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_step)); c->attr_step[2] = (int)c->lds_step; }
        }
    } else {
    { const size_t T = (size_t)(D + 1 + 15) / 16; c->lds_step = 8 * TILE_SZ * (T * (T + 1) / 2); }   // 16x16-tiled (row stride 17) lower storage incl. the rhs row
    c->step_lds = c->lds_step + step_static_lds((const void*)k_step<true, 0>) + 256 <= 160 * 1024;
    if (c->step_lds) {
        if ((int)c->lds_step > c->attr_step[0]) { HIPCHK(hipFuncSetAttribute((const void*)k_step<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_step)); c->attr_step[0] = (int)c->lds_step; }
    } else {
        // tile array in global memory; LDS stages the active tile column of the factorisation (T tiles)
        const size_t T = (size_t)(D + 1 + 15) / 16;
        c->lds_step = 8 * (size_t)TILE_SZ * T;
        if (c->lds_step + step_static_lds((const void*)k_step<false, 0>) + 256 > 160 * 1024) return VIL_ERR_UNSUPPORTED;
        if ((int)c->lds_step > c->attr_step[3]) { HIPCHK(hipFuncSetAttribute((const void*)k_step<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_step)); c->attr_step[3] = (int)c->lds_step; }
    }
    }
    UPTICK("h2d+attrs");
    // one-time set-up: IMU sqrt-information, prior contraction
    if (!ws) HIPCHK(hipMemsetAsync(c->d_status, 0, sizeof(int), c->stream));      // (resident window: nothing writes it -- its status word is the window's own, vil_win_solve -- and a 4-byte fill is a 5 us launch)
    if (ws) {
        // resident window: expand the landmark table into the factor tables, gather the IMU records, copy the state -- everything k_setup
        // would compute (sqrt-information, prior contractions) already sits next to its source
        WinPack W; memset(&W, 0, sizeof W);
        W.L = L; W.F = n_vis; W.stride = P.vis_stride; W.T = ws->T;
        W.lm_start = P.lm_start; W.lm_track = wd_track; W.lm_startf = wd_startf; W.store = ws->d_store;
        for (int k = 0; k < K; ++k) { W.fslot[k] = ws->fslot[k]; W.islot[k] = ws->islot[k]; }
        W.vis_c = const_cast<double*>(P.vis_c); W.vis_i = const_cast<int*>(P.vis_i); W.vis_j = const_cast<int*>(P.vis_j); W.vis_l = const_cast<int*>(P.vis_l); W.fcol = const_cast<int*>(P.fcol); W.vfinv = P.vfinv;
        W.n_imu = P.n_imu; W.rec = ws->d_rec; W.U = ws->d_U; W.imu_c = const_cast<double*>(P.imu_c); W.imu_U = const_cast<double*>(P.imu_U);
        W.NS = NS; W.x0 = P.x[0]; W.x1 = P.x[1]; W.xorig = c->d_x0;
        const int nbf = (n_vis + 255) / 256, nbs = (NS + 255) / 256;
        hipLaunchKernelGGL(k_win_pack, dim3(nbf + P.n_imu + nbs), dim3(256), 0, c->stream, W, nbf);
    }
    const int nb_setup = ws ? 0 : P.n_imu + (P.pn ? 64 : 0);
    if (nb_setup > 0) hipLaunchKernelGGL(k_setup, dim3(nb_setup), dim3(VIL_THREADS), 0, c->stream, P, const_cast<double*>(P.imu_U), c->d_status);
    if (check_setup) {
        c->h_word[0] = 0;
        HIPCHK(hipMemcpyAsync(c->h_word, c->d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        HIPCHK(hipGetLastError());
        if (c->h_word[0] != 0) return VIL_ERR_NOT_POSITIVE_DEFINITE;
    }
    UPTICK("setup");
    c->uploaded = true; c->resident_kind = 2;    // vil_upload / vil_solve promote it to 1
    return VIL_OK;
}

static int comm_agree(vil_ctx* c, int st);
static int ensure_pin(vil_ctx* c, size_t bytes);

// rank r keeps the visual factors of its landmark range, a contiguous slice of the LiDAR points, and
// (rank 0 only) the IMU / prior / ICP / LPS factors.  Landmark indices and the state stay global.
static int upload_sharded(vil_ctx* c, const vil_problem* p, const vil_state* s) {
    int32_t lb, le, eb, ee, pb, pe;
    int st = validate(p, s);                     // vil_shard_ranges indexes with vis_l: check the tables first
    if (st == VIL_OK) st = vil_shard_ranges(p, c->rank, c->world, &lb, &le, &eb, &ee, &pb, &pe);
    if (st != VIL_OK) return comm_agree(c, st);  // every rank sees the same problem, but stay collective anyway
    vil_problem q = *p;
    int f0 = 0, f1 = 0;
    for (int f = 0; f < p->n_vis; ++f) { if (p->vis_l[f] < lb) f0 = f + 1; if (p->vis_l[f] < le) f1 = f + 1; }
    q.n_vis = f1 - f0; q.vis_i = p->vis_i + f0; q.vis_j = p->vis_j + f0; q.vis_l = p->vis_l + f0; q.vis_const = p->vis_const + (size_t)f0 * 14;
    if (p->n_plane == VIL_LIDAR_RESIDENT && p->n_edge == VIL_LIDAR_RESIDENT) { /* the slabs of this context already hold this rank's slice of every frame (vil_lidar_push) */ }
    else {
        q.n_edge = ee - eb; q.edge_pose = p->edge_pose + eb; q.edge_const = p->edge_const + (size_t)eb * 9;
        q.n_plane = pe - pb; q.plane_pose = p->plane_pose + pb; q.plane_const = p->plane_const + (size_t)pb * 7;
    }
    if (c->rank != 0) { q.n_imu = 0; q.n_icp = 0; q.n_lps = 0; q.prior.n = 0; q.prior.nblk = 0; }
    c->lm_b = lb; c->lm_e = le;
    {   // who owns which slice of the landmark arrays of the message (the same arithmetic on every rank)
        OwnSeg& S = c->own; memset(&S, 0, sizeof S);
        S.n = c->world <= 8 ? c->world : 0; S.Lp = std::max(p->L, 1);      // (more than eight ranks: RCCL only, which sums the whole set -- n = 0)
        for (int r = 0; r < S.n; ++r) {
            int32_t b = 0; vil_shard_ranges(p, r, c->world, &b, nullptr, nullptr, nullptr, nullptr, nullptr);
            S.lb[r] = b; int fb = 0; for (int f = 0; f < p->n_vis; ++f) if (p->vis_l[f] < b) fb = f + 1;
            S.fb[r] = fb;
        }
        S.lb[S.n] = p->L; S.fb[S.n] = p->n_vis;
        const int D = 15 * p->K + 7;
        S.cam = ((size_t)D * D + 3 * (size_t)D + 3 + 1) & ~size_t(1);
    }
    st = comm_agree(c, upload_impl(c, &q, s, true, nullptr, p, f0));       // e.g. rank 0's IMU set-up failed: every rank reports it
    if (st != VIL_OK) c->uploaded = false;
    return st;
}

static int upload_window(vil_ctx* c, const vil_problem* p, const vil_state* s, bool check_setup) {
    if (!c) return VIL_ERR_INVALID_ARGUMENT;
    if (c->ipc && c->world > 1 && !c->ipc->ready) return VIL_ERR_COMM;             // vil_comm_ipc_export without vil_comm_ipc_init
    const int st = (c->world > 1 && c->has_comm()) ? upload_sharded(c, p, s)      // a world > 1 context without a communicator works un-sharded
                                                           : upload_impl(c, p, s, false, nullptr, nullptr, 0, check_setup);
    if (st == VIL_OK) c->resident_kind = 1;
    return st;
}
int vil_upload(vil_ctx* c, const vil_problem* p, const vil_state* s) { return upload_window(c, p, s, true); }

// sum over the ranks of the communicator, in stream order: recv = sum of every rank's send (recv may be send)
static const OwnSeg kNoSeg = {0, 0, 0, {0}, {0}};
static int ipc_all_reduce(vil_ctx* c, const double* send, double* recv, size_t cnt, int sym = 0, const OwnSeg& seg = kNoSeg) {
    IpcComm* ic = c->ipc.get();
    if (!ic->ready) return VIL_ERR_COMM;                // vil_comm_ipc_export without vil_comm_ipc_init: the peers' inboxes are not mapped yet
    if (cnt > ic->cap) return VIL_ERR_UNSUPPORTED;      // the inbox was sized at vil_comm_ipc_export
    IpcPtrs I; memset(&I, 0, sizeof I);
    for (int r = 0; r < ic->world; ++r) { I.inbox[r] = ic->inbox(r); I.flags[r] = ic->flags(r); }
    I.world = ic->world; I.rank = ic->rank; I.cap = ic->cap; I.seq = ic->seq(); I.count = ic->seq() + 16;
    const unsigned nb = (unsigned)std::min<size_t>(128, (cnt + 255) / 256);
    hipLaunchKernelGGL(k_ipc_push, dim3(nb), dim3(256), 0, c->stream, send, cnt, I, sym, seg);
    hipLaunchKernelGGL(k_ipc_wait, dim3(1), dim3(64), 0, c->stream, I);
    hipLaunchKernelGGL(k_ipc_sum, dim3(nb), dim3(256), 0, c->stream, recv, cnt, I, sym, seg);
    return VIL_OK;
}
static SlimLay slim_layout(const vil_ctx* c) {
    SlimLay Y; memset(&Y, 0, sizeof Y);
    const OwnSeg& S = c->own;
    Y.D = c->D; Y.Lp = S.Lp; Y.n = S.n; Y.rank = c->rank; Y.cam = S.cam; Y.span = c->span;
    Y.camS = ((size_t)Y.D * (Y.D + 1) / 2 + (S.cam - (size_t)Y.D * Y.D) + 1) & ~size_t(1);
    for (int r = 0; r <= S.n; ++r) { Y.lb[r] = S.lb[r]; Y.fb[r] = S.fb[r]; }
    for (int r = 0; r < S.n; ++r) Y.gmax = std::max(Y.gmax, (size_t)17 * (S.lb[r + 1] - S.lb[r]) + (size_t)6 * (S.fb[r + 1] - S.fb[r]));
    Y.gmax = (Y.gmax + 2) & ~size_t(1);
    return Y;
}
// the per-iteration collective as a packed message: pack -> all-reduce of M + all-gather of the owners' slices -> unpack (see SlimLay)
static int slim_all_reduce(vil_ctx* c, const double* send, double* recv) {
    const SlimLay Y = slim_layout(c);
    const size_t need = 2 * Y.camS + (size_t)(1 + c->world) * Y.gmax;
    LocalComm* lc = c->comm ? nullptr : c->lcomm.get();
    int st = VIL_OK;
    if (need > c->slim_cap) {
        if (c->slim_buf) hipFree(c->slim_buf);
        c->slim_buf = nullptr; c->slim_cap = 0;
        if (hipMalloc(&c->slim_buf, 8 * need) == hipSuccess) { c->slim_cap = need; if (hipMemsetAsync(c->slim_buf, 0, 8 * need, c->stream) != hipSuccess) st = VIL_ERR_DEVICE; }
        else st = VIL_ERR_DEVICE;
        if (st != VIL_OK && !lc) return st;          // (in-process ranks carry a failure to the agreement point below)
    }
    double* M = c->slim_buf; double* G = M + Y.camS; double* Msum = G + Y.gmax; double* Gall = Msum + Y.camS;
    const unsigned nb = (unsigned)std::min<size_t>(256, (Y.span + 255) / 256);
    if (st == VIL_OK) hipLaunchKernelGGL(k_slim_pack, dim3(nb), dim3(256), 0, c->stream, send, M, G, Y);
    if (!lc) {
        if (g_rccl.AllReduce(M, Msum, Y.camS, ncclDouble, ncclSum, c->comm, c->stream) != ncclSuccess) return VIL_ERR_COMM;
        if (g_rccl.AllGather(G, Gall, Y.gmax, ncclDouble, c->comm, c->stream) != ncclSuccess) return VIL_ERR_COMM;
    } else {
        if (hipStreamSynchronize(c->stream) != hipSuccess) st = VIL_ERR_DEVICE;
        lc->ptr[c->rank] = c->slim_buf;
        st = lc->agree(c->rank, st);                          // every rank's M and G are complete and published
        if (st != VIL_OK) return st;
        PeerPtrs pp; pp.n = lc->n;
        for (int r = 0; r < 8; ++r) pp.p[r] = r < lc->n ? lc->ptr[r] : nullptr;
        hipLaunchKernelGGL(k_slim_emul, dim3(nb), dim3(256), 0, c->stream, Msum, Gall, pp, Y);
        if (hipStreamSynchronize(c->stream) != hipSuccess) st = VIL_ERR_DEVICE;
        st = lc->agree(c->rank, st);                          // every rank has read every staging buffer
        if (st != VIL_OK) return st;
    }
    hipLaunchKernelGGL(k_slim_unpack, dim3(nb), dim3(256), 0, c->stream, recv, (const double*)Msum, (const double*)Gall, Y);
    return VIL_OK;
}
static int all_reduce2(vil_ctx* c, const double* send, double* recv, size_t cnt, int sym = 0, const OwnSeg& seg = kNoSeg) {
    if (c->ipc) return ipc_all_reduce(c, send, recv, cnt, sym, seg);
    if (seg.n > 0 && sym > 0 && cnt == c->span && c->slim()) return slim_all_reduce(c, send, recv);
    if (c->comm) return g_rccl.AllReduce(send, recv, cnt, ncclDouble, ncclSum, c->comm, c->stream) == ncclSuccess ? VIL_OK : VIL_ERR_COMM;
    if (c->lcomm) {
        LocalComm* lc = c->lcomm.get();
        int st = VIL_OK;
        const bool inplace = send == recv;
        if (inplace && cnt > c->lc_cap) { if (c->lc_tmp) hipFree(c->lc_tmp); c->lc_tmp = nullptr; c->lc_cap = 0; if (hipMalloc(&c->lc_tmp, 8 * cnt) == hipSuccess) c->lc_cap = cnt; else st = VIL_ERR_DEVICE; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) st = VIL_ERR_DEVICE;
        lc->ptr[c->rank] = const_cast<double*>(send);
        st = lc->agree(c->rank, st);                          // all buffers are complete and published (or somebody failed)
        if (st != VIL_OK) return st;
        PeerPtrs pp; pp.n = lc->n;
        for (int r = 0; r < 8; ++r) pp.p[r] = r < lc->n ? lc->ptr[r] : nullptr;
        hipLaunchKernelGGL(k_sum_peers, dim3((unsigned)std::min<size_t>(256, (cnt + 255) / 256)), dim3(256), 0, c->stream, inplace ? c->lc_tmp : recv, pp, cnt, sym, seg);
        if (hipStreamSynchronize(c->stream) != hipSuccess) st = VIL_ERR_DEVICE;
        st = lc->agree(c->rank, st);                          // every rank has read every buffer
        if (st != VIL_OK) return st;
        if (inplace) HIPCHK(hipMemcpyAsync(recv, c->lc_tmp, 8 * cnt, hipMemcpyDeviceToDevice, c->stream));
        return VIL_OK;
    }
    if (send != recv) HIPCHK(hipMemcpyAsync(recv, send, 8 * cnt, hipMemcpyDeviceToDevice, c->stream));     // a single rank (VIL_FORCE_SPLIT)
    return VIL_OK;
}
static int all_reduce(vil_ctx* c, double* buf, size_t cnt) {
    if (c->ipc) return ipc_all_reduce(c, buf, buf, cnt);      // (push reads, sum writes: different kernels, stream order)
    if (c->comm) return g_rccl.AllReduce(buf, buf, cnt, ncclDouble, ncclSum, c->comm, c->stream) == ncclSuccess ? VIL_OK : VIL_ERR_COMM;
    if (c->lcomm) {
        // every HIP failure is carried to the next agreement point instead of returning past a barrier the other ranks wait at
        LocalComm* lc = c->lcomm.get();
        int st = VIL_OK;
        if (cnt > c->lc_cap) { if (c->lc_tmp) hipFree(c->lc_tmp); c->lc_tmp = nullptr; c->lc_cap = 0; if (hipMalloc(&c->lc_tmp, 8 * cnt) == hipSuccess) c->lc_cap = cnt; else st = VIL_ERR_DEVICE; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) st = VIL_ERR_DEVICE;
        lc->ptr[c->rank] = buf;
        st = lc->agree(c->rank, st);                          // all buffers are complete and published (or somebody failed)
        if (st != VIL_OK) return st;
        PeerPtrs pp; pp.n = lc->n;
        for (int r = 0; r < 8; ++r) pp.p[r] = r < lc->n ? lc->ptr[r] : nullptr;
        hipLaunchKernelGGL(k_sum_peers, dim3((unsigned)std::min<size_t>(256, (cnt + 255) / 256)), dim3(256), 0, c->stream, c->lc_tmp, pp, cnt, 0, kNoSeg);
        if (hipStreamSynchronize(c->stream) != hipSuccess) st = VIL_ERR_DEVICE;
        st = lc->agree(c->rank, st);                          // every rank has read every buffer
        if (st != VIL_OK) return st;
        HIPCHK(hipMemcpyAsync(buf, c->lc_tmp, 8 * cnt, hipMemcpyDeviceToDevice, c->stream));
    }
    return VIL_OK;
}

// the ranks of a communicator agree on a status (the worst one): a rank-asymmetric failure -- only rank 0 holds the IMU / prior
// factors whose set-up can fail -- must not leave the other ranks waiting in the first collective of the solve
static int comm_agree(vil_ctx* c, int st) {
    if (c->lcomm) return c->lcomm->agree(c->rank, st);
    if (c->ipc) {                                          // every rank's status in its own slot of a world-long message: the worst one wins
        const int w = c->ipc->world;
        if (ensure_pin(c, 8 * 64) != VIL_OK) return VIL_ERR_DEVICE;
        double* h = c->h_pin; for (int r = 0; r < w; ++r) h[r] = r == c->ipc->rank ? (double)st : 0.0;
        if (!c->ipc_tmp && hipMalloc(&c->ipc_tmp, 8 * 64) != hipSuccess) return VIL_ERR_DEVICE;
        if (hipMemcpyAsync(c->ipc_tmp, h, 8 * (size_t)w, hipMemcpyHostToDevice, c->stream) != hipSuccess) return VIL_ERR_DEVICE;
        if (ipc_all_reduce(c, c->ipc_tmp, c->ipc_tmp + 32, (size_t)w) != VIL_OK) return VIL_ERR_COMM;
        if (hipMemcpyAsync(h + 32, c->ipc_tmp + 32, 8 * (size_t)w, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return VIL_ERR_DEVICE;
        if (hipStreamSynchronize(c->stream) != hipSuccess) return VIL_ERR_DEVICE;
        int worst = 0; for (int r = 0; r < w; ++r) worst = std::min(worst, (int)h[32 + r]);
        return worst;
    }
    if (c->comm) {
        int* h = c->h_word + 4; *h = st;
        if (hipMemcpyAsync(c->d_status, h, sizeof(int), hipMemcpyHostToDevice, c->stream) != hipSuccess) return VIL_ERR_DEVICE;
        if (g_rccl.AllReduce(c->d_status, c->d_status, 1, ncclInt, ncclMin, c->comm, c->stream) != ncclSuccess) return VIL_ERR_COMM;
        if (hipMemcpyAsync(h, c->d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return VIL_ERR_DEVICE;
        if (hipStreamSynchronize(c->stream) != hipSuccess) return VIL_ERR_DEVICE;
        return *h;
    }
    return st;
}

// Multi-GPU: ONE collective per trust-region iteration.  Every rank sweeps its shard into set 0 -- the reduced camera
// system of its factors plus the landmark arrays (h_ll, b_l, 1/pivot, scale, e_l) of the landmarks it owns, zeros elsewhere -- the
// whole set is all-reduced into set 1, and every rank runs the complete single-GPU step kernel (helper workgroups, chain path)
// on identical data: identical decisions, identical candidates, no second exchange.  view 0: what the sweep / gather write
// (both buffer slots alias set 0: the partial system is consumed by the collective at once); view 1: what the step kernel reads.
static DevP view(const vil_ctx* c, int which) {
    DevP P = c->P;
    if (c->split) { P.sys[0] = P.sys[1] = c->P.sys[which]; if (which == 1) { P.rank = 0; P.world = 1; } }
    return P;
}
static int launch_sweep(vil_ctx* c, const SolveOpts& so) {
    if (c->P.vis_ts == 2) hipLaunchKernelGGL(k_sweep<2>, dim3(c->n_blocks_sweep), dim3(VIL_SWEEP_THREADS), c->lds_sweep, c->stream, view(c, 0), so);
    else hipLaunchKernelGGL(k_sweep<5>, dim3(c->n_blocks_sweep), dim3(VIL_SWEEP_THREADS), c->lds_sweep, c->stream, view(c, 0), so);
    return VIL_OK;
}
// one trust-region iteration as ONE launch (vil_iter.hpp); only un-sharded solves of a window the upload marked `fused`
static int launch_iter(vil_ctx* c, const SolveOpts& so) {
    DevP Pi = c->P;
    Pi.gather_pose_only = 1;
    Pi.prof = c->profiling ? c->d_prof : nullptr; Pi.wg_launch = c->profiling ? c->wg_launch : -1;
    const dim3 g(c->n_blocks_sweep + 1 + c->n_gather_m + 1 + c->P.n_help + c->n_ww), b(VIL_STEP_THREADS);      // [sweep roles | chain | master | helpers | W W^T tiles | gather]
    if (c->P.vis_ts == 2) hipLaunchKernelGGL(k_iter<2>, g, b, c->lds_iter, c->stream, Pi, so);
    else hipLaunchKernelGGL(k_iter<5>, g, b, c->lds_iter, c->stream, Pi, so);
    return VIL_OK;
}
// the whole solve as ONE resident launch (vil_iter.hpp, k_solve); budget_ticks: what is left of max_time_s on the device's 100 MHz clock (0: no cap)
static int launch_solve(vil_ctx* c, const SolveOpts& so, long long budget_ticks) {
    DevP Pi = c->P;
    Pi.gather_pose_only = 1; Pi.prof = c->stamps ? c->d_prof : nullptr; Pi.wg_launch = -1; Pi.persist = 1;
    vil_k_solve_launch(c->P.vis_ts, (unsigned)(c->n_blocks_sweep + 2 + c->P.n_help + c->n_ww), c->lds_solve, c->stream, Pi, so, budget_ticks);      // [sweep roles | chain | master | helpers | W W^T tiles]
    return VIL_OK;
}
static int launch_reduce_step(vil_ctx* c, const SolveOpts& so, bool step, hipEvent_t ev_mid = nullptr, hipEvent_t ev_coll = nullptr) {
    const bool merged = step && c->P.rs_merged;          // one GPU: the gather rides in the step kernel's launch (vil_step.hpp)
    // (chain eliminated inside k_sweep: one workgroup per W W^T tile rides in the gather launch, one for the inverses of the chain's diagonal blocks in the step launch;
    //  a solve on the prechain path reads S' on the visual sub-space + the diagonal only -- vil_linearize, the marginalisation and sharded solves all of it)
    if (!merged) {
        DevP Pg = view(c, 0);
        const bool po = step && c->P.prechain != 0 && !c->split;
        Pg.gather_pose_only = po ? 1 : 0;
        const int ng = po ? c->n_blocks_reduce_po : c->n_blocks_reduce;
        hipLaunchKernelGGL(k_reduce, dim3(ng + ((step && c->P.prechain == 2) ? c->n_ww : 0)), dim3(VIL_THREADS), 0, c->stream, Pg, ng);
    }
    if (ev_mid) hipEventRecord(ev_mid, c->stream);
    if (c->split) {                                    // the one collective of the iteration
        const int st = all_reduce2(c, c->P.sys[0].ar, c->P.sys[1].ar, c->span, c->D, c->own);      // (S' first: its lower triangle travels; landmark arrays: the owners' slices)
        if (st != VIL_OK) return st;
    }
    if (ev_coll) hipEventRecord(ev_coll, c->stream);
    if (!step) return VIL_OK;
    DevP Ps = view(c, 1);
    if (!merged) { Ps.rs_merged = 0; Ps.n_ww = 0; Ps.n_gather = 0; }
    Ps.gather_pose_only = merged ? 1 : 0;
    const dim3 g(1 + c->P.n_help + (merged ? (c->P.prechain ? 1 : 0) + c->n_ww + c->n_gather_m : (c->P.prechain == 2 ? 1 : 0))), b(VIL_STEP_THREADS);      // (prechain 2: + prechain_inverses)
    if (c->P.chain == 3) hipLaunchKernelGGL((k_step<true, 3>), g, b, c->lds_step, c->stream, Ps, so);
    else if (c->P.chain == 1) hipLaunchKernelGGL((k_step<true, 1>), g, b, c->lds_step, c->stream, Ps, so);
    else if (c->P.chain == 2) hipLaunchKernelGGL((k_step<true, 2>), g, b, c->lds_step, c->stream, Ps, so);
    else if (c->step_lds) hipLaunchKernelGGL((k_step<true, 0>), g, b, c->lds_step, c->stream, Ps, so);
    else hipLaunchKernelGGL((k_step<false, 0>), g, b, c->lds_step, c->stream, Ps, so);
    return VIL_OK;
}

// vil_reset_state is deferred: whoever touches the resident state next restores it first (a solve inside its init launch)
static void flush_reset(vil_ctx* c) {
    if (!c->reset_pending) return;
    hipLaunchKernelGGL(k_state_reset, dim3((c->NS + 255) / 256), dim3(256), 0, c->stream, c->P.x[0], c->P.x[1], (const double*)c->d_x0, c->NS);
    c->reset_pending = false;
}
static int init_ctl(vil_ctx* c, const vil_options* o, int lin_mode) {
    // (a kernel with the values in its arguments: an asynchronous copy out of the one pinned h_ctl could be overtaken by the next call's init --
    //  vil_win_marginalize / push / drop return without a stream synchronisation)
    static_assert(sizeof(Ctl) % 8 == 0, "Ctl is cleared as doubles");
    if (c->reset_pending && lin_mode == 0) {               // vil_reset_state + vil_solve_resident: the state copy rides in the init launch
        hipLaunchKernelGGL(k_solve_init_reset, dim3((c->NS + 255) / 256), dim3(256), 0, c->stream, c->P.ctl, ++c->solve_gen, o->initial_radius, o->min_mu, lin_mode, c->P.x[0], c->P.x[1], (const double*)c->d_x0, c->NS, c->P.abortf, c->d_xsave);
        c->reset_pending = false;
    } else {
        flush_reset(c);
        double* const xs = lin_mode == 0 ? c->d_xsave : nullptr;      // (a solve keeps its start state aside: vil_solve_resident's retry / failure path)
        hipLaunchKernelGGL(k_solve_init, dim3(xs ? (c->NS + 255) / 256 : 1), dim3(256), 0, c->stream, c->P.ctl, ++c->solve_gen, o->initial_radius, o->min_mu, lin_mode, c->P.abortf, (const double*)c->P.x[0], xs, c->NS);
    }
    memset(c->h_ctl, 0, sizeof(Ctl));
    return VIL_OK;
}

int vil_profile_enable(vil_ctx* c, int on) {
    if (!c) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    if (on && c->ev.empty()) { c->ev.resize(2 * VIL_MAX_CHUNK + 2); for (auto& e : c->ev) HIPCHK(hipEventCreate(&e)); c->ev_mid.resize(VIL_MAX_CHUNK + 1); for (auto& e : c->ev_mid) HIPCHK(hipEventCreate(&e)); c->ev_coll.resize(VIL_MAX_CHUNK + 1); for (auto& e : c->ev_coll) HIPCHK(hipEventCreate(&e)); }
    if (on && !c->d_prof) { HIPCHK(hipMalloc((void**)&c->d_prof, 8 * (64 * VIL_PROF_SLOTS + 2 * VIL_PROF_WGS))); HIPCHK(hipMemset(c->d_prof, 0, 8 * (64 * VIL_PROF_SLOTS + 2 * VIL_PROF_WGS))); }
    c->profiling = on == 1;
    c->stamps = on == 2;          // 2: the phase stamps alone -- the launch structure stays the library's choice (the persistent solve keeps its one launch; no events)
    return VIL_OK;
}
/* average position (us after the launch's first workgroup started) of the phase stamps of the one-launch iterations timed since the last reset:
 * [0] 0, [1] last visual / LiDAR / ICP-LPS role done, [2] last IMU role done, [3] chain: records seen, [4] chain: W^T complete, [5] last gather workgroup done,
 * [6] last gather workgroup saw the visual flags, [7] master started, [8] master saw the gather's flags, [9] master saw the W W^T tiles,
 * [10] dense factorisation done, [11] x_p published, [12] master done, [13] last tile workgroup done, [14] chain: its part of S' gathered, [15] prior role done,
 * [16 .. 20] IMU role of factor 0: entered, inputs staged, raw blocks done, whitened, record stores issued; [21 .. 23] master: chain back-substituted (solve done),
 * step vectors + helpers' sums in, candidate formed */
int vil_profile_phases(vil_ctx* c, double* avg_us, int64_t* launches, int reset) {
    if (!c || !avg_us) return VIL_ERR_INVALID_ARGUMENT;
    for (int k = 0; k < VIL_PROF_SLOTS; ++k) avg_us[k] = c->phase_n ? c->phase_us[k] / (double)c->phase_n : 0.0;
    avg_us[0] = c->period_n ? c->phase_us[0] / (double)c->period_n : 0.0;      // [0]: average iteration period inside a persistent solve (0: launches)
    if (launches) *launches = c->phase_n;
    if (reset) { for (double& v : c->phase_us) v = 0.0; c->phase_n = 0; c->period_n = 0; }
    return VIL_OK;
}
int vil_profile_read(vil_ctx* c, vil_profile* out, int reset) {
    if (!c || !out) return VIL_ERR_INVALID_ARGUMENT;
    *out = c->prof;
    if (reset) c->prof = vil_profile{0, 0.0, 0, 0.0, 0.0, 0.0};
    return VIL_OK;
}

int vil_debug_read_stamps(vil_ctx* c, uint64_t* out, int32_t max_launches) {
    if (!c || !out || max_launches < 0) return VIL_ERR_INVALID_ARGUMENT;
    const size_t n = std::min((size_t)max_launches * VIL_PROF_SLOTS, c->last_stamps.size());
    for (size_t i = 0; i < n; ++i) out[i] = c->last_stamps[i];
    return (int)(n / VIL_PROF_SLOTS);
}
int vil_debug_marg_stamps(vil_ctx* c, uint64_t* out16) {
    if (!c || !out16 || !c->d_marg_ts) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(out16, c->d_marg_ts, 8 * 16, hipMemcpyDeviceToHost));
    return VIL_OK;
}
int vil_debug_read(vil_ctx* c, long long* out64) {
    if (!c || !c->uploaded) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipMemcpy(out64, c->P.dbg, 8 * 64, hipMemcpyDeviceToHost));
    return VIL_OK;
}

int vil_reset_state(vil_ctx* c) {
    if (!c || !c->uploaded) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    c->reset_pending = true;                                 // (carried out by the next call that touches the state: flush_reset / init_ctl)
    c->mirror_state = false;
    return VIL_OK;
}

// One attempt at the solve with the launch structure the context holds right now (c->fused: one launch per iteration; else sweep + gather / step launches).
// *gave_up: a wait inside a launch gave up (status -2 from the master, or no launch ever reported `done`): the caller decides about the retry.
static int solve_attempt(vil_ctx* c, const vil_options* o, vil_summary* sum, const std::chrono::steady_clock::time_point t0, const bool direct, bool* gave_up) {
    const SolveOpts so = to_dev_opts(o);
    c->mirror_state = false;
    *gave_up = false;
    const bool persist = c->persist && c->fused && !c->split && !c->profiling;      // (the phase stamps and the per-launch events belong to the one-launch iteration)
    int st = init_ctl(c, o, 0);
    if (st != VIL_OK) return st;
    if ((c->profiling || (c->stamps && persist)) && c->fused && c->d_prof) HIPCHK(hipMemsetAsync(c->d_prof, 0, 8 * 64 * VIL_PROF_SLOTS, c->stream));
    // every iteration = sweep + gather + step kernel; `done` turns the tail of a chunk into no-ops, and the first sweep launch that finds
    // the solve finished writes the result out (vil_finish.hpp); k_finish at the end of every chunk covers a solve that ends in its last iteration
    bool finished = false, polled_done = false;
    // iterations are enqueued in chunks without host round trips; the first chunk is sized by the previous solve of
    // this context (consecutive windows of a tracker need similar iteration counts), later chunks are short
    int chunk = std::min(15, std::max(3, c->last_live));
    // One-launch iterations write the result out themselves as soon as a launch finds the solve finished (vil_iter.hpp): a chunk that is too LONG costs the stream
    // ~5 us per dead launch and the host nothing, one that is too SHORT costs a host round trip (~70 us) and a second chunk.  The first solve of an upload -- what a
    // tracker runs per image, iteration counts wandering by one or two from image to image -- therefore enqueues the largest count of the last eight solves plus two;
    // re-solves of one upload (graph replay, the same count again and again) keep the exact size.
    if (c->fused && c->solves_since_upload == 0) { int mx = 3; for (int v : c->recent_live) mx = std::max(mx, v); chunk = std::min(VIL_MAX_CHUNK, mx + 2); }
    if (c->profiling) chunk = std::min(chunk, (int)c->ev.size() / 2 - 1);      // (events bracket every launch of a chunk: ev[2 q], ev[2 q + 1], ev[2 launched])
    for (int it = 0; it <= o->max_iterations + 8 && !finished; chunk = 3) {
        int launched = 0;
        const int sweeps_before = (it == 0) ? 0 : c->h_ctl->n_sweeps;
        // Repeated solves of ONE upload (bench, re-solves after a rejected frame) replay a captured hipGraph of the chunk:
        // ~2 % less inter-kernel gap.  The first solve of an upload launches directly -- capturing costs more than it saves.
        if (c->use_graph < 0) { const char* ev = VIL_TUNE_ENV("VIL_GRAPH"); c->use_graph = ev ? atoi(ev) : 1; }
        const int nthis = std::min(chunk, o->max_iterations + 9 - it);
        // hipGraph replay: un-sharded solves, and sharded ones whose collective is the library's own kernels (peer-buffer exchange).  RCCL calls are
        // launched directly (whether a given RCCL build captures correctly is not something this path bets the multi-GPU run on); the in-process
        // communicator synchronises on the host.  Polling the finished solve's mirror works for everything that is in stream order.
        const bool no_graph = c->split && !c->ipc;
        if (c->use_graph && c->solves_since_upload > 0 && !c->profiling && !no_graph && !c->graph_failed && !direct && !persist && nthis > 0) {      // (direct: a retry, or a solve with a debug hook in its parameter block)
            hipGraphExec_t exec = nullptr;
            for (auto& g : c->graphs) if (g.n == nthis && memcmp(&g.so, &so, sizeof so) == 0) exec = g.exec;
            if (!exec) {
                hipGraph_t graph = nullptr;
                HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                int cst = VIL_OK;
                for (int q = 0; q < nthis && cst == VIL_OK; ++q) { if (c->fused) cst = launch_iter(c, so); else { launch_sweep(c, so); cst = launch_reduce_step(c, so, true, nullptr); } }
                hipLaunchKernelGGL(k_finish, dim3(1), dim3(VIL_SWEEP_THREADS), 0, c->stream, view(c, 0), -1);
                const hipError_t ce = hipStreamEndCapture(c->stream, &graph);
                const bool forced = c->fail_capture > 0;
                if (forced) --c->fail_capture;
                if (cst != VIL_OK || ce != hipSuccess || forced || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
                    // a chunk that cannot be captured or instantiated (a collective the RCCL build at hand does not capture, a driver out of graph memory): nothing has
                    // run yet, so THIS solve and every later one of the context take the direct launches right below -- the caller (optimization()) has no retry
                    (void)hipGetLastError();
                    if (graph) hipGraphDestroy(graph);
                    exec = nullptr; c->graph_failed = true;
                    // (a capture that could not be ENDED may leave the stream in an invalidated capture state: the direct launches below would fail the same way.
                    //  The collectives advance no host-side state while captured -- the peer-buffer exchange counts in device memory, when its kernels run)
                    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                    if (ce != hipSuccess && (hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)) { (void)hipGetLastError(); return VIL_ERR_DEVICE; }
                } else {
                    hipGraphDestroy(graph);
                    c->graphs.push_back({nthis, so, exec});
                }
            }
            if (exec) { HIPCHK(hipGraphLaunch(exec, c->stream)); it += nthis; launched = nthis; }
        }
        if (launched == 0 && persist) {
            // ONE launch runs every iteration and writes the result out; the time cap travels with it (the master reads the device clock where ceres reads its own)
            long long ticks = 0;
            if (o->max_time_s > 0) ticks = std::max(1LL, (long long)((o->max_time_s - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()) * 1e8));
            if (c->stamps && !c->ev.empty()) HIPCHK(hipEventRecord(c->ev[0], c->stream));
            st = launch_solve(c, so, ticks);
            if (st != VIL_OK) return st;
            if (c->stamps && !c->ev.empty()) HIPCHK(hipEventRecord(c->ev[1], c->stream));
            launched = 1; it = o->max_iterations + 9;
        }
        if (launched == 0) {
            for (int q = 0; q < chunk && it <= o->max_iterations + 8; ++q, ++it, ++launched) {
                if (c->profiling) HIPCHK(hipEventRecord(c->ev[2 * q], c->stream));
                if (c->fused) {
                    // (one launch: the events bracket the whole iteration; what the sweep roles took inside it comes from the launch's own clock stamps, below)
                    st = launch_iter(c, so);
                    if (st != VIL_OK) return st;
                    continue;
                }
                launch_sweep(c, so);
                if (c->profiling) HIPCHK(hipEventRecord(c->ev[2 * q + 1], c->stream));
                st = launch_reduce_step(c, so, true, c->profiling ? c->ev_mid[q] : nullptr, c->profiling ? c->ev_coll[q] : nullptr);
                if (st != VIL_OK) return st;          // a failed collective fails on every rank (all_reduce): nobody is left waiting
            }
            if (c->profiling) HIPCHK(hipEventRecord(c->ev[2 * launched], c->stream));
            hipLaunchKernelGGL(k_finish, dim3(1), dim3(VIL_SWEEP_THREADS), 0, c->stream, view(c, 0), -1);      // the write-out of a solve that ended in the chunk's last iteration
        }
        // solve_finish leaves Ctl, the final state and then the solve generation in pinned host memory: poll that word instead of
        // synchronising (the no-op tail of the chunk is not waited for, nothing is copied afterwards).  A chunk that runs out without
        // finishing is seen by hipStreamQuery and takes the copy + synchronise route, as do profiling and multi-rank solves.
        bool polled = false;
        if (c->d_hseq && !c->profiling && !(c->split && c->lcomm != nullptr)) {
            volatile int* seq = (volatile int*)(c->h_mirror + sizeof(Ctl));
            const int gen = c->solve_gen;
            const auto tp0 = std::chrono::steady_clock::now();
            for (long spin = 1;; ++spin) {
                if (*seq == gen) { polled = true; break; }
                if ((spin & 0x3ff) == 0) {
                    const hipError_t q = hipStreamQuery(c->stream);
                    if (q == hipSuccess) { polled = *seq == gen; break; }
                    if (q != hipErrorNotReady) return VIL_ERR_DEVICE;
                    if (std::chrono::steady_clock::now() - tp0 > std::chrono::seconds(2)) break;
                }
            }
            if (polled) { std::atomic_thread_fence(std::memory_order_acquire); memcpy(c->h_ctl, c->h_mirror, sizeof(Ctl)); polled_done = true; }
        }
        if (!polled) {
            HIPCHK(hipMemcpyAsync(c->h_ctl, c->P.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
        }
        finished = c->h_ctl->done != 0;
        if (c->profiling) {
            int live = c->h_ctl->n_sweeps - sweeps_before;   // launches that found done == 0
            live = std::max(0, std::min(live, launched));
            for (int q = 0; q < live; ++q) {
                float ms = 0.f;
                if (c->fused) {      // one launch: the events give its whole duration (to step_ms; the sweep phase's share moves to sweep_ms from the launch's own stamps, below)
                    HIPCHK(hipEventElapsedTime(&ms, c->ev[2 * q], c->ev[2 * q + 2])); c->prof.step_ms += ms; c->prof.sweep_launches++; c->prof.step_launches++;
                    continue;
                }
                HIPCHK(hipEventElapsedTime(&ms, c->ev[2 * q], c->ev[2 * q + 1])); c->prof.sweep_ms += ms; c->prof.sweep_launches++;
                HIPCHK(hipEventElapsedTime(&ms, c->ev[2 * q + 1], c->ev[2 * q + 2])); c->prof.step_ms += ms; c->prof.step_launches++;
                HIPCHK(hipEventElapsedTime(&ms, c->ev[2 * q + 1], c->ev_mid[q])); c->prof.reduce_ms += ms;
                HIPCHK(hipEventElapsedTime(&ms, c->ev_mid[q], c->ev_coll[q])); c->prof.collective_ms += ms;
            }
        }
        // (sharded solves ignore the host-timed cap: the ranks' clocks disagree, and a rank that stops enqueuing chunks leaves its peers waiting in
        //  the collective of the next iteration -- the iteration cap is the bound every rank applies identically)
        if (!finished && !c->split && o->max_time_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= o->max_time_s) {
            // ceres max_solver_time_in_seconds (estimator.cpp:1411): the host ends the solve; the accepted state is written out on request
            hipLaunchKernelGGL(k_finish, dim3(1), dim3(VIL_SWEEP_THREADS), 0, c->stream, c->P, (int)VIL_TERM_MAX_TIME);
            HIPCHK(hipMemcpyAsync(c->h_ctl, c->P.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            finished = true; polled_done = false;
        }
    }
    HIPCHK(hipGetLastError());
    const Ctl& ctl = *c->h_ctl;
    // a wait inside a launch gave up (vil_math.hpp, spin_until_eq): the master ended the solve with status -2 -- or never ran, and the launches drained without a `done`
    if (!finished || ctl.status == VIL_ERR_DEVICE) { *gave_up = true; return VIL_ERR_DEVICE; }
    c->solves_since_upload++;
    c->last_live = ctl.n_sweeps;
    c->recent_live[c->recent_at++ & 7] = ctl.n_sweeps;
    if ((c->profiling || (c->stamps && persist)) && c->fused && c->d_prof && ctl.n_sweeps <= 64) {
        // the launches' own clock stamps (100 MHz): the sweep phase of a one-launch iteration = first workgroup started -> last sweep role posted
        std::vector<unsigned long long>& hp = c->last_stamps; hp.assign((size_t)64 * VIL_PROF_SLOTS, 0ull);
        HIPCHK(hipMemcpyAsync(hp.data(), c->d_prof, 8 * hp.size(), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (persist && c->stamps && !c->ev.empty()) {      // the resident launch as a whole (HIP events on the library's stream): step_ms / step_launches; its iterations: sweep_launches
            float ms = 0.f;
            HIPCHK(hipEventSynchronize(c->ev[1]));
            HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
            c->prof.step_ms += ms; c->prof.step_launches++; c->prof.sweep_launches += ctl.n_sweeps;
        }
        for (int q = 0; q < ctl.n_sweeps; ++q) {
            const unsigned long long* r = hp.data() + (size_t)q * VIL_PROF_SLOTS;
            if (!r[0] || !r[1]) continue;
            const unsigned long long t0 = ~r[0];
            // the phase table describes a FULL iteration (linearisation, dense solve, step): the last launch of a solve judges a candidate and ends -- averaged in,
            // as until round 5, it pulled every late stamp ~10 % towards zero ("master done 50.3 us" was 55.7)
            if (r[10] >= t0 && r[12] >= t0) {
                for (int k = 1; k < VIL_PROF_SLOTS; ++k) if (r[k] >= t0) c->phase_us[k] += (double)(r[k] - t0) * 0.01;
                c->phase_n++;
                if (q + 1 < ctl.n_sweeps && r[VIL_PROF_SLOTS]) { c->phase_us[0] += (double)(~r[VIL_PROF_SLOTS] - t0) * 0.01; c->period_n++; }      // (persistent solve: first role of this iteration -> first role of the next)
            }
            const double sweep_us = (double)(r[1] - t0) * 0.01, gather_us = r[5] > r[1] ? (double)(r[5] - r[1]) * 0.01 : 0.0;
            c->prof.sweep_ms += sweep_us * 1e-3; c->prof.step_ms -= sweep_us * 1e-3; c->prof.reduce_ms += gather_us * 1e-3;      // (the events gave the whole launch to step_ms)
        }
    }
    memset(sum, 0, sizeof *sum);
    sum->iterations = ctl.iter; sum->successful_steps = ctl.nsucc; sum->termination = ctl.term;
    sum->initial_cost = ctl.initial_cost; sum->final_cost = ctl.cost_cur;
    for (int i = 0; i < VIL_MAX_TRACE; ++i) { sum->cost_trace[i] = ctl.cost_trace[i]; sum->radius_trace[i] = ctl.radius_trace[i]; }
    sum->t_solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // (the accepted state is x[0] = x[1] on the device, gauge-fixed when asked for: solve_finish ran in stream order; whoever reads the
    //  window afterwards orders itself behind it on the stream)
    c->mirror_state = polled_done && c->d_hstate != nullptr;
    if (ctl.status != 0) return ctl.status;
    if (!std::isfinite(ctl.cost_cur)) return VIL_ERR_NON_FINITE;
    return VIL_OK;
}

int vil_solve_resident(vil_ctx* c, const vil_options* o, vil_summary* sum) {
    if (!c || !o || !sum || !c->uploaded || c->resident_kind != 1) return VIL_ERR_INVALID_ARGUMENT;   // vil_marginalize / vil_eval_factors / vil_linearize replaced the window
    if (o->precision != 0 && o->precision != 1) return VIL_ERR_UNSUPPORTED;
    HIPCHK(hipSetDevice(c->device));
    const auto t0 = std::chrono::steady_clock::now();
    // the step kernel's master, helpers and (merged launch) tile workgroups wait for one another inside a launch: like the other persistent kernels of
    // the library a solve holds its device's gate until its result has arrived, so that it never shares the device with a half-resident
    // k_pose_solve / k_vgicp_align of another thread (vil_coop.hpp).  Not taken by the ranks of a communicator (in-process, peer buffers, RCCL): they wait
    // for EACH OTHER's launches inside the per-iteration collective -- two ranks driven from threads of one process would deadlock on it.
    // (the solves of a vil_solve_batch share the gate: one launch per iteration each, never the persistent solve -- the batch has checked that their waiting workgroups fit together)
    std::unique_lock<std::shared_mutex> coop_lock(vilcoop::gate(c->device), std::defer_lock);
    std::shared_lock<std::shared_mutex> coop_shared(vilcoop::gate(c->device), std::defer_lock);
    if (!c->split) { if (c->in_batch) coop_shared.lock(); else coop_lock.lock(); }
    if ((c->P.gauge_on != 0) != c->gauge_on) {            // vil_set_gauge_fix since the upload: the captured graphs carry the old flag
        c->P.gauge_on = c->gauge_on ? 1 : 0;
        for (auto& g : c->graphs) hipGraphExecDestroy(g.exec);
        c->graphs.clear();
    }
    // vil_debug_drop_flag: armed for THIS solve only (its launches carry the hook in their parameter block -- launched directly, the captured graphs do not know it)
    const bool hook = c->drop_role != -1;
    const bool hook_sticky = hook && (c->drop_launch & 0x10000) != 0;      // (... and for the retry as well: the test of a solve that fails on both structures)
    c->P.drop_role = c->drop_role; c->P.drop_launch = c->drop_launch & 0xffff;
    c->drop_role = -1; c->drop_launch = -1;
    // The ladder of launch structures: the persistent solve (EVERY role resident at once), one launch per iteration (the handful of waiting workgroups resident at
    // once), two launches per iteration.  A wait inside a launch that gives up (vil_math.hpp: 50 ms of the device clock) moves the SAME solve one rung down, from the
    // state it started from -- co-residency is a property of the whole device (another process's persistent kernel, a CU mask the occupancy query does not see) and the
    // library's gate is process-local, so this is the way out instead of a hang; the caller (optimization(), estimator.cpp:1400-1414) has no retry of its own.
    // A rung that gives up in two consecutive solves is left out for the next 256 solves of the context (a masked device must not pay the wait per image).
    const bool cfg_persist = c->persist, cfg_fused = c->fused;
    auto restore = [&]() -> int {
        HIPCHK(hipStreamSynchronize(c->stream));      // the drained launches of the attempt (every wait returns at once behind the abort word)
        hipLaunchKernelGGL(k_state_reset, dim3((c->NS + 255) / 256), dim3(256), 0, c->stream, c->P.x[0], c->P.x[1], (const double*)c->d_xsave, c->NS);
        c->reset_pending = false; c->mirror_state = false;
        return VIL_OK;
    };
    for (int q = 0; q < 2; ++q) if (c->rung_cooldown[q] > 0) c->rung_cooldown[q]--;
    const bool can_persist = cfg_persist && cfg_fused && !c->split && !c->profiling && !c->in_batch, can_fused = cfg_fused && !c->split;
    int first = can_persist && c->rung_cooldown[0] == 0 ? 0 : (can_fused && c->rung_cooldown[1] == 0 ? 1 : 2);
    if (!can_fused) first = 2;
    bool gave_up = false, retried = false;
    int st = VIL_ERR_DEVICE;
    for (int rung = first; rung <= 2; ++rung) {
        if (rung == 1 && !can_fused) continue;
        c->persist = rung == 0; c->fused = cfg_fused && rung <= 1;
        st = solve_attempt(c, o, sum, t0, hook || retried, &gave_up);
        c->persist = cfg_persist; c->fused = cfg_fused;
        if (!hook_sticky) { c->P.drop_role = -1; c->P.drop_launch = -1; }
        if (!gave_up) {
            if (rung < 2) c->rung_fail_run[rung] = 0;
            if (retried) c->n_recovered++;
            c->P.drop_role = -1; c->P.drop_launch = -1;
            return st;
        }
        if (rung < 2 && ++c->rung_fail_run[rung] >= 2) { c->rung_cooldown[rung] = 256; c->rung_fail_run[rung] = 0; }
        const int rs = restore();
        if (rs != VIL_OK) return rs;
        retried = true;
    }
    c->P.drop_role = -1; c->P.drop_launch = -1;
    c->n_aborted++;
    HIPCHK(hipStreamSynchronize(c->stream));
    return VIL_ERR_DEVICE;                             // the resident state is the one the solve started from
}

// B windows solved CONCURRENTLY on one device: every context on its own stream, one launch per trust-region iteration each (k_iter), driven by a host thread per
// context.  A single window leaves most of the device idle behind its sweep phase (one master workgroup on dependent fp64 chains); B of them interleave -- the
// single-GPU form of the `replicas` leg of a multi-GPU run.  Each window's result is bit-equal to its solo solve with one launch per iteration.
int vil_solve_batch(vil_ctx** ctxs, int32_t n, const vil_options* o, vil_summary* sums, int32_t* statuses) {
    if (!ctxs || n < 1 || n > 16 || !o || !sums || !statuses) return VIL_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || !ctxs[i]->uploaded || ctxs[i]->resident_kind != 1 || ctxs[i]->split || ctxs[i]->device != ctxs[0]->device) return VIL_ERR_INVALID_ARGUMENT;
        for (int j = 0; j < i; ++j) if (ctxs[j] == ctxs[i]) return VIL_ERR_INVALID_ARGUMENT;
    }
    // how many of them may wait inside their launches at once: the waiting workgroups of all (master, helpers, tiles, chain) must fit the device per XCD
    int group = n;
    {
        int waiting = 0, cap = 1 << 30;
        for (int i = 0; i < n; ++i) { const vil_ctx* c = ctxs[i]; waiting = std::max(waiting, 2 + c->P.n_help + c->n_ww); const int v = c->P.vis_ts == 2 ? 0 : 1; if (c->fused && c->cap_iter[v] > 0) cap = std::min(cap, c->cap_iter[v]); }
        while (group > 1 && !fits_per_xcd(group * waiting, cap)) --group;
        // ... and at most half of the device may wait: the workgroups the waiting ones wait for need the other half.  How many launches really run at once is the
        // runtime's number of hardware queues (four unless GPU_MAX_HW_QUEUES says otherwise): with eight queues eight windows of configs[1] waited on 208 of 256
        // compute units and the batch took 14 ms instead of 2 (profiles/r06_concurrent_windows_8queues.json, before this rule)
        {
            const char* qe = getenv("GPU_MAX_HW_QUEUES");
            const int hwq = qe && atoi(qe) > 0 ? atoi(qe) : 4;
            while (group > 1 && std::min(group, hwq) * waiting > cap / 2) --group;
        }
    }
    for (int i0 = 0; i0 < n; i0 += group) {
        const int m = std::min(group, n - i0);
        std::vector<std::thread> th;
        for (int k = 0; k < m; ++k) {
            vil_ctx* c = ctxs[i0 + k];
            th.emplace_back([c, o, sums, statuses, i0, k]() { c->in_batch = true; statuses[i0 + k] = vil_solve_resident(c, o, &sums[i0 + k]); c->in_batch = false; });
        }
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < n; ++i) if (statuses[i] != VIL_OK) return statuses[i];
    return VIL_OK;
}

int vil_debug_drop_flag(vil_ctx* c, int32_t role, int32_t launch) {
    if (!c || launch < 0) return VIL_ERR_INVALID_ARGUMENT;
    c->drop_role = role; c->drop_launch = launch;
    return VIL_OK;
}
int vil_recovery_counts(vil_ctx* c, int64_t* recovered, int64_t* failed) {
    if (!c) return VIL_ERR_INVALID_ARGUMENT;
    if (recovered) *recovered = c->n_recovered;
    if (failed) *failed = c->n_aborted;
    return VIL_OK;
}

int vil_download_state(vil_ctx* c, vil_state* s) {
    if (!c || !s || !c->uploaded || s->K != c->K || s->L != c->L) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    flush_reset(c);
    const double* x = nullptr;
    std::vector<double> xb;
    if (c->mirror_state) x = (const double*)(c->h_mirror + sizeof(Ctl) + 64);      // left there by solve_finish: no device operation
    else {
        xb.resize(c->NS);
        HIPCHK(hipMemcpyAsync(xb.data(), c->P.x[0], 8 * (size_t)c->NS, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        x = xb.data();
    }
    for (int i = 0; i < c->NS; ++i) if (!std::isfinite(x[i])) return VIL_ERR_NON_FINITE;
    const int K = c->K, L = c->L;
    memcpy(s->pose, &x[0], 8 * (size_t)7 * K); memcpy(s->speedbias, &x[7 * K], 8 * (size_t)9 * K);
    memcpy(s->ex_pose, &x[16 * K], 56); s->td[0] = x[16 * K + 7];
    if (L) memcpy(s->inv_depth, &x[16 * K + 8], 8 * (size_t)L);
    return VIL_OK;
}

int vil_solve(vil_ctx* c, const vil_problem* p, vil_state* s, const vil_options* o, vil_summary* sum) {
    if (!c || !p || !s || !o || !sum) return VIL_ERR_INVALID_ARGUMENT;
    const auto t0 = std::chrono::steady_clock::now();
    int st = upload_window(c, p, s, false);            // nothing is waited for: the solve's launches queue up behind the upload
    if (st != VIL_OK) return st;
    const auto t1 = std::chrono::steady_clock::now();
    st = vil_solve_resident(c, o, sum);
    sum->t_prepare_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (st != VIL_OK) return st;                       // state left unchanged on any error
    const auto t2 = std::chrono::steady_clock::now();
    st = vil_download_state(c, s);
    sum->t_readback_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count();
    return st;
}

int vil_solve_device_lidar(vil_ctx* c, const vil_problem* p, const vil_device_lidar* dl, void* producer_stream, vil_state* s, const vil_options* o, vil_summary* sum) {
    if (!c || !p || !dl || !s || !o || !sum) return VIL_ERR_INVALID_ARGUMENT;
    if ((p->n_plane > 0 && (!dl->plane_soa || dl->plane_stride < p->n_plane)) || (p->n_edge > 0 && (!dl->edge_soa || dl->edge_stride < p->n_edge))) return VIL_ERR_INVALID_ARGUMENT;
    if (c->world > 1 && c->has_comm()) return VIL_ERR_UNSUPPORTED;
    const auto t0 = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(c->device));
    if (producer_stream && (hipStream_t)producer_stream != c->stream) {      // the tables are written by work on another stream: order after it on the device
        if (!c->dep_ev) HIPCHK(hipEventCreateWithFlags(&c->dep_ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->dep_ev, (hipStream_t)producer_stream));
        HIPCHK(hipStreamWaitEvent(c->stream, c->dep_ev, 0));
    }
    int st = upload_impl(c, p, s, false, dl);
    if (st != VIL_OK) return st;
    c->resident_kind = 1;
    const auto t1 = std::chrono::steady_clock::now();
    st = vil_solve_resident(c, o, sum);
    sum->t_prepare_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (st != VIL_OK) return st;
    const auto t2 = std::chrono::steady_clock::now();
    st = vil_download_state(c, s);
    sum->t_readback_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count();
    return st;
}

static int ensure_pin(vil_ctx* c, size_t bytes) {
    if (bytes <= c->h_pin_bytes) return VIL_OK;
    if (c->h_pin) hipHostFree(c->h_pin);
    c->h_pin = nullptr; c->h_pin_bytes = 0;
    HIPCHK(hipHostMalloc(&c->h_pin, bytes, hipHostMallocDefault));
    c->h_pin_bytes = bytes;
    return VIL_OK;
}

int vil_eval_factors(vil_ctx* c, const vil_problem* p, const vil_state* s, int cls, double* r, double* J) {
    if (!c || !r) return VIL_ERR_INVALID_ARGUMENT;
    int st = upload_impl(c, p, s, false);
    if (st != VIL_OK) return st;
    const DevP& P = c->P;
    size_t nr = 0, nj = 0; int nfac = 0;
    switch (cls) {
        case VIL_FACTOR_IMU: nfac = P.n_imu; nr = 15; nj = 480; break;
        case VIL_FACTOR_VISUAL: nfac = P.n_vis; nr = 2; nj = 46; break;
        case VIL_FACTOR_ICP: nfac = P.n_icp; nr = 3; nj = 84; break;
        case VIL_FACTOR_LPS: nfac = P.n_lps; nr = 3; nj = 42; break;
        case VIL_FACTOR_EDGE: nfac = P.n_edge; nr = 3; nj = 21; break;
        case VIL_FACTOR_PLANE: nfac = P.n_plane; nr = 1; nj = 7; break;
        case VIL_FACTOR_PRIOR: nfac = P.pn ? 1 : 0; nr = P.pn; nj = 0; for (int b = 0; b < P.pnblk; ++b) { int k = p->prior.blk_kind[b]; nj += (size_t)P.pn * ((k == 0 || k == 2) ? 7 : (k == 1 ? 9 : 1)); } break;
        default: return VIL_ERR_INVALID_ARGUMENT;
    }
    if (nfac == 0) return VIL_OK;
    const size_t br = 8 * nr * nfac, bj = J ? 8 * nj * nfac : 0;
    double *d_r = nullptr, *d_J = nullptr; int* d_joff = nullptr;
    struct Free { double*& r; double*& J; int*& o; ~Free() { if (r) hipFree(r); if (J) hipFree(J); if (o) hipFree(o); } } free_tmp{d_r, d_J, d_joff};   // also on the error returns
    HIPCHK(hipMalloc(&d_r, br));
    if (J) HIPCHK(hipMalloc(&d_J, bj));
    const double* x = P.x[0];
    const int nb = (nfac + VIL_THREADS - 1) / VIL_THREADS;
    switch (cls) {
        case VIL_FACTOR_IMU: hipLaunchKernelGGL(k_eval_imu, dim3(nfac), dim3(VIL_THREADS), 0, c->stream, P, x, d_r, d_J); break;
        case VIL_FACTOR_VISUAL: hipLaunchKernelGGL(k_eval_visual, dim3(nb), dim3(VIL_THREADS), 0, c->stream, P, x, d_r, d_J); break;
        case VIL_FACTOR_ICP: hipLaunchKernelGGL(k_eval_rel, dim3(2), dim3(VIL_THREADS), 0, c->stream, P, x, 1, d_r, d_J); break;
        case VIL_FACTOR_LPS: hipLaunchKernelGGL(k_eval_rel, dim3(2), dim3(VIL_THREADS), 0, c->stream, P, x, 0, d_r, d_J); break;
        case VIL_FACTOR_EDGE: hipLaunchKernelGGL(k_eval_lidar<3>, dim3(nb), dim3(VIL_THREADS), 0, c->stream, P, x, d_r, d_J); break;
        case VIL_FACTOR_PLANE: hipLaunchKernelGGL(k_eval_lidar<1>, dim3(nb), dim3(VIL_THREADS), 0, c->stream, P, x, d_r, d_J); break;
        case VIL_FACTOR_PRIOR:
            HIPCHK(hipMalloc(&d_joff, 4 * c->prior_joff.size()));
            HIPCHK(hipMemcpyAsync(d_joff, c->prior_joff.data(), 4 * c->prior_joff.size(), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(k_eval_prior, dim3(1), dim3(VIL_THREADS), 8 * (size_t)P.pn, c->stream, P, x, d_r, d_J, d_joff);
            break;
    }
    st = ensure_pin(c, br + bj);
    if (st != VIL_OK) return st;
    HIPCHK(hipMemcpyAsync(c->h_pin, d_r, br, hipMemcpyDeviceToHost, c->stream));
    if (J) HIPCHK(hipMemcpyAsync((char*)c->h_pin + br, d_J, bj, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    const double* hr = c->h_pin; const double* hj = (const double*)((char*)c->h_pin + br);
    if (cls == VIL_FACTOR_EDGE || cls == VIL_FACTOR_PLANE) {   // undo the pose sort
        const std::vector<int>& perm = cls == VIL_FACTOR_EDGE ? c->edge_perm : c->plane_perm;
        for (int sidx = 0; sidx < nfac; ++sidx) {
            const int f = perm[sidx];
            memcpy(r + (size_t)f * nr, hr + (size_t)sidx * nr, 8 * nr);
            if (J) memcpy(J + (size_t)f * nj, hj + (size_t)sidx * nj, 8 * nj);
        }
    } else { memcpy(r, hr, br); if (J) memcpy(J, hj, bj); }
    return VIL_OK;
}

int vil_eval_lidar_functors(vil_ctx* c, int32_t kind, int32_t n, const double* consts, const double* q_lb, const double* t_lb, const double* pose7, double* r, double* J) {
    if (!c || kind < VIL_LIDAR_EDGE || kind > VIL_LIDAR_DISTANCE || n < 0 || (n > 0 && !consts) || !q_lb || !t_lb || !pose7 || !r) return VIL_ERR_INVALID_ARGUMENT;
    if (n == 0) return VIL_OK;
    HIPCHK(hipSetDevice(c->device));
    const int nc = kind == VIL_LIDAR_EDGE ? 9 : (kind == VIL_LIDAR_PLANE3 ? 12 : (kind == VIL_LIDAR_PLANE_NORM ? 7 : 6)), nr = (kind == VIL_LIDAR_EDGE || kind == VIL_LIDAR_DISTANCE) ? 3 : 1;
    LidarFunctorArgs A;
    { double R[9]; quat_to_R_host(q_lb, R); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A.Rbl[3 * i + j] = R[3 * j + i];          // body <- LiDAR: (R_lb^T, -R_lb^T t_lb), as the window upload
      for (int i = 0; i < 3; ++i) A.tbl[i] = -(A.Rbl[3 * i] * t_lb[0] + A.Rbl[3 * i + 1] * t_lb[1] + A.Rbl[3 * i + 2] * t_lb[2]); }
    memcpy(A.pose, pose7, sizeof A.pose);
    const size_t bc = 8 * (size_t)nc * n, br = 8 * (size_t)nr * n, bj = J ? 8 * (size_t)nr * 7 * n : 0;
    double *d_c = nullptr, *d_r = nullptr, *d_J = nullptr;
    struct Free { double*& a; double*& b; double*& c; ~Free() { if (a) hipFree(a); if (b) hipFree(b); if (c) hipFree(c); } } free_tmp{d_c, d_r, d_J};
    HIPCHK(hipMalloc(&d_c, bc)); HIPCHK(hipMalloc(&d_r, br));
    if (J) HIPCHK(hipMalloc(&d_J, bj));
    HIPCHK(hipMemcpyAsync(d_c, consts, bc, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_eval_lidar_functors, dim3((n + VIL_THREADS - 1) / VIL_THREADS), dim3(VIL_THREADS), 0, c->stream, (int)kind, (int)n, d_c, A, d_r, d_J);
    HIPCHK(hipMemcpyAsync(r, d_r, br, hipMemcpyDeviceToHost, c->stream));
    if (J) HIPCHK(hipMemcpyAsync(J, d_J, bj, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    return VIL_OK;
}

int vil_linearize(vil_ctx* c, const vil_problem* p, const vil_state* s, const vil_options* o, double* cost, double* S, double* g) {
    if (!c || !o || !cost || !S || !g) return VIL_ERR_INVALID_ARGUMENT;
    int st = vil_upload(c, p, s);
    if (st != VIL_OK) return st;
    const SolveOpts so = to_dev_opts(o);
    st = init_ctl(c, o, 1);
    if (st != VIL_OK) return st;
    launch_sweep(c, so);                                   // partial records of system set 1 (cand = 1 - cur)
    launch_reduce_step(c, so, false);
    const size_t D = c->D;
    st = ensure_pin(c, 8 * (D * D + D + 1));
    if (st != VIL_OK) return st;
    const double* src = c->P.sys[1].ar;        // un-sharded: the candidate set (cur = 0); sharded: the all-reduced set
    HIPCHK(hipMemcpyAsync(c->h_pin, src, 8 * D * D, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_pin + D * D, src + D * D, 8 * D, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(c->h_pin + D * D + D, src + D * D + 3 * D, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    memcpy(S, c->h_pin, 8 * D * D); memcpy(g, c->h_pin + D * D, 8 * D); *cost = c->h_pin[D * D + D];
    return VIL_OK;
}

// common tail of vil_marginalize / vil_marginalize_resident: the normal equations of the collected factors sit in sys[1] (or the
// all-reduce staging buffer); Schur complement + square root on the device (k_marg), block metadata with the address shift as an
// index remap (estimator.cpp:1599-1611, 1654-1677)
static int marg_finish(vil_ctx* c, const int K, const bool old_, const int drop_pose, const std::vector<char>& pose_t, const std::vector<char>& sb_t,
                       const bool ex_t, const bool td_t, const bool use_td, const int n_lm_elim, const vil_state* s, vil_prior_out* out,
                       const bool to_slot = false, vil_win_prior_info* winfo = nullptr) {
    int st = VIL_OK;
    // ---- which reduced columns are dropped / kept (canonical kept order: poses, speed-biases, ex, td) ----------------
    const int D = c->D;
    std::vector<int> drop_cols, keep_cols, kinds, index;
    if (pose_t[drop_pose]) for (int k = 0; k < 6; ++k) drop_cols.push_back(6 * drop_pose + k);
    if (old_ && sb_t[0]) for (int k = 0; k < 9; ++k) drop_cols.push_back(6 * K + 7 + k);
    for (int k = 0; k < K; ++k) if (pose_t[k] && k != drop_pose) { kinds.push_back(VIL_BLK_POSE); index.push_back(k); for (int q2 = 0; q2 < 6; ++q2) keep_cols.push_back(6 * k + q2); }
    for (int k = 0; k < K; ++k) if (sb_t[k] && !(old_ && k == 0)) { kinds.push_back(VIL_BLK_SPEEDBIAS); index.push_back(k); for (int q2 = 0; q2 < 9; ++q2) keep_cols.push_back(6 * K + 7 + 9 * k + q2); }
    if (ex_t) { kinds.push_back(VIL_BLK_EX); index.push_back(0); for (int q2 = 0; q2 < 6; ++q2) keep_cols.push_back(6 * K + q2); }
    if (td_t && use_td) { kinds.push_back(VIL_BLK_TD); index.push_back(0); keep_cols.push_back(6 * K + 6); }
    const int nd = (int)drop_cols.size(), n = (int)keep_cols.size();
    int n_max, nblk_max, x0_max;
    vil_prior_capacity(K, &n_max, &nblk_max, &x0_max);
    if (n > n_max || n > 136 || nd > 15 || (int)kinds.size() > nblk_max || n <= 0 || nd <= 0) return VIL_ERR_UNSUPPORTED;
    // ---- device work space + kernel ------------------------------------------------------------------------------
    const size_t nn = (size_t)n * n;
    const size_t bytes = 8 * ((size_t)nd * nd * 2 + nd + nn * 5 + (size_t)n * 3) + 4 * (size_t)(nd + n) + 8192 + 256;
    if (bytes > c->marg_ws_bytes) {          // grow-only work space: no allocation on the per-frame path once warm
        if (c->marg_ws) hipFree(c->marg_ws);
        c->marg_ws = nullptr; c->marg_ws_bytes = 0;
        HIPCHK(hipMalloc(&c->marg_ws, bytes));
        c->marg_ws_bytes = bytes;
    }
    char* dw = c->marg_ws;
    size_t off = 0;
    auto take = [&](size_t b2) { char* r = dw + off; off += (b2 + 255) & ~size_t(255); return r; };
    MargDev M;
    M.D = D; M.nd = nd; M.n = n; M.eps = 1e-8;
    M.S = c->P.sys[1].S; M.g = M.S + (size_t)D * D;
    int* d_drop = (int*)take(4 * (size_t)(nd + n)); int* d_keep = d_drop + nd;       // one table, one copy
    M.drop_cols = d_drop; M.keep_cols = d_keep;
    M.Add = (double*)take(8 * (size_t)nd * nd); M.Vd = (double*)take(8 * (size_t)nd * nd); M.wd = (double*)take(8 * (size_t)nd);
    M.T = (double*)take(8 * nn);
    M.V = (double*)take(8 * nn); M.w = (double*)take(8 * (size_t)n);
    M.J0 = (double*)take(8 * (2 * nn + 2 * (size_t)n)); M.A = M.J0 + nn; M.r0 = M.A + nn; M.b = M.r0 + n;      // what goes back to the host: one block, one copy
    M.stat = (int*)take(32);
    M.ts = (unsigned long long*)take(8 * 16); c->d_marg_ts = M.ts;
    if (off > bytes) return VIL_ERR_DEVICE;
    // the column table travels through pinned memory: the copy is asynchronous and the resident window never waits for it (the next use
    // of this staging area is a whole solve away)
    st = ensure_pin(c, 8 * (nn * 2 + 2 * (size_t)n + 16 * (size_t)K + 8) + 4 * (size_t)(nd + n) + 64);
    if (st != VIL_OK) return st;
    int* hcols = (int*)((char*)c->h_pin + 8 * (nn * 2 + 2 * (size_t)n + 16 * (size_t)K + 8));
    for (int q = 0; q < nd; ++q) hcols[q] = drop_cols[q];
    for (int q = 0; q < n; ++q) hcols[nd + q] = keep_cols[q];
    HIPCHK(hipMemcpyAsync(d_drop, hcols, 4 * (size_t)(nd + n), hipMemcpyHostToDevice, c->stream));
    {
        const size_t a_bytes = 8 * nn, cap = 156 * 1024;
        if (a_bytes > cap) return VIL_ERR_UNSUPPORTED;
        const size_t dyn = std::max<size_t>(4096, a_bytes);
        if (dyn > 48 * 1024 && (int)dyn > c->attr_marg[0]) { HIPCHK(hipFuncSetAttribute((const void*)k_marg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)); c->attr_marg[0] = (int)dyn; }
        hipLaunchKernelGGL(k_marg, dim3(1), dim3(MARG_THREADS), dyn, c->stream, M, 0);
        if (!VIL_TUNE_ENV("VIL_MARG_PIVOTED")) {               // un-pivoted factorisation on the matrix cores first; pivoted fallback below
            const size_t Tm = (size_t)(n + 1 + 15) / 16, tb = 8 * (size_t)TILE_SZ * (Tm * (Tm + 1) / 2);
            if (tb + sizeof(vd::StepShared) + 512 <= 160 * 1024) {
                if (tb > 48 * 1024 && (int)tb > c->attr_marg[1]) { HIPCHK(hipFuncSetAttribute((const void*)k_marg_fast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb)); c->attr_marg[1] = (int)tb; }
                hipLaunchKernelGGL(k_marg_fast, dim3(1), dim3(VIL_STEP_THREADS), tb, c->stream, M);
            }
        }
        hipLaunchKernelGGL(k_marg, dim3(1), dim3(MARG_THREADS), dyn, c->stream, M, 1);
    }
    if (to_slot) {
        // resident window: the new prior goes device-to-device into the other prior slot (J0, r0, x0 from the device state, contractions);
        // nothing is read back, nothing is waited for.  A non-finite result raises the window's status word.
        auto& w = c->win;
        if (n > w.nmax || (int)kinds.size() > VIL_WIN_MAXBLK) return VIL_ERR_UNSUPPORTED;
        const int dst = 1 - w.cur;
        PriorCommit pc; memset(&pc, 0, sizeof pc);
        pc.n = n; pc.nblk = (int)kinds.size(); pc.J0 = M.J0; pc.r0 = M.r0; pc.x = c->P.x[0]; pc.K = K;
        for (size_t b = 0; b < kinds.size(); ++b) { pc.kind[b] = kinds[b]; pc.index[b] = index[b]; }
        pc.pJ0 = w.pJ0(dst); pc.pr0 = w.pr0(dst); pc.px0 = w.px0(dst); pc.pH = w.pH(dst); pc.pg0 = w.pg0(dst); pc.pc0 = w.pc0(dst); pc.status = w.d_wstat;
        {
            const size_t pl = 8 * ((size_t)n * (n + 1) + n + 8);
            if (pl > 48 * 1024 && (int)pl > c->attr_commit) { HIPCHK(hipFuncSetAttribute((const void*)k_prior_commit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl)); c->attr_commit = (int)pl; }
            hipLaunchKernelGGL(k_prior_commit, dim3(16), dim3(256), pl, c->stream, pc);
        }
        w.cur = dst; w.pn = n; w.pm = nd + n_lm_elim;
        w.pkind.assign(kinds.begin(), kinds.end()); w.pindex.clear(); w.pcol.clear();
        int col = 0;
        for (size_t b = 0; b < kinds.size(); ++b) {
            const int kind = kinds[b], idx = index[b];
            int ni = idx;
            if (kind == VIL_BLK_POSE || kind == VIL_BLK_SPEEDBIAS) ni = old_ ? idx - 1 : (idx == K - 1 ? K - 2 : idx);      // the address shift as an index remap (estimator.cpp:1599-1611, 1654-1677)
            w.pindex.push_back(ni); w.pcol.push_back(col);
            col += (kind == VIL_BLK_POSE || kind == VIL_BLK_EX) ? 6 : (kind == VIL_BLK_SPEEDBIAS ? 9 : 1);
        }
        if (winfo) {
            winfo->n = n; winfo->nblk = (int)kinds.size(); winfo->m = w.pm;
            for (size_t b = 0; b < kinds.size(); ++b) { winfo->blk_kind[b] = w.pkind[b]; winfo->blk_index[b] = w.pindex[b]; winfo->blk_col[b] = w.pcol[b]; }
        }
        return VIL_OK;
    }
    const size_t ncam = 16 * (size_t)K + 8;
    HIPCHK(hipMemcpyAsync(c->h_pin, M.J0, 8 * (2 * nn + 2 * (size_t)n), hipMemcpyDeviceToHost, c->stream));      // J0 | A | r0 | b
    // x0 of the new prior = the state the factors were LINEARISED at, i.e. the device state (the reference stores the very values it
    // marginalises at, marginalization_factor.cpp:110-139) -- not whatever the caller holds
    double* hx = c->h_pin + 2 * nn + 2 * (size_t)n;
    HIPCHK(hipMemcpyAsync(hx, c->P.x[0], 8 * ncam, hipMemcpyDeviceToHost, c->stream));
    int mstat[4] = {0, 0, 0, 0};
    if (VIL_TUNE_ENV("VIL_MARG_DEBUG")) HIPCHK(hipMemcpyAsync(mstat, M.stat, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    if (VIL_TUNE_ENV("VIL_MARG_DEBUG")) fprintf(stderr, "[vil_marginalize] n=%d nd=%d one-sided Jacobi stages: %d (dropped block) %d (n x n, %.1f sweeps)\n", n, nd, mstat[0], mstat[1], mstat[1] / (double)(((n + 1) & ~1) - 1));
    for (size_t e = 0; e < 2 * nn + 2 * (size_t)n; ++e) if (!std::isfinite(c->h_pin[e])) return VIL_ERR_NON_FINITE;
    // ---- getParameterBlocks with the address shift as an index remap (estimator.cpp:1599-1611, 1654-1677) --------------
    out->n = n; out->m = nd + n_lm_elim; out->nblk = (int)kinds.size();
    memcpy(out->J0, c->h_pin, 8 * nn); memcpy(out->r0, c->h_pin + 2 * nn, 8 * (size_t)n);
    if (out->A) memcpy(out->A, c->h_pin + nn, 8 * nn);
    if (out->b) memcpy(out->b, c->h_pin + 2 * nn + n, 8 * (size_t)n);
    int col = 0, xo = 0;
    for (size_t b = 0; b < kinds.size(); ++b) {
        const int kind = kinds[b], idx = index[b];
        out->blk_kind[b] = kind;
        int ni = idx;
        if (kind == VIL_BLK_POSE || kind == VIL_BLK_SPEEDBIAS) ni = old_ ? idx - 1 : (idx == K - 1 ? K - 2 : idx);
        out->blk_index[b] = ni; out->blk_col[b] = col;
        const double* src = kind == VIL_BLK_POSE ? hx + 7 * idx : (kind == VIL_BLK_SPEEDBIAS ? hx + 7 * K + 9 * idx : (kind == VIL_BLK_EX ? hx + 16 * K : hx + 16 * K + 7));
        const int gs = (kind == VIL_BLK_POSE || kind == VIL_BLK_EX) ? 7 : (kind == VIL_BLK_SPEEDBIAS ? 9 : 1), ls = gs == 7 ? 6 : gs;
        for (int k = 0; k < gs; ++k) out->x0[xo + k] = src[k];
        xo += gs; col += ls;
    }
    return VIL_OK;
}

int vil_marginalize(vil_ctx* c, const vil_problem* p, const vil_state* s, const vil_options* o, const vil_marg_spec* spec, vil_prior_out* out) {
    if (!c || !p || !s || !o || !spec || !out) return VIL_ERR_INVALID_ARGUMENT;
    { const int vst = validate(p, s); if (vst != VIL_OK) return vst; }      // the host tables are indexed below, before upload_impl sees them
    const int K = p->K;
    if (K < 3) return VIL_ERR_INVALID_ARGUMENT;
    const bool old_ = spec->flag == VIL_MARGIN_OLD;
    const int drop_pose = old_ ? 0 : K - 2;
    // ---- derived sub-problem: the factors MarginalizationInfo collects (estimator.cpp:1489-1589 / 1626-1641), every block free
    vil_problem q = *p;
    q.pose_const = nullptr; q.sb_const = nullptr; q.lm_const = nullptr; q.ex_const = 0; q.td_const = 0;
    q.n_edge = 0; q.n_plane = 0; q.edge_pose = nullptr; q.plane_pose = nullptr; q.edge_const = nullptr; q.plane_const = nullptr;
    std::vector<int> imu_i, imu_j, vis_i, vis_j, vis_l, icp_ids, lps_ids, edge_pose, plane_pose;
    std::vector<double> imu_c, vis_c, icp_c, lps_c, edge_c, plane_c;
    std::vector<char> pose_t(K, 0), sb_t(K, 0);
    bool ex_t = false, td_t = false;
    int n_lm_elim = 0;
    if (p->prior.n > 0) {
        bool has_drop = false;
        for (int b = 0; b < p->prior.nblk; ++b) {
            const int kind = p->prior.blk_kind[b], idx = p->prior.blk_index[b];
            if (kind == VIL_BLK_POSE) { pose_t[idx] = 1; if (idx == drop_pose) has_drop = true; }
            else if (kind == VIL_BLK_SPEEDBIAS) sb_t[idx] = 1;
            else if (kind == VIL_BLK_EX) ex_t = true; else td_t = true;
        }
        if (!old_ && !has_drop) { out->n = -1; return VIL_OK; }     // estimator.cpp:1620-1621: prior kept as is
    } else if (!old_) { out->n = -1; return VIL_OK; }
    if (old_) {
        for (int f = 0; f < p->n_imu; ++f) if (p->imu_i[f] == 0 && p->imu_j[f] == 1 && p->imu_const[(size_t)f * 287 + 16] < 10.0) {
            imu_i.push_back(0); imu_j.push_back(1); imu_c.insert(imu_c.end(), p->imu_const + (size_t)f * 287, p->imu_const + (size_t)(f + 1) * 287);
            pose_t[0] = pose_t[1] = 1; sb_t[0] = sb_t[1] = 1;
        }
        int last_l = -1;
        for (int f = 0; f < p->n_vis; ++f) if (p->vis_i[f] == 0) {
            vis_i.push_back(0); vis_j.push_back(p->vis_j[f]); vis_l.push_back(p->vis_l[f]);
            vis_c.insert(vis_c.end(), p->vis_const + (size_t)f * 14, p->vis_const + (size_t)(f + 1) * 14);
            pose_t[0] = 1; pose_t[p->vis_j[f]] = 1; ex_t = true; if (p->use_td) td_t = true;
            if (p->vis_l[f] != last_l) { ++n_lm_elim; last_l = p->vis_l[f]; }
        }
        if (spec->icp_marg >= 0 && spec->icp_marg < p->n_icp) {
            for (int b = 0; b < 4; ++b) { const int id = b == 0 ? 0 : p->icp_ids[4 * spec->icp_marg + b]; icp_ids.push_back(id); pose_t[id] = 1; }
            icp_c.insert(icp_c.end(), p->icp_const + (size_t)spec->icp_marg * 10, p->icp_const + (size_t)(spec->icp_marg + 1) * 10);
        }
        if (spec->lps_marg >= 0 && spec->lps_marg < p->n_lps) {
            for (int b = 0; b < 2; ++b) { const int id = b == 0 ? 0 : p->lps_ids[2 * spec->lps_marg + b]; lps_ids.push_back(id); pose_t[id] = 1; }
            lps_c.insert(lps_c.end(), p->lps_const + (size_t)spec->lps_marg * 7, p->lps_const + (size_t)(spec->lps_marg + 1) * 7);
        }
        // LiDAR point factors of the dropped pose (extended mode): MarginalizationInfo folds every factor that touches a
        // dropped block (marginalization_factor.cpp:176-316); these touch pose 0 only and reach the prior through A_mm / b_m
        for (int f = 0; f < p->n_edge; ++f) if (p->edge_pose[f] == 0) { edge_pose.push_back(0); edge_c.insert(edge_c.end(), p->edge_const + (size_t)f * 9, p->edge_const + (size_t)(f + 1) * 9); }
        for (int f = 0; f < p->n_plane; ++f) if (p->plane_pose[f] == 0) { plane_pose.push_back(0); plane_c.insert(plane_c.end(), p->plane_const + (size_t)f * 7, p->plane_const + (size_t)(f + 1) * 7); }
        if (!edge_pose.empty() || !plane_pose.empty()) pose_t[0] = 1;
    }
    q.n_imu = (int)imu_i.size(); q.imu_i = imu_i.data(); q.imu_j = imu_j.data(); q.imu_const = imu_c.data();
    q.n_vis = (int)vis_i.size(); q.vis_i = vis_i.data(); q.vis_j = vis_j.data(); q.vis_l = vis_l.data(); q.vis_const = vis_c.data();
    q.n_icp = (int)icp_ids.size() / 4; q.icp_ids = icp_ids.data(); q.icp_const = icp_c.data();
    q.n_lps = (int)lps_ids.size() / 2; q.lps_ids = lps_ids.data(); q.lps_const = lps_c.data();
    q.n_edge = (int)edge_pose.size(); q.edge_pose = edge_pose.data(); q.edge_const = edge_c.data();
    q.n_plane = (int)plane_pose.size(); q.plane_pose = plane_pose.data(); q.plane_const = plane_c.data();
    // under a communicator the collected factors are sharded like a solve's (visual by landmark owner, LiDAR points in slices, the rest on
    // rank 0), A and b are all-reduced ONCE and every rank finishes the small dense part redundantly -- the reference's own pattern
    // (marginalization_factor.cpp:235-264: factors dealt to four threads, private A / b, summed after the join)
    int st = (c->world > 1 && c->has_comm()) ? upload_sharded(c, &q, s) : upload_impl(c, &q, s, false);
    if (st != VIL_OK) return st;
    const SolveOpts so = to_dev_opts(o);
    st = init_ctl(c, o, 2);
    if (st != VIL_OK) return st;
    launch_sweep(c, so);
    st = launch_reduce_step(c, so, false);
    if (st != VIL_OK) return st;
    c->resident_kind = 2;
    return marg_finish(c, K, old_, drop_pose, pose_t, sb_t, ex_t, td_t, p->use_td != 0, n_lm_elim, s, out);
}

int vil_gauge_fix(const double* pose0_before, vil_state* s) {
    if (!pose0_before || !s) return VIL_ERR_INVALID_ARGUMENT;
    gauge_fix_core(pose0_before, s->K, s->pose, s->speedbias, s->ex_pose);
    return VIL_OK;
}

// visual factors by landmark owner (contiguous landmark ranges balanced by factor count),
// LiDAR points in contiguous equal chunks.  Pure host logic, no device needed.
int vil_shard_ranges(const vil_problem* p, int rank, int world, int32_t* lm_begin, int32_t* lm_end, int32_t* edge_begin, int32_t* edge_end, int32_t* plane_begin, int32_t* plane_end) {
    if (!p || world <= 0 || rank < 0 || rank >= world) return VIL_ERR_INVALID_ARGUMENT;
    std::vector<int> lms(p->L + 1, 0);
    for (int f = 0; f < p->n_vis; ++f) lms[p->vis_l[f] + 1]++;
    for (int l = 0; l < p->L; ++l) lms[l + 1] += lms[l];
    if (lm_begin) *lm_begin = shard_cut(lms, p->L, rank, world);
    if (lm_end) *lm_end = shard_cut(lms, p->L, rank + 1, world);
    if (edge_begin) *edge_begin = (int)((long long)p->n_edge * rank / world);
    if (edge_end) *edge_end = (int)((long long)p->n_edge * (rank + 1) / world);
    if (plane_begin) *plane_begin = (int)((long long)p->n_plane * rank / world);
    if (plane_end) *plane_end = (int)((long long)p->n_plane * (rank + 1) / world);
    return VIL_OK;
}

// bytes this rank sends to EACH peer per trust-region iteration through the library's own exchanges (peer buffers, in-process communicator): the lower
// triangle of S' + the vectors + its own slice of the landmark arrays.  (RCCL all-reduces the whole set: vil_comm_init.)
int vil_comm_message_bytes(vil_ctx* c, int64_t* bytes_per_peer, int64_t* bytes_full_set) {
    if (!c || !c->uploaded || !c->split) return VIL_ERR_INVALID_ARGUMENT;
    const OwnSeg& S = c->own;
    const size_t D = (size_t)c->D, cam_sent = S.cam - D * (D - 1) / 2;
    size_t own = 0;
    if (S.n > 0) own = (size_t)17 * (S.lb[c->rank + 1] - S.lb[c->rank]) + (size_t)6 * (S.fb[c->rank + 1] - S.fb[c->rank]);
    else own = c->span - S.cam;
    if (c->slim()) { const SlimLay Y = slim_layout(c); if (bytes_per_peer) *bytes_per_peer = (int64_t)(8 * (Y.camS + Y.gmax)); }      // RCCL: M all-reduced + this rank's padded slice all-gathered
    else if (c->comm && c->world > 1) { if (bytes_per_peer) *bytes_per_peer = (int64_t)(8 * c->span); }      // RCCL beyond eight ranks: the whole set is all-reduced
    else if (bytes_per_peer) *bytes_per_peer = (int64_t)(8 * (cam_sent + own));
    if (bytes_full_set) *bytes_full_set = (int64_t)(8 * c->span);
    return VIL_OK;
}

// ---- window residency across frames (include/vilsolve.h) ------------------------------------------------------------------------
int vil_debug_fail_graph_capture(vil_ctx* c, int32_t n) { if (!c || n < 0) return VIL_ERR_INVALID_ARGUMENT; c->fail_capture = n; c->graph_failed = false; return VIL_OK; }
int vil_debug_set_launch_mode(vil_ctx* c, int32_t mode) { if (!c || mode < 0 || mode > 4) return VIL_ERR_INVALID_ARGUMENT; c->launch_mode = mode; c->uploaded = false; c->resident_kind = 0; return VIL_OK; }
int vil_comm_info(vil_ctx* c, int32_t* rank, int32_t* world, int32_t* transport) {
    if (!c) return VIL_ERR_INVALID_ARGUMENT;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (transport) *transport = c->comm ? 1 : (c->lcomm ? 2 : (c->ipc ? (c->ipc->ready ? 3 : -3) : 0));
    return VIL_OK;
}
int vil_debug_dense_solve(vil_ctx* c, int32_t D, const double* A, double* L, double* x, int32_t* ok, int32_t variant) {
    if (!c || !A || !L || !x || !ok || D < 1 || D > 159 || variant < 0 || variant > 1) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    const int R = D + 1, T = (R + 15) / 16, nt = T * (T + 1) / 2;
    const size_t lds = 8 * (size_t)TILE_SZ * nt, nb = 8 * (size_t)R * R;
    double *dA = nullptr, *dL = nullptr, *dx = nullptr; int* dok = nullptr;
    HIPCHK(hipMalloc((void**)&dA, nb)); HIPCHK(hipMalloc((void**)&dL, nb)); HIPCHK(hipMalloc((void**)&dx, 8 * (size_t)R)); HIPCHK(hipMalloc((void**)&dok, 64));
    HIPCHK(hipMemcpy(dA, A, nb, hipMemcpyHostToDevice));
    const bool small = nt <= 18;                         // (three register tiles per tile wave are enough -- what the step takes at K <= 10)
    const void* fn = variant ? (small ? (const void*)k_debug_dense<3, true> : (const void*)k_debug_dense<CH_SLOTS, true>) : (small ? (const void*)k_debug_dense<3, false> : (const void*)k_debug_dense<CH_SLOTS, false>);
    HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* args[] = {(void*)&dA, (void*)&dL, (void*)&dx, (void*)&dok, (void*)&D};
    HIPCHK(hipLaunchKernel(fn, dim3(1), dim3(VIL_STEP_THREADS), args, lds, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int hok = 0;
    HIPCHK(hipMemcpy(L, dL, nb, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(x, dx, 8 * (size_t)D, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(&hok, dok, 4, hipMemcpyDeviceToHost));
    *ok = hok;
    hipFree(dA); hipFree(dL); hipFree(dx); hipFree(dok);
    return VIL_OK;
}
int vil_profile_workgroups(vil_ctx* c, int32_t launch, uint64_t* times, int32_t max_workgroups) {
    if (!c) return VIL_ERR_INVALID_ARGUMENT;
    if (!times) {                                         // arm: the solves from now on record launch `launch` (< 0: none)
        c->wg_launch = launch;
        if (c->d_prof) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipMemset((char*)c->d_prof + 8 * 64 * VIL_PROF_SLOTS, 0, 8 * 2 * VIL_PROF_WGS)); }
        return VIL_OK;
    }
    if (!c->d_prof || max_workgroups < 0) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipStreamSynchronize(c->stream));
    const int n = std::min<int>(max_workgroups, VIL_PROF_WGS);
    HIPCHK(hipMemcpy(times, (char*)c->d_prof + 8 * 64 * VIL_PROF_SLOTS, 8 * 2 * (size_t)n, hipMemcpyDeviceToHost));
    return VIL_OK;
}
int vil_debug_set_slim_emul(vil_ctx* c, int32_t on) { if (!c) return VIL_ERR_INVALID_ARGUMENT; c->slim_emul = on != 0; return VIL_OK; }
int vil_debug_get_launch_structure(vil_ctx* c, int32_t* launches_per_iteration, int32_t* one_launch) {
    if (!c || !c->uploaded) return VIL_ERR_INVALID_ARGUMENT;
    const bool merged = c->P.rs_merged != 0 && !c->split;
    if (launches_per_iteration) *launches_per_iteration = (c->fused && !c->split) ? ((c->persist && !c->profiling) ? 0 : 1) : (merged ? 2 : 3);      // 0: the whole solve is one resident launch (k_solve)
    if (one_launch) *one_launch = (c->fused && !c->split) ? 1 : 0;
    return VIL_OK;
}
int vil_debug_set_split(vil_ctx* c, int32_t on) { if (!c) return VIL_ERR_INVALID_ARGUMENT; c->force_split = on != 0; c->uploaded = false; c->resident_kind = 0; return VIL_OK; }
int vil_set_gauge_fix(vil_ctx* c, int32_t on) { if (!c) return VIL_ERR_INVALID_ARGUMENT; c->gauge_on = on != 0; return VIL_OK; }      // (takes effect in the next solve)

int vil_lidar_reset(vil_ctx* c) {
    if (!c) return VIL_ERR_INVALID_ARGUMENT;
    c->slabs.clear(); c->free_slots.clear();
    for (int q = c->nslot - 1; q >= 0; --q) c->free_slots.push_back(q);
    if (c->lidar_resident) { c->uploaded = false; c->resident_kind = 0; }      // the resident window's chunk list points into the old slab order
    return VIL_OK;
}
int vil_lidar_count(vil_ctx* c, int32_t* n_slabs, int32_t* n_plane, int32_t* n_edge) {
    if (!c) return VIL_ERR_INVALID_ARGUMENT;
    int np = 0, ne = 0;
    for (auto& sl : c->slabs) { np += sl.np; ne += sl.ne; }
    if (n_slabs) *n_slabs = (int)c->slabs.size(); if (n_plane) *n_plane = np; if (n_edge) *n_edge = ne;
    return VIL_OK;
}
int vil_lidar_drop(vil_ctx* c, int32_t slab) {
    if (!c || slab < 0 || slab >= (int)c->slabs.size()) return VIL_ERR_INVALID_ARGUMENT;
    c->free_slots.push_back(c->slabs[slab].slot);
    c->slabs.erase(c->slabs.begin() + slab);          // the later frames move down: an index remap, the points stay where they are
    if (c->lidar_resident) { c->uploaded = false; c->resident_kind = 0; }      // slab <-> pose changed under the resident window's chunk list
    return VIL_OK;
}
int vil_lidar_push(vil_ctx* c, int32_t n_plane, const double* plane_const, int32_t n_edge, const double* edge_const) {
    if (!c || n_plane < 0 || n_edge < 0 || (n_plane > 0 && !plane_const) || (n_edge > 0 && !edge_const)) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    const int np_all = n_plane, ne_all = n_edge;
    if (c->world > 1 && c->has_comm()) {       // factor set sharded over ranks: every rank is handed the whole frame and keeps its contiguous slice
        const int pb = (int)((long long)n_plane * c->rank / c->world), pe = (int)((long long)n_plane * (c->rank + 1) / c->world);
        const int eb = (int)((long long)n_edge * c->rank / c->world), ee = (int)((long long)n_edge * (c->rank + 1) / c->world);
        plane_const += (size_t)7 * pb; n_plane = pe - pb; edge_const += (size_t)9 * eb; n_edge = ee - eb;
    }
    // capacity: points per slab and number of physical slabs only ever grow (the existing slabs are copied once when they do)
    if (n_plane > c->cap_p || n_edge > c->cap_e || c->free_slots.empty()) {
        const int ncp = std::max(c->cap_p, ((n_plane + n_plane / 4 + 255) / 256) * 256), nce = std::max(c->cap_e, ((n_edge + n_edge / 4 + 255) / 256) * 256);
        const int nns = std::max(c->nslot, (int)c->slabs.size() + 4);
        double *npl = nullptr, *ned = nullptr;
        HIPCHK(hipMalloc(&npl, 8 * (size_t)7 * nns * std::max(ncp, 256))); HIPCHK(hipMalloc(&ned, 8 * (size_t)9 * nns * std::max(nce, 256)));
        const int cp2 = std::max(ncp, 256), ce2 = std::max(nce, 256);
        HIPCHK(hipStreamSynchronize(c->stream));
        for (auto& sl : c->slabs) {
            for (int q = 0; q < 7 && sl.np; ++q) HIPCHK(hipMemcpyAsync(npl + ((size_t)q * nns + sl.slot) * cp2, c->d_pl + ((size_t)q * c->nslot + sl.slot) * c->cap_p, 8 * (size_t)sl.np, hipMemcpyDeviceToDevice, c->stream));
            for (int q = 0; q < 9 && sl.ne; ++q) HIPCHK(hipMemcpyAsync(ned + ((size_t)q * nns + sl.slot) * ce2, c->d_ed + ((size_t)q * c->nslot + sl.slot) * c->cap_e, 8 * (size_t)sl.ne, hipMemcpyDeviceToDevice, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        if (c->d_pl) hipFree(c->d_pl); if (c->d_ed) hipFree(c->d_ed);
        c->d_pl = npl; c->d_ed = ned;
        for (int q = nns - 1; q >= c->nslot; --q) c->free_slots.push_back(q);
        c->cap_p = cp2; c->cap_e = ce2; c->nslot = nns;
        c->uploaded = false;                          // a resident problem holds pointers / strides of the old tables
    }
    const size_t need = (size_t)7 * n_plane + (size_t)9 * n_edge;
    if (need > c->lstage_cap) {
        if (c->lpush_pending) { HIPCHK(hipEventSynchronize(c->lpush_ev)); c->lpush_pending = false; }
        if (c->d_lstage) hipFree(c->d_lstage); if (c->h_lstage) hipHostFree(c->h_lstage);
        c->d_lstage = nullptr; c->h_lstage = nullptr; c->lstage_cap = 0;
        const size_t cap = need + need / 4 + 1024;
        HIPCHK(hipMalloc(&c->d_lstage, 8 * cap)); HIPCHK(hipHostMalloc(&c->h_lstage, 8 * cap, hipHostMallocDefault));
        c->lstage_cap = cap;
    }
    if (c->lpush_pending) { HIPCHK(hipEventSynchronize(c->lpush_ev)); c->lpush_pending = false; }     // the previous frame's DMA has left the pinned image
    const int slot = c->free_slots.back(); c->free_slots.pop_back();
    if (need) {
        if (n_plane) memcpy(c->h_lstage, plane_const, 8 * (size_t)7 * n_plane);
        if (n_edge) memcpy(c->h_lstage + (size_t)7 * n_plane, edge_const, 8 * (size_t)9 * n_edge);
        HIPCHK(hipMemcpyAsync(c->d_lstage, c->h_lstage, 8 * need, hipMemcpyHostToDevice, c->stream));
        if (!c->lpush_ev) HIPCHK(hipEventCreateWithFlags(&c->lpush_ev, hipEventDisableTiming));
        HIPCHK(hipEventRecord(c->lpush_ev, c->stream)); c->lpush_pending = true;
        // point-major rows -> the component-major tables the sweep reads (coalesced, thread = point)
        if (n_plane) hipLaunchKernelGGL(k_aos2soa, dim3((n_plane + 255) / 256), dim3(256), 0, c->stream, c->d_lstage, n_plane, 7, c->d_pl + (size_t)slot * c->cap_p, (size_t)c->nslot * c->cap_p);
        if (n_edge) hipLaunchKernelGGL(k_aos2soa, dim3((n_edge + 255) / 256), dim3(256), 0, c->stream, c->d_lstage + (size_t)7 * n_plane, n_edge, 9, c->d_ed + (size_t)slot * c->cap_e, (size_t)c->nslot * c->cap_e);
    }
    c->slabs.push_back({n_plane, n_edge, slot, np_all, ne_all});
    if (c->lidar_resident) { c->uploaded = false; c->resident_kind = 0; }      // (a recycled slot would feed this frame's points to the old pose)
    return VIL_OK;
}

static int marginalize_resident_impl(vil_ctx* c, const vil_state* s, const vil_options* o, const vil_marg_spec* spec, vil_prior_out* out, const bool to_slot, vil_win_prior_info* winfo) {
    if (!c || !o || !spec || (!to_slot && (!s || !out)) || !c->uploaded || c->resident_kind != 1) return VIL_ERR_INVALID_ARGUMENT;
    const int K = c->K;
    if (K < 3 || (s && (s->K != K || s->L != c->L))) return VIL_ERR_INVALID_ARGUMENT;
    int none = 0; int& out_n = out ? out->n : none;
    HIPCHK(hipSetDevice(c->device));
    const vil_ctx::MargMeta& mm = c->mm;
    const bool old_ = spec->flag == VIL_MARGIN_OLD;
    const int drop_pose = old_ ? 0 : K - 2;
    std::vector<char> pose_t(K, 0), sb_t(K, 0);
    bool ex_t = false, td_t = false;
    int n_lm_elim = 0;
    if (mm.has_prior) {
        bool has_drop = false;
        for (size_t b = 0; b < mm.prior_kind.size(); ++b) {
            const int kind = mm.prior_kind[b], idx = mm.prior_index[b];
            if (kind == VIL_BLK_POSE) { pose_t[idx] = 1; if (idx == drop_pose) has_drop = true; }
            else if (kind == VIL_BLK_SPEEDBIAS) sb_t[idx] = 1;
            else if (kind == VIL_BLK_EX) ex_t = true; else td_t = true;
        }
        if (!old_ && !has_drop) { out_n = -1; if (winfo) winfo->n = -1; return VIL_OK; }     // estimator.cpp:1620-1621: prior kept as is
    } else if (!old_) { out_n = -1; if (winfo) winfo->n = -1; return VIL_OK; }
    int icp_m = -1, lps_m = -1;
    if (old_) {
        if (mm.imu01) { pose_t[0] = pose_t[1] = 1; sb_t[0] = sb_t[1] = 1; }
        if (mm.n_lm0 > 0) { pose_t[0] = 1; for (int k = 0; k < K; ++k) if (mm.obs0[k]) pose_t[k] = 1; ex_t = true; if (mm.use_td) td_t = true; n_lm_elim = mm.n_lm0; }
        if (spec->icp_marg >= 0 && 4 * (size_t)spec->icp_marg + 3 < mm.icp_ids.size()) {
            icp_m = spec->icp_marg;
            if (mm.icp_ids[4 * icp_m] != 0) return VIL_ERR_UNSUPPORTED;      // the remembered constraint starts in frame 0 (estimator.cpp:1381-1389)
            for (int b = 0; b < 4; ++b) pose_t[mm.icp_ids[4 * icp_m + b]] = 1;
        }
        if (spec->lps_marg >= 0 && 2 * (size_t)spec->lps_marg + 1 < mm.lps_ids.size()) {
            lps_m = spec->lps_marg;
            if (mm.lps_ids[2 * lps_m] != 0) return VIL_ERR_UNSUPPORTED;
            for (int b = 0; b < 2; ++b) pose_t[mm.lps_ids[2 * lps_m + b]] = 1;
        }
        if (mm.lidar0) pose_t[0] = 1;
    }
    const SolveOpts so = to_dev_opts(o);
    int st = init_ctl(c, o, 2);
    if (st != VIL_OK) return st;
    const DevP keep = c->P;
    c->P.marg = old_ ? 1 : 2; c->P.marg_icp = icp_m; c->P.marg_lps = lps_m;
    launch_sweep(c, so);                               // the resident tables at the resident (solved, gauge-fixed) state; masks select the factors
    st = launch_reduce_step(c, so, false);             // (sharded: A and b of this rank's factors, all-reduced once)
    c->P = keep;
    if (st != VIL_OK) return st;
    c->resident_kind = 2;                              // the work space now holds the marginalisation's linearisation
    return marg_finish(c, K, old_, drop_pose, pose_t, sb_t, ex_t, td_t, mm.use_td, n_lm_elim, s, out, to_slot, winfo);
}
int vil_marginalize_resident(vil_ctx* c, const vil_state* s, const vil_options* o, const vil_marg_spec* spec, vil_prior_out* out) {
    return marginalize_resident_impl(c, s, o, spec, out, false, nullptr);
}

// ---- the fully resident window (include/vilsolve.h: vil_win_*; device side: vil_window.hpp) ------------------------------------------
int vil_win_open(vil_ctx* c, const vil_win_cfg* cfg) {
    if (!c || !cfg || cfg->K < 3 || cfg->K > VIL_WIN_MAXK || cfg->max_tracks < 1 || cfg->max_samples < 1) return VIL_ERR_INVALID_ARGUMENT;
    if (c->ipc && c->world > 1 && !c->ipc->ready) return VIL_ERR_COMM;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    auto& w = c->win;
    const int K = cfg->K, NF = K + 2;
    int nmax, nblk_max, x0max;
    vil_prior_capacity(K, &nmax, &nblk_max, &x0max);
    if (!w.open || w.K != K || w.T != cfg->max_tracks || w.S != cfg->max_samples) {
        hipFree(w.d_store); hipFree(w.d_samp); hipFree(w.d_hdr); hipFree(w.d_rec); hipFree(w.d_U); hipFree(w.d_prior[0]); hipFree(w.d_prior[1]);
        w.d_store = w.d_samp = w.d_hdr = w.d_rec = w.d_U = w.d_prior[0] = w.d_prior[1] = nullptr; w.open = false;
        w.K = K; w.T = cfg->max_tracks; w.S = cfg->max_samples; w.NF = NF; w.nmax = nmax; w.x0max = x0max;
        HIPCHK(hipMalloc(&w.d_store, 8 * (size_t)NF * w.T * VIL_WIN_OBS));
        HIPCHK(hipMalloc(&w.d_samp, 8 * (size_t)NF * 7 * w.S)); HIPCHK(hipMalloc(&w.d_hdr, 8 * (size_t)NF * 12));
        HIPCHK(hipMalloc(&w.d_rec, 8 * (size_t)NF * 287)); HIPCHK(hipMalloc(&w.d_U, 8 * (size_t)NF * 225));
        for (int q = 0; q < 2; ++q) HIPCHK(hipMalloc(&w.d_prior[q], 8 * w.prior_doubles()));
        if (!w.d_wstat) HIPCHK(hipMalloc(&w.d_wstat, 64));
    }
    HIPCHK(hipMemsetAsync(w.d_store, 0, 8 * (size_t)NF * w.T * VIL_WIN_OBS, c->stream));
    HIPCHK(hipMemsetAsync(w.d_rec, 0, 8 * (size_t)NF * 287, c->stream)); HIPCHK(hipMemsetAsync(w.d_U, 0, 8 * (size_t)NF * 225, c->stream));
    HIPCHK(hipMemsetAsync(w.d_wstat, 0, 64, c->stream));
    w.cfg = *cfg; w.open = true;
    w.fslot.clear(); w.islot.clear(); w.free_f.clear(); w.free_i.clear();
    for (int q = NF - 1; q >= 0; --q) { w.free_f.push_back(q); w.free_i.push_back(q); }
    w.ns.assign(NF, 0); w.sum_dt.assign(NF, 0.0);
    w.cur = 0; w.pn = 0; w.pm = 0; w.pkind.clear(); w.pindex.clear(); w.pcol.clear();
    c->uploaded = false; c->resident_kind = 0;
    return vil_lidar_reset(c);
}

int vil_win_push_frame(vil_ctx* c, const vil_win_frame* f) {
    if (!c || !f || !c->win.open) return VIL_ERR_INVALID_ARGUMENT;
    auto& w = c->win;
    if ((int)w.fslot.size() >= w.K || f->n_samples < 0 || f->n_samples > w.S || f->n_obs < 0 || f->n_obs > w.T) return VIL_ERR_INVALID_ARGUMENT;
    if ((f->n_samples > 0 && (!f->dt || !f->acc || !f->gyr)) || (f->n_obs > 0 && (!f->obs_track || !f->obs))) return VIL_ERR_INVALID_ARGUMENT;
    for (int q = 0; q < f->n_obs; ++q) if (f->obs_track[q] < 0 || f->obs_track[q] >= w.T) return VIL_ERR_INVALID_ARGUMENT;
    // (the LiDAR arguments are checked BEFORE the frame / IMU slots are committed: a frame whose points fail must not leave fslot one entry ahead of
    //  the slabs -- "slab i <-> pose K - count + i" would then attach every later frame's points to the wrong pose)
    if (f->n_plane < 0 || f->n_edge < 0 || (f->n_plane > 0 && !f->plane_const) || (f->n_edge > 0 && !f->edge_const)) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    const int ns = f->n_samples, no = f->n_obs;
    // one staging block, one DMA: [hdr 12 | dt | acc | gyr | observations | track slots]
    const size_t o_samp = 0, o_obs = 8 * (size_t)(12 + 7 * ns), o_trk = o_obs + 8 * (size_t)no * VIL_WIN_OBS, total = o_trk + 4 * (size_t)no + 64;
    if (w.pending) { HIPCHK(hipEventSynchronize(w.ev)); w.pending = false; }     // the previous frame's DMA has left the pinned block
    if (total > w.stage_cap) {
        HIPCHK(hipStreamSynchronize(c->stream));
        if (w.h_stage) hipHostFree(w.h_stage); hipFree(w.d_stage); w.h_stage = nullptr; w.d_stage = nullptr; w.stage_cap = 0;
        const size_t cap = 2 * total + 4096;
        HIPCHK(hipHostMalloc((void**)&w.h_stage, cap, hipHostMallocDefault)); HIPCHK(hipMalloc(&w.d_stage, cap));
        w.stage_cap = cap;
    }
    double* hs = (double*)(w.h_stage + o_samp);
    memcpy(hs, f->acc0, 24); memcpy(hs + 3, f->gyr0, 24); memcpy(hs + 6, f->lin_ba, 24); memcpy(hs + 9, f->lin_bg, 24);
    double sum = 0.0;
    if (ns) { memcpy(hs + 12, f->dt, 8 * (size_t)ns); memcpy(hs + 12 + ns, f->acc, 24 * (size_t)ns); memcpy(hs + 12 + 4 * (size_t)ns, f->gyr, 24 * (size_t)ns); for (int q = 0; q < ns; ++q) sum += f->dt[q]; }
    if (no) { memcpy(w.h_stage + o_obs, f->obs, 8 * (size_t)no * VIL_WIN_OBS); memcpy(w.h_stage + o_trk, f->obs_track, 4 * (size_t)no); }
    HIPCHK(hipMemcpyAsync(w.d_stage, w.h_stage, o_trk + 4 * (size_t)no, hipMemcpyHostToDevice, c->stream));
    if (!w.ev) HIPCHK(hipEventCreateWithFlags(&w.ev, hipEventDisableTiming));
    HIPCHK(hipEventRecord(w.ev, c->stream)); w.pending = true;
    const int fs = w.free_f.back(), is = w.free_i.back();
    w.free_f.pop_back(); w.free_i.pop_back();
    WinFrameIn A;
    A.n_obs = no; A.track = (const int*)(w.d_stage + o_trk); A.obs = (const double*)(w.d_stage + o_obs); A.store = w.d_store; A.T = w.T; A.fslot = fs;
    A.ns = ns; A.samp = (const double*)(w.d_stage + o_samp); A.hdr = w.d_hdr + 12 * (size_t)is; A.dt = w.dt(is); A.acc = w.acc(is); A.gyr = w.gyr(is);
    const int nbo = (no * VIL_WIN_OBS + 255) / 256;
    hipLaunchKernelGGL(k_win_frame_in, dim3(nbo + 1), dim3(256), 0, c->stream, A, nbo);
    if (ns > 0) {                                        // the interval's record and its sqrt-information, where the solver will read them
        vpre_launch_slot(c->stream, ns, w.dt(is), w.acc(is), w.gyr(is), w.d_hdr + 12 * (size_t)is, w.cfg.noise, w.d_rec + 287 * (size_t)is);
        hipLaunchKernelGGL(k_imu_sqrtinfo, dim3(1), dim3(VIL_THREADS), 0, c->stream, w.d_rec + 287 * (size_t)is, w.d_U + 225 * (size_t)is, w.d_wstat, 1);
    }
    w.fslot.push_back(fs); w.islot.push_back(is); w.ns[is] = ns; w.sum_dt[is] = sum;
    c->uploaded = false; c->resident_kind = 0;
    const int lst = vil_lidar_push(c, f->n_plane, f->plane_const, f->n_edge, f->edge_const);
    if (lst != VIL_OK) {                                 // (an allocation failed): the frame is taken back -- slots returned, nothing refers to what the kernels above wrote
        w.fslot.pop_back(); w.islot.pop_back(); w.free_f.push_back(fs); w.free_i.push_back(is);
    }
    return lst;
}

int vil_win_drop_frame(vil_ctx* c, int32_t flag) {
    if (!c || !c->win.open) return VIL_ERR_INVALID_ARGUMENT;
    auto& w = c->win;
    const int n = (int)w.fslot.size();
    if (n < 2) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    c->uploaded = false; c->resident_kind = 0;
    if (flag == VIL_MARGIN_OLD) {
        // frame 0 leaves.  Its IMU slot held the interval that ENDED in it -- not a factor of the window; the interval (0, 1) stays with frame 1
        w.free_f.push_back(w.fslot[0]); w.free_i.push_back(w.islot[0]);
        w.fslot.erase(w.fslot.begin()); w.islot.erase(w.islot.begin());
        return vil_lidar_drop(c, 0);
    }
    // MARGIN_SECOND_NEW (estimator.cpp:1754-1786): frame n-2 leaves; the newest frame takes its place and its samples continue the interval
    // that ended in n-2 (same first measurement, same linearisation point): appended on the device, re-integrated, re-factored
    const int ia = w.islot[n - 2], ib = w.islot[n - 1];
    if (w.ns[ia] + w.ns[ib] > w.S) return VIL_ERR_UNSUPPORTED;
    if (w.ns[ib] > 0) {
        hipLaunchKernelGGL(k_win_append, dim3(1), dim3(256), 0, c->stream, w.dt(ia), w.acc(ia), w.gyr(ia), w.ns[ia], w.dt(ib), w.acc(ib), w.gyr(ib), w.ns[ib]);
        w.ns[ia] += w.ns[ib]; w.sum_dt[ia] += w.sum_dt[ib];
        vpre_launch_slot(c->stream, w.ns[ia], w.dt(ia), w.acc(ia), w.gyr(ia), w.d_hdr + 12 * (size_t)ia, w.cfg.noise, w.d_rec + 287 * (size_t)ia);
        hipLaunchKernelGGL(k_imu_sqrtinfo, dim3(1), dim3(VIL_THREADS), 0, c->stream, w.d_rec + 287 * (size_t)ia, w.d_U + 225 * (size_t)ia, w.d_wstat, 1);
    }
    w.free_f.push_back(w.fslot[n - 2]); w.free_i.push_back(ib);
    w.fslot.erase(w.fslot.begin() + (n - 2));            // the newest frame's observations and LiDAR points stay where they are
    w.islot.erase(w.islot.begin() + (n - 1));            // ... and it now ends the merged interval
    return vil_lidar_drop(c, n - 2);
}

int vil_win_solve(vil_ctx* c, const vil_win_problem* wp, vil_state* s, const vil_options* o, vil_summary* sum) {
    if (!c || !wp || !s || !o || !sum || !c->win.open) return VIL_ERR_INVALID_ARGUMENT;
    auto& w = c->win;
    const int K = w.K, L = wp->L;
    if ((int)w.fslot.size() != K || L < 0 || s->K != K || s->L != L) return VIL_ERR_INVALID_ARGUMENT;
    if (L > 0 && (!wp->lm_track || !wp->lm_start || !wp->lm_nobs)) return VIL_ERR_INVALID_ARGUMENT;
    int n_vis = 0;
    for (int l = 0; l < L; ++l) {
        if (wp->lm_track[l] < 0 || wp->lm_track[l] >= w.T || wp->lm_start[l] < 0 || wp->lm_nobs[l] < 2 || wp->lm_start[l] + wp->lm_nobs[l] > K) return VIL_ERR_INVALID_ARGUMENT;
        n_vis += wp->lm_nobs[l] - 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    vil_problem q; memset(&q, 0, sizeof q);
    q.K = K; q.L = L; q.pose_const = wp->pose_const; q.sb_const = wp->sb_const; q.lm_const = wp->lm_const;
    q.ex_const = wp->ex_const; q.td_const = wp->td_const; q.use_td = w.cfg.use_td;
    int imu_i[VIL_WIN_MAXK], imu_j[VIL_WIN_MAXK];
    for (int k = 0; k + 1 < K; ++k) { imu_i[k] = k; imu_j[k] = k + 1; }
    q.n_imu = K - 1; q.imu_i = imu_i; q.imu_j = imu_j; q.imu_const = nullptr;
    q.prior.n = w.pn; q.prior.nblk = (int)w.pkind.size(); q.prior.blk_kind = w.pkind.data(); q.prior.blk_index = w.pindex.data(); q.prior.blk_col = w.pcol.data();
    q.n_icp = wp->n_icp; q.icp_ids = wp->icp_ids; q.icp_const = wp->icp_const; q.n_lps = wp->n_lps; q.lps_ids = wp->lps_ids; q.lps_const = wp->lps_const;
    q.n_plane = q.n_edge = VIL_LIDAR_RESIDENT;
    memcpy(q.q_lb, w.cfg.q_lb, sizeof q.q_lb); memcpy(q.t_lb, w.cfg.t_lb, sizeof q.t_lb); memcpy(q.G, w.cfg.G, sizeof q.G);
    q.sqrt_info_px = w.cfg.sqrt_info_px; q.tr_over_row = w.cfg.tr_over_row;
    WinSrc ws; memset(&ws, 0, sizeof ws);
    ws.lm_track = wp->lm_track; ws.lm_startf = wp->lm_start; ws.lm_nobs = wp->lm_nobs; ws.n_vis = n_vis; ws.T = w.T;
    ws.d_store = w.d_store; ws.fslot = w.fslot.data(); ws.islot = w.islot.data(); ws.d_rec = w.d_rec; ws.d_U = w.d_U; ws.sum_dt = w.sum_dt.data();
    ws.pJ0 = w.pJ0(w.cur); ws.pr0 = w.pr0(w.cur); ws.px0 = w.px0(w.cur); ws.pH = w.pH(w.cur); ws.pg0 = w.pg0(w.cur); ws.pc0 = w.pc0(w.cur);
    ws.lb = 0; ws.le = L; ws.f0 = 0; ws.n_vis_all = n_vis;
    // vil_win_solve returns the gauge-fixed state and vil_win_marginalize linearises at it (double2vector() precedes the marginalisation,
    // estimator.cpp:1419 / :1487): the device gauge fix is part of the resident window whatever vil_set_gauge_fix was left at
    const bool gauge_was = c->gauge_on; c->gauge_on = true;
    int st;
    if (c->world > 1 && c->has_comm()) {
        // SURVEY 8e on the resident window: every rank was handed every frame (vil_win_push_frame: the observation store and the IMU slots are whole on
        // every rank, the LiDAR slabs hold the rank's slice) and is handed the same small tables here; it keeps the visual factors of ITS landmark range
        // -- the rule of vil_shard_ranges on the landmark list -- and rank 0 alone the IMU / prior / ICP / LPS factors.  The prior slot is written by
        // every rank from the all-reduced marginalisation system (identical bits), read by rank 0.
        std::vector<int> lms(L + 1, 0);
        for (int l = 0; l < L; ++l) lms[l + 1] = lms[l] + wp->lm_nobs[l] - 1;
        OwnSeg& S = c->own; memset(&S, 0, sizeof S);
        S.n = c->world <= 8 ? c->world : 0; S.Lp = std::max(L, 1);          // (more than eight ranks: RCCL only, which sums the whole set -- n = 0)
        for (int r = 0; r <= S.n; ++r) { S.lb[r] = shard_cut(lms, L, r, c->world); S.fb[r] = lms[S.lb[r]]; }
        S.cam = ((size_t)(15 * K + 7) * (15 * K + 7) + 3 * (size_t)(15 * K + 7) + 3 + 1) & ~size_t(1);
        ws.lb = shard_cut(lms, L, c->rank, c->world); ws.le = shard_cut(lms, L, c->rank + 1, c->world); ws.f0 = lms[ws.lb]; ws.n_vis = lms[ws.le] - lms[ws.lb];
        c->lm_b = ws.lb; c->lm_e = ws.le;
        vil_problem ql = q;
        if (c->rank != 0) { ql.n_imu = 0; ql.n_icp = 0; ql.n_lps = 0; ql.prior.n = 0; ql.prior.nblk = 0; }
        st = comm_agree(c, upload_impl(c, &ql, s, true, nullptr, &q, ws.f0, false, &ws));
        if (st != VIL_OK) c->uploaded = false;
    } else st = upload_impl(c, &q, s, false, nullptr, nullptr, 0, false, &ws);
    if (st != VIL_OK) { c->gauge_on = gauge_was; return st; }
    c->resident_kind = 1;
    c->P.setup_stat = w.d_wstat;                       // the window's sticky status word: a failed pre-integration / prior ends the solve with it
    const auto t1 = std::chrono::steady_clock::now();
    st = vil_solve_resident(c, o, sum);
    c->gauge_on = gauge_was;                           // (the caller's setting governs vil_solve / vil_solve_resident of other uploads again)
    sum->t_prepare_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (st != VIL_OK) return st;                       // state left unchanged on any error
    const auto t2 = std::chrono::steady_clock::now();
    st = vil_download_state(c, s);
    sum->t_readback_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count();
    return st;
}

int vil_win_marginalize(vil_ctx* c, const vil_options* o, const vil_marg_spec* spec, vil_win_prior_info* info) {
    if (!c || !c->win.open) return VIL_ERR_INVALID_ARGUMENT;
    return marginalize_resident_impl(c, nullptr, o, spec, nullptr, true, info);
}

int vil_win_prior_download(vil_ctx* c, vil_prior_out* out) {
    if (!c || !out || !c->win.open) return VIL_ERR_INVALID_ARGUMENT;
    auto& w = c->win;
    HIPCHK(hipSetDevice(c->device));
    int& wst = c->h_word[8]; wst = 0;
    HIPCHK(hipMemcpyAsync(&wst, w.d_wstat, 4, hipMemcpyDeviceToHost, c->stream));
    out->n = w.pn; out->nblk = (int)w.pkind.size(); out->m = w.pm;
    if (w.pn > 0) {
        const size_t n = (size_t)w.pn;
        int xo = 0;
        for (size_t b = 0; b < w.pkind.size(); ++b) { out->blk_kind[b] = w.pkind[b]; out->blk_index[b] = w.pindex[b]; out->blk_col[b] = w.pcol[b]; xo += (w.pkind[b] == VIL_BLK_POSE || w.pkind[b] == VIL_BLK_EX) ? 7 : (w.pkind[b] == VIL_BLK_SPEEDBIAS ? 9 : 1); }
        HIPCHK(hipMemcpyAsync(out->J0, w.pJ0(w.cur), 8 * n * n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(out->r0, w.pr0(w.cur), 8 * n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(out->x0, w.px0(w.cur), 8 * (size_t)xo, hipMemcpyDeviceToHost, c->stream));
        if (out->A) HIPCHK(hipMemcpyAsync(out->A, w.pH(w.cur), 8 * n * n, hipMemcpyDeviceToHost, c->stream));      // J0^T J0
        if (out->b) HIPCHK(hipMemcpyAsync(out->b, w.pg0(w.cur), 8 * n, hipMemcpyDeviceToHost, c->stream));          // J0^T r0
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    return wst;
}

int vil_win_prior_set(vil_ctx* c, const vil_prior* pr) {
    if (!c || !c->win.open) return VIL_ERR_INVALID_ARGUMENT;
    auto& w = c->win;
    HIPCHK(hipSetDevice(c->device));
    if (!pr || pr->n <= 0) { w.pn = 0; w.pm = 0; w.pkind.clear(); w.pindex.clear(); w.pcol.clear(); return VIL_OK; }
    if (pr->n > w.nmax || pr->nblk <= 0 || pr->nblk > VIL_WIN_MAXBLK || !pr->blk_kind || !pr->blk_index || !pr->blk_col || !pr->x0 || !pr->J0 || !pr->r0) return VIL_ERR_INVALID_ARGUMENT;
    const size_t n = (size_t)pr->n;
    int xo = 0;
    for (int b = 0; b < pr->nblk; ++b) xo += (pr->blk_kind[b] == VIL_BLK_POSE || pr->blk_kind[b] == VIL_BLK_EX) ? 7 : (pr->blk_kind[b] == VIL_BLK_SPEEDBIAS ? 9 : 1);
    if (xo > w.x0max) return VIL_ERR_INVALID_ARGUMENT;
    const int dst = 1 - w.cur;
    // stage J0 | r0 in the marginalisation work space layout k_prior_commit reads, x0 directly
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(w.pJ0(dst), pr->J0, 8 * n * n, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(w.pr0(dst), pr->r0, 8 * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(w.px0(dst), pr->x0, 8 * (size_t)xo, hipMemcpyHostToDevice));
    PriorCommit pc; memset(&pc, 0, sizeof pc);
    pc.n = pr->n; pc.nblk = 0; pc.J0 = w.pJ0(dst); pc.r0 = w.pr0(dst); pc.x = nullptr; pc.K = w.K;      // nblk = 0: x0 already in place
    pc.pJ0 = w.pJ0(dst); pc.pr0 = w.pr0(dst); pc.px0 = w.px0(dst); pc.pH = w.pH(dst); pc.pg0 = w.pg0(dst); pc.pc0 = w.pc0(dst); pc.status = w.d_wstat;
    {
        const size_t pl = 8 * (n * (n + 1) + n + 8);
        if (pl > 48 * 1024 && (int)pl > c->attr_commit) { HIPCHK(hipFuncSetAttribute((const void*)k_prior_commit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl)); c->attr_commit = (int)pl; }
        hipLaunchKernelGGL(k_prior_commit, dim3(16), dim3(256), pl, c->stream, pc);
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    w.cur = dst; w.pn = pr->n; w.pm = 0;
    w.pkind.assign(pr->blk_kind, pr->blk_kind + pr->nblk); w.pindex.assign(pr->blk_index, pr->blk_index + pr->nblk); w.pcol.assign(pr->blk_col, pr->blk_col + pr->nblk);
    return VIL_OK;
}

int vil_comm_unique_id(void* id128) {
    if (!id128) return VIL_ERR_INVALID_ARGUMENT;
    if (!g_rccl.load()) return VIL_ERR_COMM;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return VIL_ERR_COMM;
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return VIL_OK;
}
int vil_comm_init(vil_ctx* c, const void* id128, int rank, int world) {
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    if (!g_rccl.load()) return VIL_ERR_COMM;
    if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    if (g_rccl.CommInitRank(&c->comm, world, id, rank) != ncclSuccess) return VIL_ERR_COMM;
    c->rank = rank; c->world = world; c->uploaded = false; c->lcomm.reset();
    return VIL_OK;
}
// multi-process peer-buffer exchange: export the inbox, gather the handles (the launcher's job: torch.distributed / MPI / a pipe), import
int vil_comm_ipc_export(vil_ctx* c, int rank, int world, size_t max_doubles, void* handle64) {
    if (!c || !handle64 || world < 1 || world > 8 || rank < 0 || rank >= world || max_doubles < 64) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->ipc) { for (int r = 0; r < c->ipc->world; ++r) if (r != c->ipc->rank && c->ipc->peer_base[r]) hipIpcCloseMemHandle(c->ipc->peer_base[r]); if (c->ipc->base) hipFree(c->ipc->base); c->ipc.reset(); }
    auto ic = std::make_shared<IpcComm>();
    ic->world = world; ic->rank = rank; ic->cap = (max_doubles + 31) & ~size_t(31);
    ic->off_flags = 8 * (size_t)world * 2 * ic->cap; ic->off_seq = ic->off_flags + 64 * (size_t)world * 2;
    const size_t bytes = ic->off_seq + 256;
    HIPCHK(hipMalloc((void**)&ic->base, bytes));
    HIPCHK(hipMemset(ic->base, 0, bytes));
    hipIpcMemHandle_t h;
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "vil_comm_ipc_export hands out 64 bytes");
    HIPCHK(hipIpcGetMemHandle(&h, ic->base));
    memset(handle64, 0, 64); memcpy(handle64, &h, sizeof h);
    ic->peer_base[rank] = ic->base;
    c->ipc = ic; c->rank = rank; c->world = world; c->uploaded = false; c->lcomm.reset();
    if (c->comm && g_rccl.CommDestroy) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    return VIL_OK;
}
int vil_comm_ipc_init(vil_ctx* c, const void* handles) {
    if (!c || !handles || !c->ipc) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    IpcComm* ic = c->ipc.get();
    for (int r = 0; r < ic->world; ++r) {
        if (r == ic->rank) continue;
        hipIpcMemHandle_t h; memcpy(&h, (const char*)handles + 64 * (size_t)r, sizeof h);
        void* p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return VIL_ERR_COMM; }
        ic->peer_base[r] = p;
    }
    ic->ready = true;
    return VIL_OK;
}

int vil_comm_init_local(vil_ctx** ctxs, int n) {
    if (!ctxs || n < 1 || n > 8) return VIL_ERR_INVALID_ARGUMENT;
    for (int r = 0; r < n; ++r) if (!ctxs[r]) return VIL_ERR_INVALID_ARGUMENT;
    // k_sum_peers reads the other contexts' buffers directly: contexts on different devices need peer access in both
    // directions (xGMI / PCIe P2P), checked and enabled here; without it the configuration is refused, not faulted at run time
    for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) {
        if (ctxs[a]->device == ctxs[b]->device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, ctxs[a]->device, ctxs[b]->device) != hipSuccess || !can) return VIL_ERR_UNSUPPORTED;
        if (hipSetDevice(ctxs[a]->device) != hipSuccess) return VIL_ERR_DEVICE;
        const hipError_t e = hipDeviceEnablePeerAccess(ctxs[b]->device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return VIL_ERR_UNSUPPORTED;
        (void)hipGetLastError();
    }
    auto lc = std::make_shared<LocalComm>();
    if (pthread_barrier_init(&lc->bar, nullptr, (unsigned)n) != 0) return VIL_ERR_COMM;
    lc->n = n;
    for (int r = 0; r < n; ++r) {
        vil_ctx* c = ctxs[r];
        if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
        c->lcomm = lc; c->rank = r; c->world = n; c->uploaded = false; c->ipc.reset();
    }
    return VIL_OK;
}

}  // extern "C"
