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
#include "vil_sweep.hpp"
#include "vil_eval.hpp"
#include "vil_step.hpp"
#include "vil_marg.hpp"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[vilsolve] HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return VIL_ERR_DEVICE; } } while (0)

namespace {

struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
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
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        return GetUniqueId && CommInitRank && AllReduce && CommDestroy;
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
__global__ void k_sum_peers(double* out, PeerPtrs pp, size_t cnt) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < cnt; e += (size_t)gridDim.x * blockDim.x) {
        double s = 0;
        for (int r = 0; r < pp.n; ++r) s += pp.p[r][e];          // rank order: identical bits on every rank
        out[e] = s;
    }
}

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
    std::vector<int> plane_perm, edge_perm;   // sorted index -> caller index
    int n_blocks_sweep = 0, n_blocks_reduce = 0;
    size_t lds_sweep = 0, lds_step = 0, lds_reduce = 0;
    size_t span = 0;               // doubles of one linear-system set (SysBuf::ar): the multi-GPU all-reduce message
    bool step_lds = false;
    int* d_status = nullptr;
    Ctl* h_ctl = nullptr;          // pinned
    char* h_mirror = nullptr; Ctl* d_hctl = nullptr; int* d_hseq = nullptr;      // pinned + mapped: Ctl | sequence word, written by the step kernel (DevP::hctl)
    double* h_pin = nullptr;       // pinned scratch
    char* marg_ws = nullptr;       // device work space of vil_marginalize (grow-only)
    size_t marg_ws_bytes = 0;
    size_t h_pin_bytes = 0;
    std::vector<int> prior_joff;
    ncclComm_t comm = nullptr;     // RCCL communicator over xGMI (world > 1)
    std::shared_ptr<struct LocalComm> lcomm;   // in-process communicator (vil_comm_init_local)
    double* lc_tmp = nullptr; size_t lc_cap = 0;
    bool sharded = false;          // the resident problem is this rank's shard of the factor set
    // hipGraph of a chunk of iterations, reused by repeated solves of one upload (key: chunk length, options)
    struct ChunkGraph { int n; SolveOpts so; hipGraphExec_t exec; };
    std::vector<ChunkGraph> graphs;
    int use_graph = -1;            // VIL_GRAPH=0 disables
    int solve_gen = 0;             // generation counter of the helper-workgroup flags (Ctl::gen)
    int solves_since_upload = 0;
    bool split = false;            // sweep + gather fill set 0, the collective sums it into set 1, the step kernel reads set 1
    bool force_split = false;      // vil_debug_set_split: that plumbing on a single rank
    int last_live = 5;             // live sweep launches of the previous solve (sizes the first launch chunk)
    int lm_b = 0, lm_e = 0;        // owned landmark range
    // ---- window residency across frames (vil_lidar_*, vil_set_gauge_fix, vil_marginalize_resident) --------------------------------
    struct Slab { int np, ne, slot; };
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
    bool profiling = false;
    std::vector<hipEvent_t> ev, ev_mid;
    vil_profile prof = {0, 0.0, 0, 0.0, 0.0};
};

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

__global__ void k_aos2soa(const double* aos, int n, int ncomp, double* dst, size_t stride) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n) for (int q = 0; q < ncomp; ++q) dst[(size_t)q * stride + f] = aos[(size_t)f * ncomp + q];
}
__host__ __device__ static void gauge_fix_core(const double* pose0_before, int K, double* pose, double* speedbias, double* ex_pose);
__host__ __device__ static void gauge_rot(const double* pose0_before, const double* pose0_now, double* rot);
__host__ __device__ static void gauge_frame(const double* rot, const double* p0, const double* pose0_before, double* pp, double* sb);
__host__ __device__ static void gauge_ex(double* ex_pose);
// double2vector()'s gauge fix on the device, both state buffers: one thread per frame (+ one for the extrinsic); every thread derives the
// yaw correction from frame 0 itself, before anybody overwrites it (the frames are independent after that: one lane's latency instead of K)
__global__ void k_gauge_fix(DevP P, const double* x0) {
    const int t = threadIdx.x, K = P.K;
    double* x = P.x[0];
    double rot[9], p0[3];
    gauge_rot(x0 + xo_pose(P, 0), x + xo_pose(P, 0), rot);
    for (int i = 0; i < 3; ++i) p0[i] = x[xo_pose(P, 0) + i];
    __syncthreads();
    if (t < K) gauge_frame(rot, p0, x0 + xo_pose(P, 0), x + xo_pose(P, t), x + xo_sb(P, t));
    else if (t == K) gauge_ex(x + xo_ex(P));
    __syncthreads();
    for (int i = t; i < 16 * K + 8; i += blockDim.x) P.x[1][i] = x[i];
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
    HIPCHK(hipHostMalloc(&c->h_ctl, sizeof(Ctl), hipHostMallocDefault));
    if (!getenv("VIL_NO_POLL") && hipHostMalloc((void**)&c->h_mirror, sizeof(Ctl) + 64, hipHostMallocMapped) == hipSuccess) {
        memset(c->h_mirror, 0, sizeof(Ctl) + 64);
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, c->h_mirror, 0) == hipSuccess) { c->d_hctl = (Ctl*)dp; c->d_hseq = (int*)((char*)dp + sizeof(Ctl)); }
    }
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
    for (auto& g : c->graphs) hipGraphExecDestroy(g.exec);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->ar.d) hipFree(c->ar.d);
    if (c->ar.h) hipHostFree(c->ar.h);
    if (c->up_ev) hipEventDestroy(c->up_ev);
    if (c->dep_ev) hipEventDestroy(c->dep_ev);
    if (c->d_status) hipFree(c->d_status);
    if (c->h_ctl) hipHostFree(c->h_ctl);
    if (c->h_mirror) hipHostFree(c->h_mirror);
    if (c->h_pin) hipHostFree(c->h_pin);
    if (c->marg_ws) hipFree(c->marg_ws);
    if (c->lc_tmp) hipFree(c->lc_tmp);
    if (c->d_pl) hipFree(c->d_pl);
    if (c->d_ed) hipFree(c->d_ed);
    if (c->d_lstage) hipFree(c->d_lstage);
    if (c->h_lstage) hipHostFree(c->h_lstage);
    if (c->lpush_ev) hipEventDestroy(c->lpush_ev);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

static int validate(const vil_problem* p, const vil_state* s, bool device_lidar = false) {
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
    if (p->n_imu > 0 && (!p->imu_i || !p->imu_j || !p->imu_const)) return VIL_ERR_INVALID_ARGUMENT;
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
        if (pr.nblk <= 0 || !pr.blk_kind || !pr.blk_index || !pr.blk_col || !pr.x0 || !pr.J0 || !pr.r0) return VIL_ERR_INVALID_ARGUMENT;
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

// gp / vis_f0 (sharded): the whole window's problem and the index of this rank's first visual factor in it
static int upload_impl(vil_ctx* c, const vil_problem* p, const vil_state* s, bool sharded, const vil_device_lidar* dl = nullptr, const vil_problem* gp = nullptr, int vis_f0 = 0) {
    if (!c) return VIL_ERR_INVALID_ARGUMENT;
    int st = validate(p, s, dl != nullptr);
    if (st != VIL_OK) return st;                 // an invalid problem leaves the resident one untouched
    c->uploaded = false;                         // from here on the arena is rewritten: resident only again after a complete upload
    c->resident_kind = 0;
    HIPCHK(hipSetDevice(c->device));
    const int K = p->K, L = p->L, D = 15 * K + 7, NV = 6 * K + 7, NS = 16 * K + 8 + L;
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
    put(x.data(), sizeof(double) * NS, (void**)&P.x[1]);
    put(x.data(), sizeof(double) * NS, (void**)&c->d_x0);
    // visual
    P.n_vis = p->n_vis; P.vis_stride = (p->n_vis + 31) & ~31;
    {
        if (double* soa = (double*)reserve(8 * (size_t)14 * std::max(P.vis_stride, 1), (void**)&P.vis_c)) {
            for (int q = 0; q < 14; ++q) {
                double* row = soa + (size_t)q * P.vis_stride;
                for (int f = 0; f < p->n_vis; ++f) row[f] = p->vis_const[(size_t)f * 14 + q];
                for (int f = p->n_vis; f < P.vis_stride; ++f) row[f] = 0.0;
            }
        }
        put(p->vis_i, 4 * (size_t)p->n_vis, (void**)&P.vis_i); put(p->vis_j, 4 * (size_t)p->n_vis, (void**)&P.vis_j); put(p->vis_l, 4 * (size_t)p->n_vis, (void**)&P.vis_l);
        std::vector<int> lms(L + 1, 0);
        for (int f = 0; f < p->n_vis; ++f) lms[p->vis_l[f] + 1]++;
        for (int l = 0; l < L; ++l) lms[l + 1] += lms[l];
        put(lms.data(), 4 * (size_t)(L + 1), (void**)&P.lm_start);
        {
            std::vector<int> acol(std::max(L, 1), -1), fcol(std::max(p->n_vis, 1), 0);
            for (int f = p->n_vis - 1; f >= 0; --f) { acol[p->vis_l[f]] = 6 * p->vis_i[f]; fcol[f] = 6 * p->vis_j[f]; }
            put(acol.data(), 4 * acol.size(), (void**)&P.lm_acol); put(fcol.data(), 4 * fcol.size(), (void**)&P.fcol);
        }
        P.vis_f0 = 0;
        if (gp) {                                        // tables of the whole window for the step kernel (every rank walks every landmark)
            std::vector<int> gl(L + 1, 0), gac(std::max(L, 1), -1), gfc(std::max(gp->n_vis, 1), 0);
            for (int f = 0; f < gp->n_vis; ++f) gl[gp->vis_l[f] + 1]++;
            for (int l = 0; l < L; ++l) gl[l + 1] += gl[l];
            for (int f = gp->n_vis - 1; f >= 0; --f) { gac[gp->vis_l[f]] = 6 * gp->vis_i[f]; gfc[f] = 6 * gp->vis_j[f]; }
            put(gl.data(), 4 * gl.size(), (void**)&P.glm_start); put(gac.data(), 4 * gac.size(), (void**)&P.glm_acol); put(gfc.data(), 4 * gfc.size(), (void**)&P.gfcol);
            P.vis_f0 = vis_f0;
        }
        std::vector<int> vch;
        int l0 = 0;
        // a visual workgroup's time grows with the factors of its chunk (rounds of work items), the kernel's with its slowest workgroup:
        // chunks are closed at VIL_VCHUNK_FBAL factors (old landmarks carry up to K - 1 observations, new ones two), a single landmark may exceed it.
        // Every workgroup also writes one partial record of NV (NV + 1) / 2 doubles that k_reduce reads back: measured K = 10 (20 kB records)
        // 32 factors: sweep 24.7 -> 22.5 us, reduce 15.0 -> 15.4; K = 20 (65 kB records): sweep 42 -> 60 us -- so only for the small records
        int fbal = NV <= 80 ? VIL_VCHUNK_FBAL : VIL_VCHUNK_F;
        if (const char* ev = VIL_TUNE_ENV("VIL_VFBAL")) fbal = std::max(1, atoi(ev));
        while (l0 < L) {
            int l1 = l0, nf = 0;
            while (l1 < L && l1 - l0 < VIL_VCHUNK_LM && nf + (lms[l1 + 1] - lms[l1]) <= VIL_VCHUNK_F && (l1 == l0 || nf + (lms[l1 + 1] - lms[l1]) <= fbal)) { nf += lms[l1 + 1] - lms[l1]; ++l1; }
            if (l1 == l0) return VIL_ERR_UNSUPPORTED;   // a single landmark with > VIL_VCHUNK_F observations
            if (nf > 0) { vch.push_back(l0); vch.push_back(l1); }
            l0 = l1;
        }
        P.n_vchunk = (int)vch.size() / 2;
        put(vch.data(), 4 * vch.size(), (void**)&P.vchunk);
        // group the sub-chunks into visual workgroups (each owns one LDS triangle / one partial record)
        int vwg_max = 256;
        if (const char* ev = VIL_TUNE_ENV("VIL_VWG")) vwg_max = std::max(1, atoi(ev));
        P.n_vwg = std::min(P.n_vchunk, vwg_max);
        std::vector<int> vw;
        for (int w = 0; w < P.n_vwg; ++w) { vw.push_back((int)((long long)P.n_vchunk * w / P.n_vwg)); vw.push_back((int)((long long)P.n_vchunk * (w + 1) / P.n_vwg)); }
        put(vw.data(), 4 * vw.size(), (void**)&P.vwg);
        P.NVT = NV * (NV + 1) / 2; P.VP = P.NVT + 3 * NV + 1;
        put(nullptr, 8 * (size_t)std::max(P.n_vwg, 1) * P.VP, (void**)&P.vpart);
    }
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
            if (sharded || (int)c->slabs.size() > K) return VIL_ERR_UNSUPPORTED;
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
        put(nullptr, 8 * (size_t)28 * std::max(P.n_pchunk + P.n_echunk, 1), (void**)&P.lpart);
        put(lcp.data(), 4 * lcp.size(), (void**)&P.lchunk_pose);
    }
    // IMU
    P.n_imu = p->n_imu;
    put(p->imu_const, 8 * (size_t)287 * p->n_imu, (void**)&P.imu_c);
    put(nullptr, 8 * (size_t)225 * std::max(p->n_imu, 1), (void**)&P.imu_U);
    put(p->imu_i, 4 * (size_t)p->n_imu, (void**)&P.imu_i); put(p->imu_j, 4 * (size_t)p->n_imu, (void**)&P.imu_j);
    put(nullptr, 8 * (size_t)931 * std::max(p->n_imu, 1), (void**)&P.ipart);
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
        put(pr.x0, 8 * (size_t)xo, (void**)&P.px0); put(pr.J0, 8 * (size_t)n * n, (void**)&P.pJ0); put(pr.r0, 8 * (size_t)n, (void**)&P.pr0);
        put(nullptr, 8 * (size_t)n * n, (void**)&P.pH); put(nullptr, 8 * (size_t)n, (void**)&P.pg0); put(nullptr, 8, (void**)&P.pc0);
    }
    put(pinv.data(), 4 * (size_t)D, (void**)&P.pinv);
    put(nullptr, 8 * (size_t)((P.pn ? P.pn + 1 : 0) + 601 * (p->n_icp + p->n_lps) + 1), (void**)&P.mpart);
    // ICP / LPS
    P.n_icp = p->n_icp; P.n_lps = p->n_lps;
    put(p->icp_ids, 16 * (size_t)p->n_icp, (void**)&P.icp_ids); put(p->icp_const, 80 * (size_t)p->n_icp, (void**)&P.icp_c);
    put(p->lps_ids, 8 * (size_t)p->n_lps, (void**)&P.lps_ids); put(p->lps_const, 56 * (size_t)p->n_lps, (void**)&P.lps_c);
    // systems + work space (zero-initialised)
    // one contiguous block per set: [S | gred | bc | diag | cost | 2 spare | hll | bl | invp | sl | eA | eO]
    const size_t Lp = (size_t)std::max(L, 1), Fp = (size_t)std::max(gp ? gp->n_vis : p->n_vis, 1);
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
    { const size_t Tm = (size_t)(D + 16) / 16; put(nullptr, 8 * std::max((size_t)D * D, (size_t)TILE_SZ * (Tm * (Tm + 1) / 2)), (void**)&P.M); } put(nullptr, 8 * (size_t)D, (void**)&P.stepc); put(nullptr, 8 * 4 * 16, (void**)&P.hpart); put(nullptr, 4 * 16, (void**)&P.hflag);
    put(nullptr, 8 * 8 * 16 * 8, (void**)&P.hpart2); put(nullptr, 4 * 16 * 8, (void**)&P.hflag2);      // one slot per helper WAVE
    put(nullptr, 16, (void**)&P.xflag); put(nullptr, 16, (void**)&P.xstat);
    put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.la); put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.lb); put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.stepl);
    put(nullptr, 8 * (size_t)D, (void**)&P.tmpc); put(nullptr, 8 * (size_t)std::max(L, 1), (void**)&P.tmpl);
    put(nullptr, sizeof(Ctl), (void**)&P.ctl);
    put(nullptr, 8 * 64, (void**)&P.dbg);
    if (const char* ev = VIL_TUNE_ENV("VIL_SKIP")) P.skip_mask = atoi(ev);
    // helper workgroups of the step kernel: worth it once every master thread would own more than one landmark
    P.n_help = L >= 2 * VIL_STEP_THREADS ? 7 : (L >= VIL_STEP_THREADS ? 3 : 0);
    if (const char* ev = VIL_TUNE_ENV("VIL_HELP")) P.n_help = std::max(0, std::min(15, atoi(ev)));
    // master and helpers wait for one another inside the launch: all of them must be resident at once (vil_coop.hpp).  With its
    // dynamic LDS a step workgroup owns a compute unit; a device with fewer units than 1 + n_help runs without helpers.
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess) cus = 0;
        if (1 + P.n_help > cus / 2) P.n_help = 0;
    }
    // device allocation + single H2D copy
    if (oom) return VIL_ERR_DEVICE;
    const size_t tables = (ar.hsize + 255) & ~size_t(255), total = tables + ((ar.ssize + 255) & ~size_t(255));
    if (total > ar.cap) { if (ar.d) HIPCHK(hipFree(ar.d)); ar.d = nullptr; ar.cap = 0; HIPCHK(hipMalloc(&ar.d, total + total / 4)); ar.cap = total + total / 4; }
    for (const Fix& f : fix) *f.slot = ar.d + (f.scratch ? tables : 0) + f.off;
    if (dl) { P.pl_c = dl->plane_soa; P.pl_stride = dl->plane_stride; P.ed_c = dl->edge_soa; P.ed_stride = dl->edge_stride; }
    if (!gp) { P.glm_start = P.lm_start; P.glm_acol = P.lm_acol; P.gfcol = P.fcol; }
    if (c->lidar_resident) { P.pl_c = c->d_pl; P.pl_stride = c->nslot * c->cap_p; P.ed_c = c->d_ed; P.ed_stride = c->nslot * c->cap_e; }
    {   // what vil_marginalize_resident will need (a few passes over int tables)
        vil_ctx::MargMeta& mm = c->mm;
        mm.has_prior = p->prior.n > 0; mm.prior_kind.clear(); mm.prior_index.clear();
        if (mm.has_prior) { mm.prior_kind.assign(p->prior.blk_kind, p->prior.blk_kind + p->prior.nblk); mm.prior_index.assign(p->prior.blk_index, p->prior.blk_index + p->prior.nblk); }
        mm.obs0.assign(K, 0); mm.n_lm0 = 0; mm.imu01 = false; mm.use_td = p->use_td != 0;
        int last_l = -1;
        for (int f = 0; f < p->n_vis; ++f) if (p->vis_i[f] == 0) { mm.obs0[p->vis_j[f]] = 1; if (p->vis_l[f] != last_l) { ++mm.n_lm0; last_l = p->vis_l[f]; } }
        for (int f = 0; f < p->n_imu; ++f) if (p->imu_i[f] == 0 && p->imu_j[f] == 1 && p->imu_const[(size_t)f * 287 + 16] < 10.0) mm.imu01 = true;
        mm.icp_ids.assign(p->icp_ids, p->icp_ids + 4 * (size_t)p->n_icp); mm.lps_ids.assign(p->lps_ids, p->lps_ids + 2 * (size_t)p->n_lps);
        if (!c->lidar_resident && !dl) {
            for (int f = 0; f < p->n_plane && !mm.lidar0; ++f) if (p->plane_pose[f] == 0) mm.lidar0 = true;
            for (int f = 0; f < p->n_edge && !mm.lidar0; ++f) if (p->edge_pose[f] == 0) mm.lidar0 = true;
        }
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
    c->P = P; c->K = K; c->L = L; c->D = D; c->NS = NS;
    c->P.hctl = c->d_hctl; c->P.hseq = c->d_hseq;
    { const int per = VIL_SWEEP_THREADS / 256; c->n_blocks_sweep = P.n_imu + P.n_vwg + (P.n_pchunk + per - 1) / per + (P.n_echunk + per - 1) / per + 2; }
    c->lds_sweep = sizeof(double) * (size_t)(P.NVT + 3 * NV + VIL_VCHUNK_F * VF_STRIDE + VIL_VCHUNK_LM * 16 + 32 + VIL_VCHUNK_F + 8 + VIL_VCHUNK_LM + 8);
    if (c->lds_sweep < 8 * 2048) c->lds_sweep = 8 * 2048;
    if (c->lds_sweep > 160 * 1024) return VIL_ERR_UNSUPPORTED;
    HIPCHK(hipFuncSetAttribute((const void*)k_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_sweep));
    c->n_blocks_reduce = (D * (D + 1) / 2 + RED_EPW - 1) / RED_EPW + (2 * D + RED_EPW - 1) / RED_EPW + 1;
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
            const size_t fixed = sizeof(vd::StepShared) + 512;
            if (8 * (tiles + wt + scr) + fixed <= 160 * 1024) { P.chain = 1; c->lds_step = 8 * (tiles + wt + scr); }
            else if (8 * (tiles + scr) + fixed <= 160 * 1024) { P.chain = 2; c->lds_step = 8 * (tiles + scr); }
        }
        c->P.chain = P.chain; c->P.chain_rs = P.chain_rs;
    }
    if (P.chain) {
        c->step_lds = true;
        if (P.chain == 1) {
            HIPCHK(hipFuncSetAttribute((const void*)k_step<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_step));
        } else {
            HIPCHK(hipFuncSetAttribute((const void*)k_step<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_step));
        }
    } else {
    { const size_t T = (size_t)(D + 1 + 15) / 16; c->lds_step = 8 * TILE_SZ * (T * (T + 1) / 2); }   // 16x16-tiled (row stride 17) lower storage incl. the rhs row
    c->step_lds = c->lds_step + sizeof(vd::StepShared) + 256 <= 160 * 1024;
    if (c->step_lds) {
        HIPCHK(hipFuncSetAttribute((const void*)k_step<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_step));
    } else {
        // tile array in global memory; LDS stages the active tile column of the factorisation (T tiles)
        const size_t T = (size_t)(D + 1 + 15) / 16;
        c->lds_step = 8 * (size_t)TILE_SZ * T;
        if (c->lds_step + sizeof(vd::StepShared) + 256 > 160 * 1024) return VIL_ERR_UNSUPPORTED;
        HIPCHK(hipFuncSetAttribute((const void*)k_step<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_step));
    }
    }
    // one-time set-up: IMU sqrt-information, prior contraction
    HIPCHK(hipMemsetAsync(c->d_status, 0, sizeof(int), c->stream));
    const int nb_setup = P.n_imu + (P.pn ? 64 : 0);
    if (nb_setup > 0) hipLaunchKernelGGL(k_setup, dim3(nb_setup), dim3(VIL_THREADS), 0, c->stream, P, const_cast<double*>(P.imu_U), c->d_status);
    int hstat = 0;
    HIPCHK(hipMemcpyAsync(&hstat, c->d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    if (hstat != 0) return VIL_ERR_NOT_POSITIVE_DEFINITE;
    c->uploaded = true; c->resident_kind = 2;    // vil_upload / vil_solve promote it to 1
    return VIL_OK;
}

static int comm_agree(vil_ctx* c, int st);

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
    q.n_edge = ee - eb; q.edge_pose = p->edge_pose + eb; q.edge_const = p->edge_const + (size_t)eb * 9;
    q.n_plane = pe - pb; q.plane_pose = p->plane_pose + pb; q.plane_const = p->plane_const + (size_t)pb * 7;
    if (c->rank != 0) { q.n_imu = 0; q.n_icp = 0; q.n_lps = 0; q.prior.n = 0; q.prior.nblk = 0; }
    c->lm_b = lb; c->lm_e = le;
    st = comm_agree(c, upload_impl(c, &q, s, true, nullptr, p, f0));       // e.g. rank 0's IMU set-up failed: every rank reports it
    if (st != VIL_OK) c->uploaded = false;
    return st;
}

int vil_upload(vil_ctx* c, const vil_problem* p, const vil_state* s) {
    if (!c) return VIL_ERR_INVALID_ARGUMENT;
    const int st = (c->world > 1 && (c->comm || c->lcomm)) ? upload_sharded(c, p, s)      // a world > 1 context without a communicator works un-sharded
                                                           : upload_impl(c, p, s, false);
    if (st == VIL_OK) c->resident_kind = 1;
    return st;
}

// sum over the ranks of the communicator, in stream order: recv = sum of every rank's send (recv may be send)
static int all_reduce2(vil_ctx* c, const double* send, double* recv, size_t cnt) {
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
        hipLaunchKernelGGL(k_sum_peers, dim3((unsigned)std::min<size_t>(256, (cnt + 255) / 256)), dim3(256), 0, c->stream, inplace ? c->lc_tmp : recv, pp, cnt);
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
        hipLaunchKernelGGL(k_sum_peers, dim3((unsigned)std::min<size_t>(256, (cnt + 255) / 256)), dim3(256), 0, c->stream, c->lc_tmp, pp, cnt);
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
    if (c->comm) {
        int h = st;
        if (hipMemcpyAsync(c->d_status, &h, sizeof(int), hipMemcpyHostToDevice, c->stream) != hipSuccess) return VIL_ERR_DEVICE;
        if (g_rccl.AllReduce(c->d_status, c->d_status, 1, ncclInt, ncclMin, c->comm, c->stream) != ncclSuccess) return VIL_ERR_COMM;
        if (hipMemcpyAsync(&h, c->d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return VIL_ERR_DEVICE;
        if (hipStreamSynchronize(c->stream) != hipSuccess) return VIL_ERR_DEVICE;
        return h;
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
    hipLaunchKernelGGL(k_sweep, dim3(c->n_blocks_sweep), dim3(VIL_SWEEP_THREADS), c->lds_sweep, c->stream, view(c, 0), so);
    return VIL_OK;
}
static int launch_reduce_step(vil_ctx* c, const SolveOpts& so, bool step, hipEvent_t ev_mid = nullptr) {
    hipLaunchKernelGGL(k_reduce, dim3(c->n_blocks_reduce), dim3(VIL_THREADS), 0, c->stream, view(c, 0));
    if (ev_mid) hipEventRecord(ev_mid, c->stream);
    if (c->split) {                                    // the one collective of the iteration
        const int st = all_reduce2(c, c->P.sys[0].ar, c->P.sys[1].ar, c->span);
        if (st != VIL_OK) return st;
    }
    if (!step) return VIL_OK;
    const DevP Ps = view(c, 1);
    const dim3 g(1 + c->P.n_help), b(VIL_STEP_THREADS);
    if (c->P.chain == 1) hipLaunchKernelGGL((k_step<true, 1>), g, b, c->lds_step, c->stream, Ps, so);
    else if (c->P.chain == 2) hipLaunchKernelGGL((k_step<true, 2>), g, b, c->lds_step, c->stream, Ps, so);
    else if (c->step_lds) hipLaunchKernelGGL((k_step<true, 0>), g, b, c->lds_step, c->stream, Ps, so);
    else hipLaunchKernelGGL((k_step<false, 0>), g, b, c->lds_step, c->stream, Ps, so);
    return VIL_OK;
}

static int init_ctl(vil_ctx* c, const vil_options* o, int lin_mode) {
    Ctl ctl; memset(&ctl, 0, sizeof ctl);
    ctl.gen = ++c->solve_gen; ctl.cur = 0; ctl.first = 1; ctl.radius = o->initial_radius; ctl.mu = o->min_mu; ctl.lin_mode = lin_mode;
    *c->h_ctl = ctl;
    HIPCHK(hipMemcpyAsync(c->P.ctl, c->h_ctl, sizeof(Ctl), hipMemcpyHostToDevice, c->stream));
    return VIL_OK;
}

int vil_profile_enable(vil_ctx* c, int on) {
    if (!c) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    if (on && c->ev.empty()) { c->ev.resize(40); for (auto& e : c->ev) HIPCHK(hipEventCreate(&e)); c->ev_mid.resize(20); for (auto& e : c->ev_mid) HIPCHK(hipEventCreate(&e)); }
    c->profiling = on != 0;
    return VIL_OK;
}
int vil_profile_read(vil_ctx* c, vil_profile* out, int reset) {
    if (!c || !out) return VIL_ERR_INVALID_ARGUMENT;
    *out = c->prof;
    if (reset) c->prof = vil_profile{0, 0.0, 0, 0.0, 0.0};
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
    HIPCHK(hipMemcpyAsync(c->P.x[0], c->d_x0, 8 * (size_t)c->NS, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->P.x[1], c->d_x0, 8 * (size_t)c->NS, hipMemcpyDeviceToDevice, c->stream));
    return VIL_OK;
}

int vil_solve_resident(vil_ctx* c, const vil_options* o, vil_summary* sum) {
    if (!c || !o || !sum || !c->uploaded || c->resident_kind != 1) return VIL_ERR_INVALID_ARGUMENT;   // vil_marginalize / vil_eval_factors / vil_linearize replaced the window
    if (o->precision != 0 && o->precision != 1) return VIL_ERR_UNSUPPORTED;
    HIPCHK(hipSetDevice(c->device));
    const auto t0 = std::chrono::steady_clock::now();
    const SolveOpts so = to_dev_opts(o);
    int st = init_ctl(c, o, 0);
    if (st != VIL_OK) return st;
    // every iteration = one sweep + one step kernel; `done` turns the tail into no-ops
    bool finished = false, polled_done = false;
    // iterations are enqueued in chunks without host round trips; the first chunk is sized by the previous solve of
    // this context (consecutive windows of a tracker need similar iteration counts), later chunks are short
    int chunk = std::min(15, std::max(3, c->last_live));
    for (int it = 0; it <= o->max_iterations + 8 && !finished; chunk = 3) {
        int launched = 0;
        const int sweeps_before = (it == 0) ? 0 : c->h_ctl->n_sweeps;
        // Repeated solves of ONE upload (bench, re-solves after a rejected frame) replay a captured hipGraph of the chunk:
        // ~2 % less inter-kernel gap.  The first solve of an upload launches directly -- capturing costs more than it saves.
        if (c->use_graph < 0) { const char* ev = VIL_TUNE_ENV("VIL_GRAPH"); c->use_graph = ev ? atoi(ev) : 1; }
        const int nthis = std::min(chunk, o->max_iterations + 9 - it);
        if (c->use_graph && c->solves_since_upload > 0 && !c->profiling && !c->split && nthis > 0) {     // (the local communicator's host barriers cannot be captured)
            hipGraphExec_t exec = nullptr;
            for (auto& g : c->graphs) if (g.n == nthis && memcmp(&g.so, &so, sizeof so) == 0) exec = g.exec;
            if (!exec) {
                hipGraph_t graph = nullptr;
                HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                for (int q = 0; q < nthis; ++q) { launch_sweep(c, so); launch_reduce_step(c, so, true, nullptr); }
                HIPCHK(hipStreamEndCapture(c->stream, &graph));
                HIPCHK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
                hipGraphDestroy(graph);
                c->graphs.push_back({nthis, so, exec});
            }
            HIPCHK(hipGraphLaunch(exec, c->stream));
            it += nthis; launched = nthis;
        } else
        for (int q = 0; q < chunk && it <= o->max_iterations + 8; ++q, ++it, ++launched) {
            if (c->profiling) HIPCHK(hipEventRecord(c->ev[2 * q], c->stream));
            launch_sweep(c, so);
            if (c->profiling) HIPCHK(hipEventRecord(c->ev[2 * q + 1], c->stream));
            st = launch_reduce_step(c, so, true, c->profiling ? c->ev_mid[q] : nullptr);
            if (st != VIL_OK) return st;          // a failed collective fails on every rank (all_reduce): nobody is left waiting
        }
        if (c->profiling) HIPCHK(hipEventRecord(c->ev[2 * launched], c->stream));
        // The step kernel that finishes the solve leaves Ctl + the solve generation in pinned host memory: poll that word instead of
        // synchronising (a few microseconds earlier, and the no-op tail of the chunk is not waited for).  A chunk that runs out without
        // finishing is seen by hipStreamQuery and takes the copy + synchronise route, as do profiling and multi-rank solves.
        bool polled = false;
        if (c->d_hseq && !c->profiling && !c->split) {
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
                HIPCHK(hipEventElapsedTime(&ms, c->ev[2 * q], c->ev[2 * q + 1])); c->prof.sweep_ms += ms; c->prof.sweep_launches++;
                HIPCHK(hipEventElapsedTime(&ms, c->ev[2 * q + 1], c->ev[2 * q + 2])); c->prof.step_ms += ms; c->prof.step_launches++;
                HIPCHK(hipEventElapsedTime(&ms, c->ev[2 * q + 1], c->ev_mid[q])); c->prof.reduce_ms += ms;
            }
        }
        if (!finished && o->max_time_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() >= o->max_time_s) {
            c->h_ctl->done = 1; c->h_ctl->term = VIL_TERM_MAX_TIME; finished = true;   // ceres max_solver_time_in_seconds (estimator.cpp:1411)
        }
    }
    HIPCHK(hipGetLastError());
    c->solves_since_upload++;
    const Ctl& ctl = *c->h_ctl;
    c->last_live = ctl.n_sweeps;
    memset(sum, 0, sizeof *sum);
    sum->iterations = ctl.iter; sum->successful_steps = ctl.nsucc; sum->termination = ctl.term;
    sum->initial_cost = ctl.initial_cost; sum->final_cost = ctl.cost_cur;
    for (int i = 0; i < VIL_MAX_TRACE; ++i) { sum->cost_trace[i] = ctl.cost_trace[i]; sum->radius_trace[i] = ctl.radius_trace[i]; }
    sum->t_solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    // the accepted state is x[cur]; keep x[0] as "the" resident state
    if (ctl.cur != 0) HIPCHK(hipMemcpyAsync(c->P.x[0], c->P.x[1], 8 * (size_t)c->NS, hipMemcpyDeviceToDevice, c->stream));
    else HIPCHK(hipMemcpyAsync(c->P.x[1], c->P.x[0], 8 * (size_t)c->NS, hipMemcpyDeviceToDevice, c->stream));
    if (c->gauge_on && finished && c->h_ctl->status == 0) hipLaunchKernelGGL(k_gauge_fix, dim3(1), dim3(64), 0, c->stream, c->P, c->d_x0);      // estimator.cpp:960-1011 before the read-back
    // (whoever reads the state back -- vil_download_state, vil_marginalize_resident -- orders itself behind these on the stream and
    //  synchronises for its own copy; a solve whose end was polled does not wait for its no-op tail here)
    if (!polled_done) HIPCHK(hipStreamSynchronize(c->stream));
    if (!finished) return VIL_ERR_DEVICE;
    if (ctl.status != 0) return ctl.status;
    if (!std::isfinite(ctl.cost_cur)) return VIL_ERR_NON_FINITE;
    return VIL_OK;
}

int vil_download_state(vil_ctx* c, vil_state* s) {
    if (!c || !s || !c->uploaded || s->K != c->K || s->L != c->L) return VIL_ERR_INVALID_ARGUMENT;
    HIPCHK(hipSetDevice(c->device));
    std::vector<double> x(c->NS);
    HIPCHK(hipMemcpyAsync(x.data(), c->P.x[0], 8 * (size_t)c->NS, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (double v : x) if (!std::isfinite(v)) return VIL_ERR_NON_FINITE;
    const int K = c->K, L = c->L;
    memcpy(s->pose, &x[0], 8 * (size_t)7 * K); memcpy(s->speedbias, &x[7 * K], 8 * (size_t)9 * K);
    memcpy(s->ex_pose, &x[16 * K], 56); s->td[0] = x[16 * K + 7];
    if (L) memcpy(s->inv_depth, &x[16 * K + 8], 8 * (size_t)L);
    return VIL_OK;
}

int vil_solve(vil_ctx* c, const vil_problem* p, vil_state* s, const vil_options* o, vil_summary* sum) {
    if (!c || !p || !s || !o || !sum) return VIL_ERR_INVALID_ARGUMENT;
    const auto t0 = std::chrono::steady_clock::now();
    int st = vil_upload(c, p, s);
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
    if (c->world > 1 && (c->comm || c->lcomm)) return VIL_ERR_UNSUPPORTED;
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
                       const bool ex_t, const bool td_t, const bool use_td, const int n_lm_elim, const vil_state* s, vil_prior_out* out) {
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
    const size_t bytes = 8 * ((size_t)nd * nd * 2 + nd + nn * 5 + (size_t)n * 3) + 4 * (size_t)(nd + n) + 8192;
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
    if (off > bytes) return VIL_ERR_DEVICE;
    std::vector<int> cols(drop_cols.begin(), drop_cols.end()); cols.insert(cols.end(), keep_cols.begin(), keep_cols.end());      // (alive until the stream has been synchronised below)
    HIPCHK(hipMemcpyAsync(d_drop, cols.data(), 4 * cols.size(), hipMemcpyHostToDevice, c->stream));
    {
        const size_t a_bytes = 8 * nn, cap = 156 * 1024;
        if (a_bytes > cap) return VIL_ERR_UNSUPPORTED;
        const size_t dyn = std::max<size_t>(4096, a_bytes);
        if (dyn > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k_marg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL(k_marg, dim3(1), dim3(MARG_THREADS), dyn, c->stream, M, 0);
        if (!VIL_TUNE_ENV("VIL_MARG_PIVOTED")) {               // un-pivoted factorisation on the matrix cores first; pivoted fallback below
            const size_t Tm = (size_t)(n + 1 + 15) / 16, tb = 8 * (size_t)TILE_SZ * (Tm * (Tm + 1) / 2);
            if (tb + sizeof(vd::StepShared) + 512 <= 160 * 1024) {
                if (tb > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)k_marg_fast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tb));
                hipLaunchKernelGGL(k_marg_fast, dim3(1), dim3(VIL_STEP_THREADS), tb, c->stream, M);
            }
        }
        hipLaunchKernelGGL(k_marg, dim3(1), dim3(MARG_THREADS), dyn, c->stream, M, 1);
    }
    const size_t ncam = 16 * (size_t)K + 8;
    st = ensure_pin(c, 8 * (nn * 2 + 2 * (size_t)n + ncam));
    if (st != VIL_OK) return st;
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
    int st = upload_impl(c, &q, s, false);
    if (st != VIL_OK) return st;
    const SolveOpts so = to_dev_opts(o);
    st = init_ctl(c, o, 2);
    if (st != VIL_OK) return st;
    launch_sweep(c, so);
    launch_reduce_step(c, so, false);
    return marg_finish(c, K, old_, drop_pose, pose_t, sb_t, ex_t, td_t, p->use_td != 0, n_lm_elim, s, out);
}

// estimator.cpp:960-1011 double2vector(): yaw + translation gauge fix.  One arithmetic for the host entry point (vil_gauge_fix)
// and the device kernel (vil_set_gauge_fix): pose K x 7 [p q(xyzw)], speed-bias K x 9, ex 7.
__host__ __device__ static void gauge_q2R(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
__host__ __device__ static void gauge_R2ypr(const double* R, double* ypr) {
    const double y = atan2(R[3], R[0]);
    const double pch = atan2(-R[6], R[0] * cos(y) + R[3] * sin(y));
    const double rl = atan2(R[2] * sin(y) - R[5] * cos(y), -R[1] * sin(y) + R[4] * cos(y));
    ypr[0] = y / M_PI * 180.0; ypr[1] = pch / M_PI * 180.0; ypr[2] = rl / M_PI * 180.0;
}
__host__ __device__ static void gauge_R2q(const double* R, double* q /*xyzw*/) {
    double t = R[0] + R[4] + R[8];
    if (t > 0) { t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t; }
    else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * t; q[j] = (R[3 * j + i] + R[3 * i + j]) * t; q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
    }
}
__host__ __device__ static void gauge_rot(const double* pose0_before, const double* pose0_now, double* rot) {
    double R0[9], R00[9], a0[3], a00[3];
    gauge_q2R(pose0_before + 3, R0); gauge_q2R(pose0_now + 3, R00);
    gauge_R2ypr(R0, a0); gauge_R2ypr(R00, a00);
    const double yd = (a0[0] - a00[0]) / 180.0 * M_PI;
    rot[0] = cos(yd); rot[1] = -sin(yd); rot[2] = 0; rot[3] = sin(yd); rot[4] = cos(yd); rot[5] = 0; rot[6] = 0; rot[7] = 0; rot[8] = 1;
    if (fabs(fabs(a0[1]) - 90) < 1.0 || fabs(fabs(a00[1]) - 90) < 1.0)
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double v = 0; for (int k = 0; k < 3; ++k) v += R0[3 * i + k] * R00[3 * j + k]; rot[3 * i + j] = v; }
}
__host__ __device__ static void gauge_frame(const double* rot, const double* p0, const double* pose0_before, double* pp, double* sb) {
    {
        double qn[4], Rf[9], Rn[9], d[3], Pn[3], V[3];
        { const double n = sqrt(pp[3] * pp[3] + pp[4] * pp[4] + pp[5] * pp[5] + pp[6] * pp[6]); for (int i = 0; i < 4; ++i) qn[i] = pp[3 + i] / n; }
        gauge_q2R(qn, Rf);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double v = 0; for (int k = 0; k < 3; ++k) v += rot[3 * i + k] * Rf[3 * k + j]; Rn[3 * i + j] = v; }
        for (int i = 0; i < 3; ++i) d[i] = pp[i] - p0[i];
        for (int i = 0; i < 3; ++i) Pn[i] = rot[3 * i] * d[0] + rot[3 * i + 1] * d[1] + rot[3 * i + 2] * d[2] + pose0_before[i];
        gauge_R2q(Rn, pp + 3); pp[0] = Pn[0]; pp[1] = Pn[1]; pp[2] = Pn[2];
        for (int i = 0; i < 3; ++i) V[i] = rot[3 * i] * sb[0] + rot[3 * i + 1] * sb[1] + rot[3 * i + 2] * sb[2];
        sb[0] = V[0]; sb[1] = V[1]; sb[2] = V[2];
    }
}
__host__ __device__ static void gauge_ex(double* ex_pose) {
    double qe[4], Re[9];
    { const double n = sqrt(ex_pose[3] * ex_pose[3] + ex_pose[4] * ex_pose[4] + ex_pose[5] * ex_pose[5] + ex_pose[6] * ex_pose[6]); for (int i = 0; i < 4; ++i) qe[i] = ex_pose[3 + i] / n; }
    gauge_q2R(qe, Re); gauge_R2q(Re, ex_pose + 3);
}
__host__ __device__ static void gauge_fix_core(const double* pose0_before, int K, double* pose, double* speedbias, double* ex_pose) {
    double rot[9];
    gauge_rot(pose0_before, pose, rot);
    const double p0[3] = {pose[0], pose[1], pose[2]};
    for (int f = 0; f < K; ++f) gauge_frame(rot, p0, pose0_before, pose + 7 * f, speedbias + 9 * f);
    gauge_ex(ex_pose);
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
    auto cut = [&](int r) {   // first landmark whose factor prefix reaches r/world of the total
        if (r <= 0) return 0; if (r >= world) return p->L;
        const long long target = (long long)p->n_vis * r / world;
        return (int)(std::lower_bound(lms.begin(), lms.end(), (int)target) - lms.begin());
    };
    if (lm_begin) *lm_begin = std::min(cut(rank), p->L);
    if (lm_end) *lm_end = std::min(cut(rank + 1), p->L);
    if (edge_begin) *edge_begin = (int)((long long)p->n_edge * rank / world);
    if (edge_end) *edge_end = (int)((long long)p->n_edge * (rank + 1) / world);
    if (plane_begin) *plane_begin = (int)((long long)p->n_plane * rank / world);
    if (plane_end) *plane_end = (int)((long long)p->n_plane * (rank + 1) / world);
    return VIL_OK;
}

// ---- window residency across frames (include/vilsolve.h) ------------------------------------------------------------------------
int vil_debug_set_split(vil_ctx* c, int32_t on) { if (!c) return VIL_ERR_INVALID_ARGUMENT; c->force_split = on != 0; c->uploaded = false; c->resident_kind = 0; return VIL_OK; }
int vil_set_gauge_fix(vil_ctx* c, int32_t on) { if (!c) return VIL_ERR_INVALID_ARGUMENT; c->gauge_on = on != 0; return VIL_OK; }

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
    c->slabs.push_back({n_plane, n_edge, slot});
    if (c->lidar_resident) { c->uploaded = false; c->resident_kind = 0; }      // (a recycled slot would feed this frame's points to the old pose)
    return VIL_OK;
}

int vil_marginalize_resident(vil_ctx* c, const vil_state* s, const vil_options* o, const vil_marg_spec* spec, vil_prior_out* out) {
    if (!c || !s || !o || !spec || !out || !c->uploaded || c->resident_kind != 1) return VIL_ERR_INVALID_ARGUMENT;
    if (c->sharded) return VIL_ERR_UNSUPPORTED;
    const int K = c->K;
    if (K < 3 || s->K != K || s->L != c->L) return VIL_ERR_INVALID_ARGUMENT;
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
        if (!old_ && !has_drop) { out->n = -1; return VIL_OK; }     // estimator.cpp:1620-1621: prior kept as is
    } else if (!old_) { out->n = -1; return VIL_OK; }
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
    launch_reduce_step(c, so, false);
    c->P = keep;
    c->resident_kind = 2;                              // the work space now holds the marginalisation's linearisation
    return marg_finish(c, K, old_, drop_pose, pose_t, sb_t, ex_t, td_t, mm.use_td, n_lm_elim, s, out);
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
        c->lcomm = lc; c->rank = r; c->world = n; c->uploaded = false;
    }
    return VIL_OK;
}

}  // extern "C"
