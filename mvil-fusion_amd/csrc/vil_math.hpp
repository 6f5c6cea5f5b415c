// Device-side small linear algebra for the gfx950 kernels (fp64, registers only).
// Conventions follow the reference: Hamilton quaternions stored [x y z w] in parameter blocks
// (estimator.cpp:920-927), R(q) rotates body->world, right-multiplicative tangent
// (pose_local_parameterization.cpp:3-18).
#pragma once
#include <hip/hip_runtime.h>

// The thread index of the roles.  VIL_OPAQUE_TID (the translation unit of the persistent solve, vilpersist.hip): an OPAQUE value -- what a role derives from it is
// computed where the role runs.  The roles are inlined into k_solve's iteration loop; from a plain threadIdx.x LLVM's loop-invariant code motion hoists every
// per-thread address of every role in front of the loop and the register allocator spills them (470 dwords per lane, measured; 20 with the opaque index).
// The other kernels keep the plain index (k_iter with the opaque one: 9 dwords of scratch, from none).
#ifdef VIL_OPAQUE_TID
__device__ __forceinline__ int vil_tid() { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; }
#else
__device__ __forceinline__ int vil_tid() { return (int)threadIdx.x; }
#endif
namespace vd {

// T = double everywhere on the reference path; T = float only for the fp32-evaluation mode of the bulk factor classes
template <class T> struct V3T { T x, y, z; };
using V3 = V3T<double>;
template <class T> __device__ __forceinline__ V3T<T> operator+(V3T<T> a, V3T<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> __device__ __forceinline__ V3T<T> operator-(V3T<T> a, V3T<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> __device__ __forceinline__ V3T<T> operator*(T s, V3T<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class T> __device__ __forceinline__ V3T<T> cross(V3T<T> a, V3T<T> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <class T> __device__ __forceinline__ T dot(V3T<T> a, V3T<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// row-major 3x3 in registers
template <class T> struct M3T { T m[9]; };
using M3 = M3T<double>;
template <class T> __device__ __forceinline__ V3T<T> mul(const M3T<T>& A, V3T<T> v) {
    return {A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z, A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z};
}
template <class T> __device__ __forceinline__ V3T<T> mulT(const M3T<T>& A, V3T<T> v) {  // A^T v
    return {A.m[0] * v.x + A.m[3] * v.y + A.m[6] * v.z, A.m[1] * v.x + A.m[4] * v.y + A.m[7] * v.z, A.m[2] * v.x + A.m[5] * v.y + A.m[8] * v.z};
}
// row vector times matrix: (v^T A)^T == A^T v
template <class T> __device__ __forceinline__ V3T<T> rowmul(V3T<T> v, const M3T<T>& A) { return mulT(A, v); }
template <class T> __device__ __forceinline__ V3T<T> rowmulT(V3T<T> v, const M3T<T>& A) { return mul(A, v); }  // v^T A^T
template <class T> __device__ __forceinline__ V3T<T> v3cast(V3 v) { return {(T)v.x, (T)v.y, (T)v.z}; }
template <class T> __device__ __forceinline__ M3T<T> m3cast(const M3& A) { M3T<T> r; for (int i = 0; i < 9; ++i) r.m[i] = (T)A.m[i]; return r; }

// unit quaternion [x y z w] -> rotation matrix
__device__ __forceinline__ M3 quatR(const double* q) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    M3 R;
    R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz; R.m[2] = txz + twy;
    R.m[3] = txy + twz; R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
    R.m[6] = txz - twy; R.m[7] = tyz + twx; R.m[8] = 1 - (txx + tyy);
    return R;
}
__device__ __forceinline__ M3 loadM3(const double* p) { M3 R; for (int i = 0; i < 9; ++i) R.m[i] = p[i]; return R; }

// quaternions as (w, x, y, z) value structs for the IMU / prior algebra
struct Q4 { double w, x, y, z; };
__device__ __forceinline__ Q4 qload(const double* p /*x y z w*/) { return {p[3], p[0], p[1], p[2]}; }
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ Q4 qinv(Q4 q) {  // conjugate / squared norm (Eigen::Quaternion::inverse)
    const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    const double s = 1.0 / n2;
    return {q.w * s, -q.x * s, -q.y * s, -q.z * s};
}
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {
    V3 u{q.x, q.y, q.z};
    V3 uv = cross(u, v);
    uv = uv + uv;
    return v + q.w * uv + cross(u, uv);
}

// wave64 butterfly sum: every lane ends with the total
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// wave64 sum on the DPP crossbar (no LDS traffic, a few cycles per step instead of a ds_bpermute round trip): rotate-and-add
// inside each row of 16 lanes, then the two row broadcasts carry the row totals to lane 63, which is read back as a
// wave-uniform value.  Different summation order from wave_sum (same value to rounding).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false), h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(h2, l2);
}
// the same fold, result left in lane 63 only (no scalar registers involved)
__device__ __forceinline__ double wave_total_l63(double v) {
    v = dpp_add<0x128, 0xf>(v); v = dpp_add<0x124, 0xf>(v); v = dpp_add<0x122, 0xf>(v); v = dpp_add<0x121, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    return dpp_add<0x143, 0xc>(v);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max(double v) {          // v >= 0 (a zero from a masked-off row never wins)
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false), h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return fmax(v, __hiloint2double(h2, l2));
}
__device__ __forceinline__ double wave_max_l63(double v) {     // non-negative inputs
    v = dpp_max<0x128, 0xf>(v); v = dpp_max<0x124, 0xf>(v); v = dpp_max<0x122, 0xf>(v); v = dpp_max<0x121, 0xf>(v);
    v = dpp_max<0x142, 0xa>(v);
    return dpp_max<0x143, 0xc>(v);
}
// maxima of arbitrary-sign values, result wave-uniform: a row that a broadcast step does not address keeps its own value
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fmax(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int l2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false), h2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return fmax(v, __hiloint2double(h2, l2));
}
__device__ __forceinline__ double wave_fmax_all(double v) {
    v = dpp_fmax<0x128, 0xf>(v); v = dpp_fmax<0x124, 0xf>(v); v = dpp_fmax<0x122, 0xf>(v); v = dpp_fmax<0x121, 0xf>(v);
    v = dpp_fmax<0x142, 0xa>(v); v = dpp_fmax<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_imax(int v) { return max(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false)); }
__device__ __forceinline__ int wave_imax_all(int v) {
    v = dpp_imax<0x128, 0xf>(v); v = dpp_imax<0x124, 0xf>(v); v = dpp_imax<0x122, 0xf>(v); v = dpp_imax<0x121, 0xf>(v);
    v = dpp_imax<0x142, 0xa>(v); v = dpp_imax<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double wave_total(double v) {
    v = dpp_add<0x128, 0xf>(v); v = dpp_add<0x124, 0xf>(v); v = dpp_add<0x122, 0xf>(v); v = dpp_add<0x121, 0xf>(v);   // row_ror 8, 4, 2, 1
    v = dpp_add<0x142, 0xa>(v);                                                                                           // row_bcast15 into rows 1, 3
    v = dpp_add<0x143, 0xc>(v);                                                                                           // row_bcast31 into rows 2, 3
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Fold of N <= 32 per-lane values over the wave at once.  A value-by-value fold spends 6 steps on each of the N registers; here
// every exchange step also halves the number of live registers: lane pairs (i, i ^ m), m = 15, 7, 3, 1 (row_mirror,
// row_half_mirror and two quad permutations: a basis of the 16 lanes of a row), split the register pairs between them -- the
// lane whose distinguishing bit is 0 keeps collecting the even register, its partner the odd one.  After the four steps lane j of
// a row holds the row totals of values 16 s + rev4(j) (s = 0, 1; rev4 = the four lane bits reversed) in two registers; two
// butterflies across the rows (the only LDS-crossbar traffic: 8 ds_bpermute) leave the wave totals in every row:
// out[s] on lane 16 r + j  =  total of value 16 s + rev4(j).  ~230 instructions for 28 values instead of ~560.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false));
}
template <int N, int M, int CTRL>
__device__ __forceinline__ void fold_step(double* r, bool bit) {          // N live registers -> (N + 1) / 2
#pragma unroll
    for (int k = 0; k < (N + 1) / 2; ++k) {
        const double a = r[2 * k], b = (2 * k + 1 < N) ? r[2 * k + 1] : 0.0;
        const double keep = bit ? b : a, send = bit ? a : b;
        r[k] = keep + dpp_move<CTRL>(send);
    }
}
__device__ __forceinline__ int fold_slot(int lane) { const int j = lane & 15; return ((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3); }   // rev4
template <int N>
__device__ __forceinline__ void wave_fold(const double* v, double& out0, double& out1) {
    static_assert(N >= 1 && N <= 32, "wave_fold: at most 32 values");
    const int lane = threadIdx.x & 63;
    double r[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) r[q] = q < N ? v[q] : 0.0;
    fold_step<32, 15, 0x140>(r, (lane & 8) != 0);          // row_mirror
    fold_step<16, 7, 0x141>(r, (lane & 4) != 0);           // row_half_mirror
    fold_step<8, 3, 0x1B>(r, (lane & 2) != 0);             // quad_perm [3 2 1 0]
    fold_step<4, 1, 0xB1>(r, (lane & 1) != 0);             // quad_perm [1 0 3 2]
    // the four rows hold lane-wise partials: a lane-wise exchange across rows is not a DPP pattern, two butterflies through the LDS crossbar
    out0 = r[0]; out1 = r[1];
    out0 += __shfl_xor(out0, 16, 64); out1 += __shfl_xor(out1, 16, 64);
    out0 += __shfl_xor(out0, 32, 64); out1 += __shfl_xor(out1, 32, 64);
}

// hardware reciprocal (square root) estimate + two Newton steps: 1-2 ulp, a fraction of the IEEE divide / sqrt latency
__device__ __forceinline__ double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}
__device__ __forceinline__ double rcp_nr(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}


// robust loss: rho(s), rho'(s) for ceres CauchyLoss(a)/HuberLoss(a). Both have rho'' <= 0, so the
// corrector of marginalization_factor.cpp:37-67 reduces to r <- sqrt(rho') r, J <- sqrt(rho') J.
__device__ __forceinline__ void loss_eval(int kind, double a, double s, double& rho, double& rho1) {
    if (kind == 1) {
        const double b = a * a, c = 1.0 / b, sum = 1.0 + s * c;
        rho = b * log(sum); rho1 = 1.0 / sum;
    } else if (kind == 2) {
        const double b = a * a;
        if (s > b) { const double r = sqrt(s); rho = 2.0 * a * r - b; rho1 = a / r; }
        else { rho = s; rho1 = 1.0; }
    } else { rho = s; rho1 = 1.0; }
}

typedef double d4 __attribute__((ext_vector_type(4)));      // accumulator fragment of v_mfma_f64_16x16x4_f64

// sqrt(x) and 1/sqrt(x) together: hardware rsq seed + two coupled Newton steps (no divide on the pivot chain)
__device__ __forceinline__ void sqrt_rsqrt(double x, double& sq, double& rs) {
    // v_rsq_f64 seeds ~2^-26; one coupled Newton step squares that (the pivot chain is latency-bound: every
    // dependent fp64 op costs ~32 cycles), a residual correction on sqrt keeps it within an ulp or two
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double e = fma(-h, g, 0.5);
    g = fma(g, e, g); h = fma(h, e, h);
    sq = fma(fma(-g, g, x), h, g); rs = h + h;
}
// The same pair with the reciprocal root on the shortest dependent chain -- rsq, y * y, fma, fma (x / 2 forms beside the rsq) -- for pivot
// chains where the next pivot waits for 1 / sqrt only (chol_lookahead); sqrt follows off the chain with one residual correction.
__device__ __forceinline__ void rsqrt_sqrt(double x, double& sq, double& rs) {
    const double y = __builtin_amdgcn_rsq(x), xh = 0.5 * x;
    const double e = fma(-xh, y * y, 0.5);
    rs = fma(y, e, y);
    const double g = x * rs;
    sq = fma(fma(-g, g, x), 0.5 * rs, g);
}
// 1 / sqrt(x) alone (a factor whose diagonal nobody reads -- the solves use the reciprocal pivots): rsq seed, ONE Newton step like rsqrt_sqrt (rsqrt_nr above takes two); five instructions
// instead of the nine of the pair above (fp64 vector instructions issue at half rate on gfx950: the pivot code is bound by their count)
__device__ __forceinline__ double rsqrt_1(double x) {
    const double y = __builtin_amdgcn_rsq(x), xh = 0.5 * x;
    return fma(y, fma(-xh, y * y, 0.5), y);
}
// Workgroup barrier that orders LDS traffic only and leaves global loads in flight.  hipcc 7.2 lowers __syncthreads() on gfx950 to the same two
// instructions (a workgroup lives on one CU, so its workgroup-scope fence needs no vmcnt wait -- checked on a test kernel); it is written out where a
// prefetch RELIES on landing behind the barriers of an LDS-only phase, so that this does not hang on the toolchain's lowering of the fence.
// (Registers loaded from global memory are waited for at their first use.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Data that crosses workgroups INSIDE one launch is stored and loaded at agent scope -- the level the eight XCDs share -- so that a flag only
// has to be ordered behind the poster's own stores (s_waitcnt vmcnt(0)).  A release fence instead writes the whole XCD's L2 back
// (buffer_wbl2): measured with 390 gather workgroups doing that in one launch, the launch took 100 us instead of 75.
__device__ __forceinline__ double ld_ag(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ag(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_ag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ag(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Wait for a flag another workgroup of the launch posts.  abortf == null: wait as long as it takes.  abortf != null (the one-launch iteration, where nearly every
// workgroup waits for others): give up after VIL_WAIT_TICKS of the 100 MHz wall clock (50 ms -- TIME, not poll rounds: a poll is an L2 / fabric round trip whose
// length depends on what else the device is doing), or as soon as somebody else has, and say so in *abortf -- every later wait of every workgroup then returns at
// once, the launches end, the master marks the solve `done` with status -2, and the host re-runs the solve with the multi-launch structure (vilsolve.hip,
// vil_solve_resident).  The host's own poll window (2 s) is longer than this bound.
#define VIL_WAIT_TICKS 5000000ull
// every 1024th poll round of a bounded wait: has somebody given up, or has this wait lasted too long?  (t0 = 0 on the first call: the clock starts 1024 rounds in)
__device__ __forceinline__ bool wait_expired(unsigned long long& t0, int* abortf) {
    if (ld_ag(abortf) != 0) return true;
    const unsigned long long now = wall_clock64();
    if (t0 == 0) { t0 = now; return false; }
    return now - t0 > VIL_WAIT_TICKS;
}
__device__ __forceinline__ bool spin_until_eq(const int* f, const int v, int* abortf) {
    unsigned long long t0 = 0;
    for (int sp = 1; ld_ag(f) != v; ++sp) {
        __builtin_amdgcn_s_sleep(1);
        if ((sp & 1023) == 0 && abortf && wait_expired(t0, abortf)) { st_ag(abortf, 1); return false; }
    }
    return true;
}
// AG = true: data another workgroup of the SAME launch wrote / will read (agent scope: the level the XCDs' L2s share); false: across a kernel boundary (plain)
template <bool AG> __device__ __forceinline__ double ldx(const double* p) { if constexpr (AG) return ld_ag(p); else return *p; }
template <bool AG> __device__ __forceinline__ void stx(double* p, double v) { if constexpr (AG) st_ag(p, v); else *p = v; }
}  // namespace vd
