// The trust-region step kernel: ONE workgroup runs, per iteration and without any host round trip,
// what ceres' TrustRegionMinimizer + DoglegStrategy + the dense Cholesky of DENSE_SCHUR do around
// the sweep (estimator.cpp:1400-1414; algorithm restated in SURVEY.md Appendix B):
//   judge the candidate just swept (function tolerance, relative decrease, radius update) ->
//   on a new linearisation: Jacobi/dogleg scaling, gradient, Cauchy point, reduced Cholesky solve,
//   landmark back-substitution -> traditional dogleg blend -> model decrease -> candidate state.
// All trust-region state lives in `Ctl` in device memory; `done` makes later launches no-ops.
//
// Dense solve: the scaled reduced matrix M = Sc S' Sc + mu dc^2 (D x D, D = 15K+7) plus the right-hand side as an
// extra row is held in LDS as 16 x 16 lower tiles with a row stride of 17 doubles (120 KB at K = 10) and factored by
// a right-looking blocked Cholesky, NB = 4 (chol_blocked below): every thread factors the 4x4 diagonal block
// redundantly in registers, one thread per row solves the panel, and the trailing update -- the only dense
// contraction on this path -- runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64) with the trailing tiles resident
// in the MFMA accumulators.  The forward substitution is the factorisation of the extra row; the back substitution
// uses the inverses of the 16 x 16 diagonal tiles.  Windows whose tiles exceed LDS (K > 10) run the same code on a
// global (L2-resident) buffer.
#pragma once
#include <type_traits>

#include "vil_dev.hpp"
#include "vil_factors.hpp"
#include "vil_finish.hpp"

namespace vd {

#define STEP_NB 4
#define CH_SLOTS 7     // register-resident 16x16 tiles per wave in the Cholesky (8 waves x 7 >= 55 tiles: D <= 159)
#define PF_N ((55 * 256 + VIL_STEP_THREADS - 1) / VIL_STEP_THREADS)   // prefetched S entries per thread (55 tiles for D <= 159; larger windows loop)

struct StepShared {
    Ctl c;
    double red[64];
    unsigned char tI[256], tJ[256];   // triangular tile index -> (row, col) of the tile
    double xs[320], dinv[320];            // solution of the reduced system, reciprocal Cholesky pivots (D <= 320)
    double y[320];
    double sc[320], dcs[320], gr[320], gn[320];   // Sc, dogleg diagonal, gradient_, gauss_newton_step_ (camera part)
    double gd[320];                               // reduced gradient (rhs row of M): staged once, the packing loop must not touch global memory
    double rt[320];                               // Sc / dogleg diagonal: the candidate step is formed without a divide
    double x0[328];                               // camera part of x_cur (16K + 8 doubles), fetched while the system is being solved
    unsigned char cst[48];                        // pose_const[K] | sb_const[K]
    double hs[12];                                // the helpers' sums, gathered by a spare wave during the chain back substitution
    int need, was_first, ok, cok;
    int hdr_posted, hdr_pad_;                     // persistent solve, master: the next iteration's hand-over line went out inside step_body (below)
    int early, judged;                            // persistent solve, master: the candidate's cost alone ended the solve (judged from the sweep's cost partials, before the gather) / the judge has run
    int done_at_entry, pad0_;                     // the solve was already finished when this launch read Ctl
    long long tacc[6];
};

// block-wide sum(a), sum(b) and sum-or-max(c) with one pair of barriers
template <bool CMAX = false>
__device__ __forceinline__ void bsum3(double& a, double& b, double& c, StepShared& s) {
    a = wave_total_l63(a); b = wave_total_l63(b);          // DPP folds: the wave totals land in lane 63
    c = CMAX ? wave_max_l63(c) : wave_total_l63(c);
    __syncthreads();
    const int w = vil_tid() >> 6, nw = blockDim.x >> 6;
    if ((vil_tid() & 63) == 63) { s.red[w] = a; s.red[8 + w] = b; s.red[16 + w] = c; }
    __syncthreads();
    a = 0; b = 0; c = 0;
    for (int q = 0; q < nw; ++q) { a += s.red[q]; b += s.red[8 + q]; c = CMAX ? fmax(c, s.red[16 + q]) : c + s.red[16 + q]; }
}

// block-wide sum of a, b, d, e and max of c with one pair of barriers
__device__ __forceinline__ void bsum5(double& a, double& b, double& c, double& d, double& e, StepShared& s) {
    a = wave_total_l63(a); b = wave_total_l63(b); c = wave_max_l63(c); d = wave_total_l63(d); e = wave_total_l63(e);
    __syncthreads();
    const int w = vil_tid() >> 6, nw = blockDim.x >> 6;
    if ((vil_tid() & 63) == 63) { s.red[w] = a; s.red[8 + w] = b; s.red[16 + w] = c; s.red[24 + w] = d; s.red[32 + w] = e; }
    __syncthreads();
    a = 0; b = 0; c = 0; d = 0; e = 0;
    for (int q = 0; q < nw; ++q) { a += s.red[q]; b += s.red[8 + q]; c = fmax(c, s.red[16 + q]); d += s.red[24 + q]; e += s.red[32 + q]; }
}

// block-wide sums of six values with one pair of barriers
__device__ __forceinline__ void bsum6(double* v, StepShared& s) {
#pragma unroll
    for (int e = 0; e < 6; ++e) v[e] = wave_total_l63(v[e]);
    __syncthreads();
    const int w = vil_tid() >> 6, nw = blockDim.x >> 6;
    if ((vil_tid() & 63) == 63) { for (int e = 0; e < 6; ++e) s.red[8 * e + w] = v[e]; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 6; ++e) { double a = 0; for (int q = 0; q < nw; ++q) a += s.red[8 * e + q]; v[e] = a; }
}

__device__ __forceinline__ double bsum(double v, StepShared& s) {
    v = wave_total_l63(v);
    __syncthreads();
    if ((vil_tid() & 63) == 63) s.red[vil_tid() >> 6] = v;
    __syncthreads();
    double t = 0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) t += s.red[w];
    return t;
}
__device__ __forceinline__ double bmax(double v, StepShared& s) {
    v = wave_max_l63(v);
    __syncthreads();
    if ((vil_tid() & 63) == 63) s.red[vil_tid() >> 6] = v;
    __syncthreads();
    double t = 0;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) t = fmax(t, s.red[w]);
    return t;
}

// e_l . v_c  for landmark l (compact e storage: anchor 6 | ex 6 | td 1 | per-factor observer 6).
// Dependent fp64 ops cost ~32 cycles each on gfx950: four independent accumulators, column tables instead of
// chained index loads.  The landmark's rows are first fetched into an LmRows (anchor/ex/td row + the first three observers:
// all loads in flight together); a helper workgroup keeps that between its two passes, so the second one -- which sits on
// the step kernel's critical path -- touches no global memory for the usual landmark.
struct LmRows { double e[13], a[6], b[6], c[6]; double m1, m2; int fs, fe, ac, c0, c1, c2, stride = 3; };      // stride: observers between this lane's groups of three (3: the lane has them all; 12: a quad shares a landmark)
template <bool AG = false>      // AG: e_A / e_O were written by the visual workgroups of THIS launch (the one-launch iteration, k_iter)
__device__ __forceinline__ void lm_rows(const DevP& P, const SysBuf& sb, int l, LmRows& r) {
    r.fs = P.glm_start[l]; r.fe = P.glm_start[l + 1]; r.ac = P.glm_acol[l];
    if (r.fe == r.fs) return;
    const double* e = sb.eA + (size_t)l * 13;
#pragma unroll
    for (int k = 0; k < 13; ++k) r.e[k] = ldx<AG>(e + k);
    // observers three at a time: index clamped into the landmark's own range (always a valid address), contribution
    // masked by a select -- the global loads of a group are all in flight together instead of one dependent round trip
    // per factor, and there is no per-lane predicated load (those compile to exec-mask branches with their own waits)
    const int f = r.fs, f1 = min(f + 1, r.fe - 1), f2 = min(f + 2, r.fe - 1);
    r.c0 = P.gfcol[f]; r.c1 = P.gfcol[f1]; r.c2 = P.gfcol[f2];
    const double* e0 = sb.eO + (size_t)f * 6; const double* e1 = sb.eO + (size_t)f1 * 6; const double* e2 = sb.eO + (size_t)f2 * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) { r.a[k] = ldx<AG>(e0 + k); r.b[k] = ldx<AG>(e1 + k); r.c[k] = ldx<AG>(e2 + k); }
    r.m1 = (f + 1 < r.fe) ? 1.0 : 0.0; r.m2 = (f + 2 < r.fe) ? 1.0 : 0.0;
}
// A landmark shared by the four lanes of a quad: lane qd takes the observers [3 qd, 3 qd + 3) (and every twelfth group of three behind them), lane 0 the anchor /
// extrinsic / td row as well.  The lanes' dot products add up to lm_dot_rows' of the whole landmark (quad_total): up to TWELVE observers without a load in the second
// pass -- with one lane per landmark everything past the third observer was a dependent round trip to memory inside the pass the master waits for (configs[1]:
// a third of the landmarks have four to nine observers, so every helper paid two of them).
template <bool AG = false>
__device__ __forceinline__ void lm_rows_quad(const DevP& P, const SysBuf& sb, int l, int qd, LmRows& r, const int G = 4 /* lanes that share the landmark: 4 (a quad) or 2 (a pair: six observers in registers) */) {
    const int fs0 = P.glm_start[l], fe = P.glm_start[l + 1];
    r.ac = qd == 0 ? P.glm_acol[l] : 0; r.stride = 3 * G;
    r.fs = min(fs0 + 3 * qd, fe); r.fe = fe;
    if (fe == fs0) { r.fs = fe; return; }                  // (no observer: the landmark takes no part, anchor row included -- as lm_rows)
    const double* e = sb.eA + (size_t)l * 13;
#pragma unroll
    for (int k = 0; k < 13; ++k) { const double v = ldx<AG>(e + k); r.e[k] = qd == 0 ? v : 0.0; }
    const int f = min(fs0 + 3 * qd, fe - 1), f1 = min(f + 1, fe - 1), f2 = min(f + 2, fe - 1);
    r.c0 = P.gfcol[f]; r.c1 = P.gfcol[f1]; r.c2 = P.gfcol[f2];
    const double* e0 = sb.eO + (size_t)f * 6; const double* e1 = sb.eO + (size_t)f1 * 6; const double* e2 = sb.eO + (size_t)f2 * 6;
    const bool m0 = fs0 + 3 * qd < fe;
#pragma unroll
    for (int k = 0; k < 6; ++k) { const double va = ldx<AG>(e0 + k); r.a[k] = m0 ? va : 0.0; r.b[k] = ldx<AG>(e1 + k); r.c[k] = ldx<AG>(e2 + k); }
    r.m1 = (fs0 + 3 * qd + 1 < fe) ? 1.0 : 0.0; r.m2 = (fs0 + 3 * qd + 2 < fe) ? 1.0 : 0.0;
}
__device__ __forceinline__ double quad_total(double v) { v = dpp_add<0xB1, 0xf>(v); return dpp_add<0x4E, 0xf>(v); }      // quad_perm [1,0,3,2], [2,3,0,1]: the same bits in the four lanes
__device__ __forceinline__ double pair_total(double v) { return dpp_add<0xB1, 0xf>(v); }      // quad_perm [1,0,3,2]: lanes 2 m and 2 m + 1
template <bool AG = false>
__device__ __forceinline__ double lm_dot_rows(const DevP& P, const SysBuf& sb, const LmRows& r, const double* vc) {
    if (r.fe == r.fs) return 0.0;
    const double* e = r.e;
    const double* va = vc + r.ac; const double* vx = vc + col_ex(P);
    double s0 = e[0] * va[0], s1 = e[1] * va[1], s2 = e[2] * va[2], s3 = e[3] * va[3];
    s0 += e[4] * va[4]; s1 += e[5] * va[5];
    s2 += e[6] * vx[0]; s3 += e[7] * vx[1]; s0 += e[8] * vx[2]; s1 += e[9] * vx[3]; s2 += e[10] * vx[4]; s3 += e[11] * vx[5];
    s0 += e[12] * vc[col_td(P)];
    {
        const double* v0 = vc + r.c0; const double* v1 = vc + r.c1; const double* v2 = vc + r.c2;
        const double* a = r.a; const double* b = r.b; const double* c = r.c;
        s0 += a[0] * v0[0]; s1 += a[1] * v0[1]; s2 += a[2] * v0[2]; s3 += a[3] * v0[3]; s0 += a[4] * v0[4]; s1 += a[5] * v0[5];
        s2 += r.m1 * (b[0] * v1[0] + b[1] * v1[1] + b[2] * v1[2]); s3 += r.m1 * (b[3] * v1[3] + b[4] * v1[4] + b[5] * v1[5]);
        s0 += r.m2 * (c[0] * v2[0] + c[1] * v2[1] + c[2] * v2[2]); s1 += r.m2 * (c[3] * v2[3] + c[4] * v2[4] + c[5] * v2[5]);
    }
    for (int f = r.fs + r.stride; f < r.fe; f += r.stride) {          // landmarks seen from more frames than the lane(s) hold: the rest from memory
        const int f1 = min(f + 1, r.fe - 1), f2 = min(f + 2, r.fe - 1);
        const int c0 = P.gfcol[f], c1 = P.gfcol[f1], c2 = P.gfcol[f2];
        const double* e0 = sb.eO + (size_t)f * 6; const double* e1 = sb.eO + (size_t)f1 * 6; const double* e2 = sb.eO + (size_t)f2 * 6;
        double a[6], b[6], c[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { a[k] = ldx<AG>(e0 + k); b[k] = ldx<AG>(e1 + k); c[k] = ldx<AG>(e2 + k); }
        const double* v0 = vc + c0; const double* v1 = vc + c1; const double* v2 = vc + c2;
        const double m1 = (f + 1 < r.fe) ? 1.0 : 0.0, m2 = (f + 2 < r.fe) ? 1.0 : 0.0;
        s0 += a[0] * v0[0]; s1 += a[1] * v0[1]; s2 += a[2] * v0[2]; s3 += a[3] * v0[3]; s0 += a[4] * v0[4]; s1 += a[5] * v0[5];
        s2 += m1 * (b[0] * v1[0] + b[1] * v1[1] + b[2] * v1[2]); s3 += m1 * (b[3] * v1[3] + b[4] * v1[4] + b[5] * v1[5]);
        s0 += m2 * (c[0] * v2[0] + c[1] * v2[1] + c[2] * v2[2]); s1 += m2 * (c[3] * v2[3] + c[4] * v2[4] + c[5] * v2[5]);
    }
    return (s0 + s1) + (s2 + s3);
}
template <bool AG = false>
__device__ __forceinline__ double lm_dot(const DevP& P, const SysBuf& sb, int l, const double* vc) {
    LmRows r; lm_rows<AG>(P, sb, l, r);
    return lm_dot_rows<AG>(P, sb, r, vc);
}

// v^T H v over all free parameters, H = J^T J of the corrected Jacobian, from the reduced pieces:
//   v_c^T H_cc v_c = v_c^T S' v_c + sum_l invp (e_l.v_c)^2     (S' = H_cc - sum_l invp e e^T)
// vc must be readable by every thread (LDS or global).
__device__ __forceinline__ double quad_form(const DevP& P, const SysBuf& sb, const double* vc, const double* vl, StepShared& s) {
    const int D = P.D, L = P.L, t = vil_tid(), NT = blockDim.x;
    double part = 0;
    for (int e = t; e < D * D; e += NT) { const int i = e / D, j = e - i * D; part += vc[i] * sb.S[e] * vc[j]; }
    for (int l = t; l < L; l += NT) {
        const double ip = sb.invp[l];
        if (ip == 0.0) continue;
        const double ev = lm_dot(P, sb, l, vc);
        part += ip * ev * ev + 2.0 * vl[l] * ev + sb.hll[l] * vl[l] * vl[l];
    }
    return bsum(part, s);
}

__device__ __forceinline__ void pose_plus(const double* in, const double* d, double* o) {
    for (int k = 0; k < 3; ++k) o[k] = in[k] + d[k];
    Q4 q = qmul(qload(in + 3), Q4{1.0, 0.5 * d[3], 0.5 * d[4], 0.5 * d[5]});
    const double n = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    o[3] = q.x * n; o[4] = q.y * n; o[5] = q.z * n; o[6] = q.w * n;
}


// triangular tile index -> tile row / column as compile-time constants, and a compile-time counted loop: element
// e = t + u * VIL_STEP_THREADS of the tile array belongs to tile 2u or 2u + 1 (512 threads, 256 elements per tile), so inside a
// statically unrolled loop the tile coordinates are constants selected by one wave-uniform bit -- no table look-up in LDS on
// the address path of the S' prefetch and of the packing loop
__host__ __device__ constexpr int tri_row_c(int q) { int I = 0; while ((I + 1) * (I + 2) / 2 <= q) ++I; return I; }
__host__ __device__ constexpr int tri_col_c(int q) { return q - tri_row_c(q) * (tri_row_c(q) + 1) / 2; }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}
static_assert(VIL_STEP_THREADS == 512, "the tile <-> element mapping above assumes two tiles per 512-thread stride");

__device__ __forceinline__ int tri_off(int i) { return (i * (i + 1)) >> 1; }
}  // namespace vd
#include "vil_chain.hpp"
#include "vil_prechain.hpp"
namespace vd {

// ---- tiled lower storage of the (D+1) x (D+1) reduced matrix (last row = right-hand side): 16 x 16 tiles,
//      tile (I, J), J <= I, at (I(I+1)/2 + J) * TILE_SZ; element (r, c) of a tile at r*TILE_RS + c.  An MFMA operand /
//      accumulator address is then `wave-uniform tile base + per-lane constant`.  TILE_RS = 17 (one padding double
//      per row): an MFMA A/B operand fragment is 16 rows x 4 k -- with a row stride of 16 doubles its 16 row-lanes
//      fall on two LDS bank pairs (8-way conflict, measured 1400 cycles per block step for the operand reads alone);
//      stride 17 spreads them over 16 bank pairs, and the panel's thread-per-row accesses likewise.
#define TILE_RS 17
#define TILE_SZ (16 * TILE_RS)
__device__ __forceinline__ int tl_base(int I, int J) { return (tri_off(I) + J) * TILE_SZ; }
__device__ __forceinline__ int tl_idx(int i, int j) { return tl_base(i >> 4, j >> 4) + (i & 15) * TILE_RS + (j & 15); }
// logical element e = tile*256 + r*16 + c (how the fill loops enumerate the tiles) -> storage index
__device__ __forceinline__ int tl_phys(int e) { return (e >> 8) * TILE_SZ + ((e >> 4) & 15) * TILE_RS + (e & 15); }

// Blocked right-looking Cholesky, NB = 4, on the tiled array A.  fp64 dependent-op latency on gfx950 is ~32
// cycles and a workgroup barrier only ~44, so the algorithm keeps every serial chain short instead of batching:
//   (1) EVERY thread factors the 4x4 diagonal block redundantly in registers (no single-wave phase, no broadcast)
//       and the threads owning a row below it do that row's triangular solve;
//   (2) the trailing update is one v_mfma_f64_16x16x4_f64 per 16x16 tile (K = NB = 4), software-pipelined over
//       the <= 8 tiles a wave owns.
// On return rows < D hold L, row D holds y = L^-1 rhs, s.dinv[j] = 1 / L_jj.  Returns false (uniformly) on a
// non-positive pivot.
struct NoPre { template <class C> __device__ __forceinline__ void operator()(C*, const int*) const {} };
// PRE: called once with the freshly loaded register tiles (REGRES) -- the chain path subtracts W W^T there
template <bool REGRES, class PTR, class PRE = NoPre>
__device__ __forceinline__ bool chol_blocked(PTR A, int D, StepShared& s, double* Acol = nullptr, PRE pre = PRE()) {
    const int t = vil_tid(), NT = blockDim.x, wave = t >> 6, lane = t & 63, NW = NT >> 6;
    const int R = D + 1;                 // rows including the rhs row
    const int T = (R + 15) >> 4;         // tile rows
    const int la = (lane & 15) * TILE_RS + (lane >> 4);     // operand element (row lane&15, k lane>>4) inside a tile
    const int lc = (lane >> 4) * TILE_RS + (lane & 15);     // accumulator element (row lane>>4 (+4g), col lane&15)
#ifdef VIL_STAMPS
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    #define CSTAMP(k) do { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); if (k >= 0) tacc[k < 0 ? 0 : k] += tt_ - tprev; tprev = tt_; } while (0)
#else
    #define CSTAMP(k) do {} while (0)
#endif
    // ---- register-resident trailing matrix: every wave OWNS up to CH_SLOTS fixed 16x16 tiles (triangular tile index
    //      g = wave + NW*u) and keeps them in MFMA accumulators; a tile goes back to LDS only when the factorisation
    //      reaches its tile column.  The per-step LDS traffic drops from a read-modify-write of the whole trailing
    //      matrix to two operand fragments per tile.  Windows with more tiles than NW*CH_SLOTS use the LDS/global RMW path.
    const int ntile_all = (T * (T + 1)) >> 1;     // REGRES requires ntile_all <= NW * CH_SLOTS (true whenever the matrix fits LDS)
    d4 Creg[REGRES ? CH_SLOTS : 1]; int tIJ[REGRES ? CH_SLOTS : 1];
    if constexpr (REGRES) {
#pragma unroll
        for (int u = 0; u < CH_SLOTS; ++u) {
            const int g = wave + NW * u;
            tIJ[u] = -1;
            if (g < ntile_all) {
                const int I = s.tI[g], J = s.tJ[g];
                tIJ[u] = (I << 8) | J;
                const int cb = tl_base(I, J) + lc;
#pragma unroll
                for (int q = 0; q < 4; ++q) Creg[u][q] = A[cb + q * (4 * TILE_RS)];
            }
        }
        pre(Creg, tIJ);
    }
    for (int kb = 0; kb < D; kb += STEP_NB) {
        const int nb = min(STEP_NB, D - kb);
        const int Kt = kb >> 4, ko = kb & 15;
        CSTAMP(-1);
        if constexpr (REGRES) if (ko == 0) {          // the factorisation enters tile column Kt: its owners publish those tiles
#pragma unroll
            for (int u = 0; u < CH_SLOTS; ++u) if (tIJ[u] >= 0 && (tIJ[u] & 255) == Kt) {
                const int cb = tl_base(tIJ[u] >> 8, Kt) + lc;
#pragma unroll
                for (int q = 0; q < 4; ++q) A[cb + q * (4 * TILE_RS)] = Creg[u][q];
            }
            __syncthreads();
            CSTAMP(4);
        }
        if constexpr (!REGRES) if (ko == 0) {         // global path: stage the ACTIVE tile column in LDS for its four block steps
            __syncthreads();                          // the deferred updates of the previous column have landed in A
            for (int e = t; e < (T - Kt) * TILE_SZ; e += NT) { const int I = Kt + e / TILE_SZ, w = e - (I - Kt) * TILE_SZ; Acol[e] = A[tl_base(I, Kt) + w]; }
            __syncthreads();
        }
        // tile (I, Kt) of the active column lives at AC[cbase(I) ...]: the tile array itself (LDS path) or the staged copy
        double* const AC = REGRES ? (double*)A : Acol;
        auto cbase = [&](int I) { return REGRES ? tl_base(I, Kt) : (I - Kt) * TILE_SZ; };
        // ---- 1. diagonal 4x4 block, redundantly per thread (identity padding for a short last block) ----------
        const int db = cbase(Kt) + ko * TILE_RS + ko;
        double d00 = AC[db], d10 = 0, d11 = 1, d20 = 0, d21 = 0, d22 = 1, d30 = 0, d31 = 0, d32 = 0, d33 = 1;
        if (nb > 1) { d10 = AC[db + TILE_RS]; d11 = AC[db + TILE_RS + 1]; }
        if (nb > 2) { d20 = AC[db + 2 * TILE_RS]; d21 = AC[db + 2 * TILE_RS + 1]; d22 = AC[db + 2 * TILE_RS + 2]; }
        if (nb > 3) { d30 = AC[db + 3 * TILE_RS]; d31 = AC[db + 3 * TILE_RS + 1]; d32 = AC[db + 3 * TILE_RS + 2]; d33 = AC[db + 3 * TILE_RS + 3]; }
        double l00, r0_, l11, r1_, l22, r2_, l33, r3_;
        bool ok = d00 > 0.0 && isfinite(d00);
        sqrt_rsqrt(d00, l00, r0_);
        const double l10 = d10 * r0_, l20 = d20 * r0_, l30 = d30 * r0_;
        d11 -= l10 * l10; ok = ok && d11 > 0.0 && isfinite(d11);
        sqrt_rsqrt(d11, l11, r1_);
        const double l21 = (d21 - l20 * l10) * r1_, l31 = (d31 - l30 * l10) * r1_;
        d22 -= l20 * l20 + l21 * l21; ok = ok && d22 > 0.0 && isfinite(d22);
        sqrt_rsqrt(d22, l22, r2_);
        const double l32 = (d32 - l30 * l20 - l31 * l21) * r2_;
        d33 -= l30 * l30 + l31 * l31 + l32 * l32; ok = ok && d33 > 0.0 && isfinite(d33);
        sqrt_rsqrt(d33, l33, r3_);
        if (!ok) return false;          // identical data in every thread: uniform exit
        CSTAMP(0);
        // ---- 2. panel rows (incl. the rhs row): forward substitution against the block ----------------------------
        const int r0 = kb + nb;
        for (int i = r0 + t; i < R; i += NT) {
            const int base = cbase(i >> 4) + (i & 15) * TILE_RS + ko;
            const double a0 = AC[base], a1 = nb > 1 ? AC[base + 1] : 0.0, a2 = nb > 2 ? AC[base + 2] : 0.0, a3 = nb > 3 ? AC[base + 3] : 0.0;
            const double x0 = a0 * r0_;
            const double x1 = (a1 - x0 * l10) * r1_;
            const double x2 = (a2 - x0 * l20 - x1 * l21) * r2_;
            const double x3 = (a3 - x0 * l30 - x1 * l31 - x2 * l32) * r3_;
            AC[base] = x0; if (nb > 1) AC[base + 1] = x1; if (nb > 2) AC[base + 2] = x2; if (nb > 3) AC[base + 3] = x3;
        }
        if (t == NT - 2) {              // the factored block itself and the reciprocal pivots (a thread without a panel row)
            AC[db] = l00; s.dinv[kb] = r0_;
            if (nb > 1) { AC[db + TILE_RS] = l10; AC[db + TILE_RS + 1] = l11; s.dinv[kb + 1] = r1_; }
            if (nb > 2) { AC[db + 2 * TILE_RS] = l20; AC[db + 2 * TILE_RS + 1] = l21; AC[db + 2 * TILE_RS + 2] = l22; s.dinv[kb + 2] = r2_; }
            if (nb > 3) { AC[db + 3 * TILE_RS] = l30; AC[db + 3 * TILE_RS + 1] = l31; AC[db + 3 * TILE_RS + 2] = l32; AC[db + 3 * TILE_RS + 3] = l33; s.dinv[kb + 3] = r3_; }
        }
        CSTAMP(5);
        __syncthreads();
        CSTAMP(1);
        // ---- 3. trailing update on the fp64 matrix cores: C[r][c] -= sum_{k<4} L[r][kb+k] L[c][kb+k], r, c >= r0 ------
        const bool kk = (lane >> 4) < nb;
        if constexpr (REGRES) {
#pragma unroll
            for (int u = 0; u < CH_SLOTS; ++u) {
                const int I = tIJ[u] >> 8, J = tIJ[u] & 255;
                if (tIJ[u] < 0 || J < Kt) continue;                       // finished (or empty) slot: wave-uniform
                if (J > Kt) {                                            // register tile: rows/cols are beyond the panel
                    // unconditional reads (inside the tile for every lane) + select: a per-lane predicated load would
                    // compile to an exec-mask branch with its own wait
                    const double a_ = A[tl_base(I, Kt) + la + ko], b_ = A[tl_base(J, Kt) + la + ko];
                    const double av = kk ? -a_ : 0.0, bv = kk ? b_ : 0.0;
                    Creg[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, Creg[u], 0, 0, 0);
                } else if (r0 < R) {                                     // tile of the active column: lives in LDS
                    const int rr = (I << 4) + (lane & 15), cr = (J << 4) + (lane & 15);
                    const double a_ = A[tl_base(I, Kt) + la + ko], b_ = A[tl_base(J, Kt) + la + ko];
                    const double av = (rr >= r0 && rr < R && kk) ? a_ : 0.0;
                    const double bv = (cr >= r0 && cr < D && kk) ? b_ : 0.0;
                    d4 z = {0.0, 0.0, 0.0, 0.0};
                    const d4 acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, z, 0, 0, 0);
                    const int cb = tl_base(I, J) + lc;
#pragma unroll
                    for (int g = 0; g < 4; ++g) A[cb + g * (4 * TILE_RS)] -= acc[g];
                }
            }
        } else { if (r0 < R) {
            // Matrix in global (L2) memory, K > 10.  Only the tiles of the ACTIVE tile column need this block step's update
            // now (the next diagonal block and panel read them); every tile to the right of it is updated ONCE per tile
            // column with the whole 16-wide panel (four MFMAs in the accumulator, one read-modify-write) -- a quarter of the
            // read-modify-write traffic through L2, which is what bounds this path.
            {   // (i) tiles (I, Kt), I >= r0 >> 4
                const int I0 = r0 >> 4, nact = T - I0;
                for (int u0 = wave; u0 < nact; u0 += NW) {
                    const int I = I0 + u0;
                    const int rr = (I << 4) + (lane & 15), cr = (Kt << 4) + (lane & 15);
                    const double a_ = AC[cbase(I) + la + ko], b_ = AC[cbase(Kt) + la + ko];
                    const double av = (rr >= r0 && rr < R && kk) ? a_ : 0.0, bv = (cr >= r0 && cr < D && kk) ? b_ : 0.0;
                    d4 z = {0.0, 0.0, 0.0, 0.0};
                    const d4 acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, z, 0, 0, 0);
                    const int cb = cbase(I) + lc;
#pragma unroll
                    for (int g = 0; g < 4; ++g) AC[cb + g * (4 * TILE_RS)] -= acc[g];
                }
            }
            const bool col_done = (ko + nb >= 16) || (kb + nb >= D);          // last block step inside tile column Kt
            if (col_done && Kt + 1 < T) {   // (ii) tiles (I, J), Kt < J <= I, with the panel columns [16 Kt, min(16 Kt + 16, D))
                __syncthreads();            // the panel of this block step (written by other threads) is part of the operands
                const int I0 = Kt + 1, n = T - I0, ntile = n * (n + 1) / 2;
                const int kv = min(16, D - (Kt << 4));                        // valid factor columns in this tile column
                for (int t0 = 0; t0 < ntile; t0 += 4 * NW) {
                    d4 acc[4]; int cb[4];
                    const int cnt = min(4, (ntile - t0 - wave + NW - 1) / NW);   // wave-uniform number of live slots
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (u < cnt) {
                        const int tile = t0 + wave + u * NW;
                        int Ir = (int)((sqrtf(8.f * (float)tile + 1.f) - 1.f) * 0.5f);
                        if (((Ir + 1) * (Ir + 2)) / 2 <= tile) ++Ir;
                        if ((Ir * (Ir + 1)) / 2 > tile) --Ir;
                        const int Jr = tile - (Ir * (Ir + 1)) / 2;
                        const int I = I0 + Ir, J = I0 + Jr;
                        cb[u] = tl_base(I, J);
                        const int rr = (I << 4) + (lane & 15), cr = (J << 4) + (lane & 15);
                        d4 c4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int ks = 0; ks < 16; ks += 4) {
                            const bool kq = ks + (lane >> 4) < kv;
                            const double a_ = AC[cbase(I) + la + ks], b_ = AC[cbase(J) + la + ks];
                            c4 = __builtin_amdgcn_mfma_f64_16x16x4f64((rr < R && kq) ? a_ : 0.0, (cr < D && kq) ? b_ : 0.0, c4, 0, 0, 0);
                        }
                        acc[u] = c4;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (u < cnt) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) A[cb[u] + lc + g * (4 * TILE_RS)] -= acc[u][g];
                    }
                }
            }
            if (col_done) {                 // the factored tile column goes back to the tile array (back substitution, later operands)
                __syncthreads();
                for (int e = t; e < (T - Kt) * TILE_SZ; e += NT) { const int I = Kt + e / TILE_SZ, w = e - (I - Kt) * TILE_SZ; A[tl_base(I, Kt) + w] = Acol[e]; }
            }
        } }
        CSTAMP(2);
        __syncthreads();
        CSTAMP(3);
    }
#ifdef VIL_STAMPS
    if (vil_tid() == 0) { for (int q = 0; q < 6; ++q) s.tacc[q] = tacc[q]; }
#endif
    return true;
}

// The same factorisation with look-ahead, for a matrix whose tiles are register-resident (REGRES above).  The serial chain of a block step --
// four dependent pivots (~230 cycles each: rsq seed, coupled Newton step, the block's own updates), then the panel's forward substitution --
// does not need the whole workgroup, and the trailing update does not need to be finished before the next block starts: block k + 1 reads
// four columns only.  So the LAST wave(s) of the workgroup (the panel group, one wave per 64 panel rows) run that chain, applying panel k to
// the four columns of block k + 1 themselves (scalar FMAs on the thread's own row, the 4 x 4 block redundantly per thread), while the other
// waves apply panel k to everything else on the matrix cores: their register tiles, and -- with the stores masked to the columns right of
// block k + 1 -- the LDS tiles of the tile column block k + 1 lives in.  A tile column goes to LDS one block step before the factorisation
// enters it.  ONE barrier per block step, and the panel group's chain (~2000 cycles) is the step; chol_blocked needs ~3500 (diagonal block,
// panel, barrier, trailing update, barrier, everybody in lockstep).
// On return rows < D hold L, row D holds y = L^-1 rhs, s.dinv[j] = 1 / L_jj.  Returns false (uniformly) on a non-positive pivot.
template <int SLOTS = CH_SLOTS, bool DIAG = true, class PTR, class PRE = NoPre>      // SLOTS: register tiles per wave (8 waves x SLOTS >= the tiles of the matrix); DIAG = false: the diagonal of L is not formed (the solves read s.dinv), its slots keep their old values
__device__ __forceinline__ bool chol_lookahead(PTR A, int D, StepShared& s, PRE pre = PRE()) {
    const int t = vil_tid(), NT = blockDim.x, wave = t >> 6, lane = t & 63, NW = NT >> 6;
    const int R = D + 1;                 // rows including the rhs row
    const int T = (R + 15) >> 4;         // tile rows
    const int la = (lane & 15) * TILE_RS + (lane >> 4);     // operand element (row lane&15, k lane>>4) inside a tile
    const int lc = (lane >> 4) * TILE_RS + (lane & 15);     // accumulator element (row lane>>4 (+4g), col lane&15)
    const int ntile_all = (T * (T + 1)) >> 1;               // <= NW * SLOTS
    // panel group: the last npw waves; thread tp of it owns row STEP_NB + tp.  The tiles are dealt to the other waves first
    const int npw = max(1, (R - min(STEP_NB, D) + 63) >> 6);   // the first block has the most panel rows: R - min(NB, D)
    // The tile wave(s) that would share a SIMD with the panel wave(s) (wave w runs on SIMD w & 3 as the workgroup is dispatched) sit the factorisation out
    // when the other tile waves have slots for every tile (K <= 12): the panel group's fp64 chain has its SIMD's issue port to itself (-3 % per block step
    // in the stand-alone timing).  A different wave placement only loses that.
    const int NWP = NW - npw;                              // first panel wave
    const int n_idle = (ntile_all <= (NWP - npw) * SLOTS && NWP >= 4) ? npw : 0;
    const int NWT = NWP - n_idle;
    const bool idle_w = wave < NWP && wave >= NWP - 4 && wave < NWP - 4 + n_idle;
    const int wr = wave - (wave >= NWP - 4 + n_idle && wave < NWP ? n_idle : 0);       // rank among the tile waves
    const int tp = t - 64 * NWP;
    d4 Creg[SLOTS]; int tIJ[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int g = idle_w ? ntile_all : (wave < NWP ? wr + NWT * u : NWT * SLOTS + (wave - NWP) + npw * u);
        tIJ[u] = -1;
        if (g < ntile_all) {
            const int I = s.tI[g], J = s.tJ[g];
            tIJ[u] = (I << 8) | J;
            const int cb = tl_base(I, J) + lc;
#pragma unroll
            for (int q = 0; q < 4; ++q) Creg[u][q] = A[cb + q * (4 * TILE_RS)];
        }
    }
    pre(Creg, tIJ);
    if (t == 0) s.cok = 1;
    // Panel group: block at column kb -- its 4 x 4 diagonal block (every thread, redundantly) and the rows below it.  UPD: first subtract the
    // previous panel (columns kb - 4 .. kb - 1, always a full block) from what is read.
    // FULL: a whole 4 x 4 block (only the last block of the matrix can be shorter) -- no branch anywhere between the loads and the stores, so that
    // the scheduler can fill the latency shadows of the pivot chain with the panel update and the forward substitution.
    // (thread tp of the group owns the FIXED row STEP_NB + tp: its LDS offset is formed once -- the tp-th row below the current block moved every step,
    //  and the integer multiplies of its tile address sat in front of every step's loads)
    const int prow = STEP_NB + tp, prc = min(max(prow, 0), R - 1);
    const int rowb = tl_base(prc >> 4, 0) + (prc & 15) * TILE_RS;
    auto diag_panel = [&](const int kb, auto full_c, auto upd_c) {
        constexpr bool FULL = decltype(full_c)::value, UPD = decltype(upd_c)::value;
        const int nb = FULL ? STEP_NB : min(STEP_NB, D - kb), Kt = kb >> 4, ko = kb & 15;
        const int db = tl_base(Kt, Kt) + ko * TILE_RS + ko;
        const bool row = prow >= kb + nb && prow < R;         // (a row inside or above the block computes on whatever it reads and stores into padding)
        const int base = rowb + Kt * TILE_SZ + ko;
        double d00 = A[db], d10 = 0, d11 = 1, d20 = 0, d21 = 0, d22 = 1, d30 = 0, d31 = 0, d32 = 0, d33 = 1;
        if (nb > 1) { d10 = A[db + TILE_RS]; d11 = A[db + TILE_RS + 1]; }
        if (nb > 2) { d20 = A[db + 2 * TILE_RS]; d21 = A[db + 2 * TILE_RS + 1]; d22 = A[db + 2 * TILE_RS + 2]; }
        if (nb > 3) { d30 = A[db + 3 * TILE_RS]; d31 = A[db + 3 * TILE_RS + 1]; d32 = A[db + 3 * TILE_RS + 2]; d33 = A[db + 3 * TILE_RS + 3]; }
        double a0 = A[base], a1 = nb > 1 ? A[base + 1] : 0.0, a2 = nb > 2 ? A[base + 2] : 0.0, a3 = nb > 3 ? A[base + 3] : 0.0;
        if constexpr (UPD) {
            const int kp = kb - STEP_NB, Kp = kp >> 4, kpo = kp & 15;      // the previous panel: columns kp .. kp + 3 of tile column Kp
            const int pb = tl_base(Kt, Kp) + ko * TILE_RS + kpo;           // rows kb .. kb + 3 of it (all in tile row Kt)
            const int pr = rowb + Kp * TILE_SZ + kpo;
            double xd[4][4], xr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) xd[r][q] = (FULL || r < nb) ? A[pb + r * TILE_RS + q] : 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[q] = A[pr + q];
            // (rows of a short last block that do not exist: xd = 0, the identity padding stays)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                d00 = fma(-xd[0][q], xd[0][q], d00);
                d10 = fma(-xd[1][q], xd[0][q], d10); d11 = fma(-xd[1][q], xd[1][q], d11);
                d20 = fma(-xd[2][q], xd[0][q], d20); d21 = fma(-xd[2][q], xd[1][q], d21); d22 = fma(-xd[2][q], xd[2][q], d22);
                d30 = fma(-xd[3][q], xd[0][q], d30); d31 = fma(-xd[3][q], xd[1][q], d31); d32 = fma(-xd[3][q], xd[2][q], d32); d33 = fma(-xd[3][q], xd[3][q], d33);
                a0 = fma(-xr[q], xd[0][q], a0); a1 = fma(-xr[q], xd[1][q], a1); a2 = fma(-xr[q], xd[2][q], a2); a3 = fma(-xr[q], xd[3][q], a3);
            }
        }
        // every pivot waits for the previous reciprocal root, one multiply and one fma; everything else is formed beside the chain
        // (a pivot that is not positive and finite turns its reciprocal root into NaN or inf -- rsq of a negative number, 0 x inf or inf x 0 in the Newton
        //  step -- and that reaches every later root through l and d: ONE comparison of the four roots' sum replaces a compare + class test per pivot)
        double l00 = 0, r0_, l11 = 0, r1_, l22 = 0, r2_, l33 = 0, r3_;
        if constexpr (DIAG) rsqrt_sqrt(d00, l00, r0_); else r0_ = rsqrt_1(d00);
        const double l10 = d10 * r0_, l20 = d20 * r0_, l30 = d30 * r0_, x0 = a0 * r0_;
        d11 = fma(-l10, l10, d11);
        const double t21 = fma(-l20, l10, d21), t31 = fma(-l30, l10, d31), u22 = fma(-l20, l20, d22), u33a = fma(-l30, l30, d33), v32 = fma(-l30, l20, d32);
        const double y1 = fma(-x0, l10, a1), y2a = fma(-x0, l20, a2), y3a = fma(-x0, l30, a3);
        if constexpr (DIAG) rsqrt_sqrt(d11, l11, r1_); else r1_ = rsqrt_1(d11);
        const double l21 = t21 * r1_, l31 = t31 * r1_, x1 = y1 * r1_;
        d22 = fma(-l21, l21, u22);
        const double t32 = fma(-l31, l21, v32), u33 = fma(-l31, l31, u33a), y2 = fma(-x1, l21, y2a), y3b = fma(-x1, l31, y3a);
        if constexpr (DIAG) rsqrt_sqrt(d22, l22, r2_); else r2_ = rsqrt_1(d22);
        const double l32 = t32 * r2_, x2 = y2 * r2_;
        d33 = fma(-l32, l32, u33);
        const double y3 = fma(-x2, l32, y3b);
        if constexpr (DIAG) rsqrt_sqrt(d33, l33, r3_); else r3_ = rsqrt_1(d33);
        const double x3 = y3 * r3_;
        const bool ok = (r0_ + r1_) + (r2_ + r3_) < 1.7976931348623157e308;      // false for NaN and for +inf
        // (a non-positive pivot: identical data in every thread of the group; what is stored below is not read -- everyone leaves after the barrier)
        // (a thread without a row stores into the padding double of ITS row in tile column 0 -- element 16 of a tile row, never read, a different address
        //  in every lane -- instead of branching: a conditional store would let the compiler sink this row's loads, update and substitution behind the
        //  pivot chain, into the branch)
        if constexpr (FULL) { const int pad = rowb + 16; A[row ? base : pad] = x0; A[row ? base + 1 : pad] = x1; A[row ? base + 2 : pad] = x2; A[row ? base + 3 : pad] = x3; }
        else if (row) { A[base] = x0; if (nb > 1) A[base + 1] = x1; if (nb > 2) A[base + 2] = x2; if (nb > 3) A[base + 3] = x3; }
        if (tp == 0) {                                       // the factored block itself and the reciprocal pivots
            if (!ok) s.cok = 0;
            s.dinv[kb] = r0_;
            if (nb > 1) { A[db + TILE_RS] = l10; s.dinv[kb + 1] = r1_; }
            if (nb > 2) { A[db + 2 * TILE_RS] = l20; A[db + 2 * TILE_RS + 1] = l21; s.dinv[kb + 2] = r2_; }
            if (nb > 3) { A[db + 3 * TILE_RS] = l30; A[db + 3 * TILE_RS + 1] = l31; A[db + 3 * TILE_RS + 2] = l32; s.dinv[kb + 3] = r3_; }
            if constexpr (DIAG) { A[db] = l00; if (nb > 1) A[db + TILE_RS + 1] = l11; if (nb > 2) A[db + 2 * TILE_RS + 2] = l22; if (nb > 3) A[db + 3 * TILE_RS + 3] = l33; }
        }
    };
    // tile column 0 to LDS (tile column 1 as well when the matrix has fewer than three blocks in column 0 -- never: a tile column has four), block 0
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) if (tIJ[u] >= 0 && (tIJ[u] & 255) == 0) {
        const int cb = tl_base(tIJ[u] >> 8, 0) + lc;
#pragma unroll
        for (int q = 0; q < 4; ++q) A[cb + q * (4 * TILE_RS)] = Creg[u][q];
    }
    lds_barrier();
    if (tp >= 0) { if (D >= STEP_NB) diag_panel(0, std::true_type{}, std::false_type{}); else diag_panel(0, std::false_type{}, std::false_type{}); }
    lds_barrier();
    if (!s.cok) return false;
    for (int kb = 0; kb + STEP_NB < D; kb += STEP_NB) {      // panel kb (a full block) is in LDS; the last block has no successor
        const int Kt = kb >> 4, ko = kb & 15;
        const int kb1 = kb + STEP_NB, Kn = kb1 >> 4;         // the next block and its tile column (LDS-resident)
        const int c0 = kb1 + min(STEP_NB, D - kb1);          // first column right of the next block: the slice [c0, c0 + 4) is published after this update
        if (tp >= 0) { if (kb1 + STEP_NB <= D) diag_panel(kb1, std::true_type{}, std::true_type{}); else diag_panel(kb1, std::false_type{}, std::true_type{}); }
        // Every tile stays in its wave's registers for the whole factorisation.  What the panel group reads next step -- the four columns [c0, c0 + 4) of the
        // block after the next, with the panels up to kb applied (it applies panel kb1 itself) -- is the only thing a step writes to LDS: 16 x 4 values per
        // tile of that tile column instead of a read-modify-write of the whole tile.  Register entries left of c0 and above the diagonal go stale: dead.
        // (slot by slot on purpose: with all operand loads hoisted in front of the matrix-core instructions every wave issues 2 x SLOTS loads per step for
        //  finished slots too, and the step got 10 % LONGER in the stand-alone timing -- the fp64 matrix pipe takes 64 cycles per instruction and SIMD, the
        //  slice stores wait for it, and the panel group's own LDS traffic queues behind the extra loads)
        const int Kc = c0 >> 4, cs = c0 & 15;
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) {
            const int I = tIJ[u] >> 8, J = tIJ[u] & 255;
            if (tIJ[u] < 0 || J < Kc) continue;                       // finished (or empty) slot: wave-uniform
            const double a_ = A[tl_base(I, Kt) + la + ko], b_ = A[tl_base(J, Kt) + la + ko];
            Creg[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_, b_, Creg[u], 0, 0, 0);
            if (J == Kc && c0 < D) {
                const int cl = lane & 15;
                if (cl >= cs && cl < cs + STEP_NB) {
                    const int cb = tl_base(I, J) + lc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) A[cb + q * (4 * TILE_RS)] = Creg[u][q];
                }
            }
        }
        lds_barrier();
    }
    // (a non-positive pivot is looked at once, here: the blocks after it computed garbage nobody uses, and the flag's LDS round trip stays off the chain)
    return s.cok != 0;
}

// ---- The dense factorisation for R = D + 1 <= 68 rows (K <= 10 frames): 16-wide panels factored by ONE wave, a matrix row per lane --------------------------
// chol_lookahead's block step (four pivots) is a serial sequence on its panel wave -- LDS round trip in, 56 fmas applying the previous panel before the first pivot can
// start, the four pivots, LDS round trip out, barrier: ~1800 ticks, 17 times at D = 67.  Here a whole tile column (16 pivots) is ONE such sequence:
//   * lane l of the chain wave owns row 16 Kt + l of tile column Kt (the diagonal tile's rows and everything below, the rhs row included) in 16 registers;
//   * column by column, right looking: l_ik = t_ik r_k, then t_ic -= l_ik l_ck for the columns c > k of the tile column -- l_ck is row c's entry, another LANE's
//     register, read with v_readlane (a uniform operand of the fma);
//   * the pivot chain is short: row k + 1's pivot candidate, without its last term, is broadcast a step early (tp); the last term's factor l_{k+1,k} is the ONE
//     broadcast on the chain, and every lane forms d = tp - l^2, its reciprocal root and its own l_{i,k+1} from it -- readlane, fma, rsq + Newton, multiply per pivot,
//     everything else (15 - k broadcasts and fmas per step) fills the latency shadows;
//   * the panels meet the matrix cores once per tile column: barrier, the tile waves subtract X X^T (four v_mfma_f64_16x16x4 per tile, tiles register-resident as in
//     chol_lookahead), the tiles of the NEXT tile column go to LDS first, barrier, the chain wave goes on while the tile waves finish the columns right of it.
//   * R = 68 at K = 10 is four rows more than a wave has lanes: tile column 0 starts like chol_lookahead -- EVERY lane factors the leading 4 x 4 block redundantly
//     (uniform loads, no broadcast) -- and its lanes own rows 4 .. 67; the broadcasts of columns 4 .. 15 come from lanes 0 .. 11.
// Same contract as chol_lookahead<SLOTS, false>: rows < D hold L below the diagonal (the diagonal slots and the diagonal tiles' upper triangles hold values nobody
// reads), row D holds y = L^-1 rhs, s.dinv[j] = 1 / L_jj; false (uniformly) on a pivot that is not positive and finite.  Requires 16 <= D, D + 1 <= 132 (past 68 rows a second wave takes the rows the chain wave has no lanes for: rowwave_panel_second), eight waves.
__device__ __forceinline__ double bcast_lane(const double v, const int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
template <bool FIRST, bool FULL, int NTC = 0, class PTR>      // FULL: sixteen pivots; NTC > 0: the pivot count of a short last panel as a constant (3 at K = 10, 15 at K = 20) -- no branch in the panel either way
__device__ __forceinline__ bool rowwave_panel(PTR A, const int Kt, const int D, const int R, StepShared& s, const int lane) {
    constexpr int o = FIRST ? 4 : 0;                 // lane l owns row 16 Kt + o + l; the row of column k lives in lane k - o
    const int c0 = Kt << 4;
    const int nt = FULL ? 16 : NTC > 0 ? NTC : min(16, D - c0);      // pivots of this tile column
    const int row = c0 + o + lane, rc = min(row, R - 1);
    const int base = tl_base(rc >> 4, Kt) + (rc & 15) * TILE_RS;
    double t[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) t[c] = A[base + c];
    // FIRST: the leading 4 x 4 block, redundantly in every lane (chol_lookahead's diag_panel arithmetic)
    double l10 = 0, l20 = 0, l21 = 0, l30 = 0, l31 = 0, l32 = 0, r4_[4] = {0, 0, 0, 0};
    if constexpr (FIRST) {
        const double d00 = A[0], d10 = A[TILE_RS], d20 = A[2 * TILE_RS], d21 = A[2 * TILE_RS + 1], d30 = A[3 * TILE_RS], d31 = A[3 * TILE_RS + 1], d32 = A[3 * TILE_RS + 2];
        double d11 = A[TILE_RS + 1], d22 = A[2 * TILE_RS + 2], d33 = A[3 * TILE_RS + 3];
        r4_[0] = rsqrt_1(d00);
        l10 = d10 * r4_[0]; l20 = d20 * r4_[0]; l30 = d30 * r4_[0];
        d11 = fma(-l10, l10, d11);
        const double t21 = fma(-l20, l10, d21), t31 = fma(-l30, l10, d31), u22 = fma(-l20, l20, d22), u33a = fma(-l30, l30, d33), v32 = fma(-l30, l20, d32);
        r4_[1] = rsqrt_1(d11);
        l21 = t21 * r4_[1]; l31 = t31 * r4_[1];
        d22 = fma(-l21, l21, u22);
        const double t32 = fma(-l31, l21, v32), u33 = fma(-l31, l31, u33a);
        r4_[2] = rsqrt_1(d22);
        l32 = t32 * r4_[2];
        d33 = fma(-l32, l32, u33);
        r4_[3] = rsqrt_1(d33);
        if (lane == 0) { A[TILE_RS] = l10; A[2 * TILE_RS] = l20; A[2 * TILE_RS + 1] = l21; A[3 * TILE_RS] = l30; A[3 * TILE_RS + 1] = l31; A[3 * TILE_RS + 2] = l32; }
    }
    // row c's entry in column m < c, final: the leading block's rows are in every lane, the others in lane c - o
    auto B = [&](const int c, const int m) -> double {
        if constexpr (FIRST) { if (c < 4) return c == 1 ? l10 : c == 2 ? (m == 0 ? l20 : l21) : (m == 0 ? l30 : m == 1 ? l31 : l32); }
        return bcast_lane(t[m], c - o);
    };
    double rlast = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (FULL || k < nt) {                            // (wave-uniform)
            double r;
            if (FIRST && k < 4) {
                if (k >= 1) t[k] = fma(-t[k - 1], B(k, k - 1), t[k]);
                r = r4_[k];
            } else {
                if (k >= 1) t[k] = fma(-t[k - 1], B(k, k - 1), t[k]);
                r = rsqrt_1(bcast_lane(t[k], k - o));
            }
            t[k] *= r;
            s.dinv[c0 + k] = r;                          // (every lane, the same value to the same address: one instruction, no select)
            rlast = r;
#pragma unroll
            for (int c = k + 2; c < 16; ++c) if (FULL || c < nt) t[c] = fma(-t[k], B(c, k), t[c]);
        }
    }
    if (row < R) {
#pragma unroll
        for (int c = 0; c < 16; ++c) if (FULL || c < nt) A[base + c] = t[c];
    }
    return rlast < 1.7976931348623157e308;               // false for NaN and for +inf: a pivot that was not positive and finite reaches every later root (through l and d), the panel's last one included
}
// The full panel (sixteen pivots) with most broadcasts through LDS.  A term l_im l_cm costs a v_readlane pair, the wait states behind it and the fma; the panel is bound
// by its instruction count, and 120 of those per lane is most of it.  Here every lane stores l_im into its tile row as soon as it exists (the panel's result, stored
// column by column instead of at the end), and the terms that are not urgent read l_cm from the diagonal tile with plain LDS loads of a uniform address, one step late:
//   term m of column c:  m = c - 1   readlane, at step c (the pivot waits for it)
//                        m = c - 2   readlane, at step c - 1
//                        m <= c - 3  LDS: loaded during step m + 1, applied at step m + 2
// Per column the terms are applied in the order of m, as the readlane-only panel applies them: the same bits.
// (Lanes past the last row own a copy of row R - 1: the same values to the same addresses, no masks.)
template <bool FIRST, bool PROG /* a second panel wave follows: the step count is posted behind every step's stores */, class PTR>
__device__ __forceinline__ bool rowwave_panel_full(PTR A, const int Kt, const int R, StepShared& s, const int lane) {
    constexpr int o = FIRST ? 4 : 0;
    const int c0 = Kt << 4;
    const int rc = min(c0 + o + lane, R - 1);
    const int base = tl_base(rc >> 4, Kt) + (rc & 15) * TILE_RS;
    const int db = tl_base(Kt, Kt);                  // the diagonal tile: row c of the panel at db + c * TILE_RS
    double t[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) t[c] = A[base + c];
    double l10 = 0, l20 = 0, l21 = 0, l30 = 0, l31 = 0, l32 = 0, r4_[4] = {0, 0, 0, 0};
    if constexpr (FIRST) {
        const double d00 = A[0], d10 = A[TILE_RS], d20 = A[2 * TILE_RS], d21 = A[2 * TILE_RS + 1], d30 = A[3 * TILE_RS], d31 = A[3 * TILE_RS + 1], d32 = A[3 * TILE_RS + 2];
        double d11 = A[TILE_RS + 1], d22 = A[2 * TILE_RS + 2], d33 = A[3 * TILE_RS + 3];
        r4_[0] = rsqrt_1(d00);
        l10 = d10 * r4_[0]; l20 = d20 * r4_[0]; l30 = d30 * r4_[0];
        d11 = fma(-l10, l10, d11);
        const double t21 = fma(-l20, l10, d21), t31 = fma(-l30, l10, d31), u22 = fma(-l20, l20, d22), u33a = fma(-l30, l30, d33), v32 = fma(-l30, l20, d32);
        r4_[1] = rsqrt_1(d11);
        l21 = t21 * r4_[1]; l31 = t31 * r4_[1];
        d22 = fma(-l21, l21, u22);
        const double t32 = fma(-l31, l21, v32), u33 = fma(-l31, l31, u33a);
        r4_[2] = rsqrt_1(d22);
        l32 = t32 * r4_[2];
        d33 = fma(-l32, l32, u33);
        r4_[3] = rsqrt_1(d33);
        if (lane == 0) { A[TILE_RS] = l10; A[2 * TILE_RS] = l20; A[2 * TILE_RS + 1] = l21; A[3 * TILE_RS] = l30; A[3 * TILE_RS + 1] = l31; A[3 * TILE_RS + 2] = l32; }
    }
    auto L4 = [&](const int c, const int m) -> double { return c == 1 ? l10 : c == 2 ? (m == 0 ? l20 : l21) : (m == 0 ? l30 : m == 1 ? l31 : l32); };
    auto Bfast = [&](const int c, const int m) -> double {       // row c's entry in column m, out of the owner's register
        if constexpr (FIRST) { if (c < 4) return L4(c, m); }
        return bcast_lane(t[m], c - o);
    };
    double bv[16], bn[16];                           // column m = k - 2 of the diagonal tile (rows k + 1 ..), and the next step's
    double r = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        // (the loads for the NEXT step's terms are issued first and pinned there: left to itself the scheduler sinks them to their uses, and every step then waits
        //  out an LDS round trip)
        if (k >= 1 && k + 2 < 16) {                  // column k - 1, rows k + 2 ..: stored at the end of the last step
#pragma unroll
            for (int c = k + 2; c < 16; ++c) { if (FIRST && c < 4) bn[c] = L4(c, k - 1); else bn[c] = A[db + c * TILE_RS + (k - 1)]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (k >= 2) {
#pragma unroll
            for (int c = k + 1; c < 16; ++c) t[c] = fma(-t[k - 2], bv[c], t[c]);
        }
        if (k >= 1 && k + 1 < 16) t[k + 1] = fma(-t[k - 1], Bfast(k + 1, k - 1), t[k + 1]);
        if (k >= 1) t[k] = fma(-t[k - 1], Bfast(k, k - 1), t[k]);
        r = (FIRST && k < 4) ? r4_[k] : rsqrt_1(bcast_lane(t[k], k - o));
        t[k] *= r;
        s.dinv[c0 + k] = r;                          // (every lane, the same value to the same address: one instruction, no select)
        A[base + k] = t[k];
        if constexpr (PROG) { asm volatile("" ::: "memory"); __hip_atomic_store(&s.pad0_, (Kt << 5) + k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }      // (a wave's LDS operations execute in order: whoever reads this count finds the step's stores)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 16; ++c) bv[c] = bn[c];
    }
    return r < 1.7976931348623157e308;               // false for NaN and for +inf: a pivot that was not positive and finite reaches every later root (through l and d), the panel's last one included
}
// The rows of a full panel past the chain wave's 64 (K > 10: up to 128 rows): a second wave, a row per lane, LEFT looking and one step behind.  It needs nothing
// from the chain wave's registers: row k of the diagonal tile and the reciprocal pivot are in LDS once the chain wave has posted step k.  The count is read FIRST
// and the operands behind it in the same round trip (LDS executes a wave's operations in order; the chain wave's stores of a step precede its count): a count
// that is too small repeats the round.  Per column k terms, contiguous operands -- about half the chain wave's instructions per step, so it keeps up.
template <class PTR>
__device__ __forceinline__ void rowwave_panel_second(PTR A, const int Kt, const int o, const int R, StepShared& s, const int lane) {
    const int c0 = Kt << 4;
    const int rc = min(c0 + o + 64 + lane, R - 1);
    const int base = tl_base(rc >> 4, Kt) + (rc & 15) * TILE_RS;
    const int db = tl_base(Kt, Kt);
    double t[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) t[c] = A[base + c];
    // (one step ahead: the operands of step k + 1 are requested before step k's arithmetic -- behind the chain wave they are there already, and the round trip
    //  disappears from the step; a count that turns out too small repeats the request where it is needed)
    // (measured and not kept: the operands of step k + 1 requested a step ahead, with and without partial sums -- 56.3 -> 58.3 / 59.5 k ticks at D = 127: this
    //  wave is then ahead of the chain wave more often, and every round it repeats is LDS traffic in the chain wave's way)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        double b[16], r;
        for (;;) {
            const int pv = __hip_atomic_load(&s.pad0_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (a ds_read: in order with the loads behind it)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int m = 0; m < k; ++m) b[m] = A[db + k * TILE_RS + m];
            r = s.dinv[c0 + k];
            asm volatile("" ::: "memory");
            if (__builtin_amdgcn_readfirstlane(pv) >= (Kt << 5) + k + 1) break;
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int m = 0; m < k; ++m) t[k] = fma(-t[m], b[m], t[k]);
        t[k] *= r;
        A[base + k] = t[k];
    }
}
template <int SLOTS = 3, class PTR, class PRE = NoPre>
__device__ __forceinline__ bool chol_rowwave(PTR A, int D, StepShared& s, PRE pre = PRE()) {
    const int t = vil_tid(), NT = blockDim.x, wave = t >> 6, lane = t & 63, NW = NT >> 6;
    const int R = D + 1, T = (R + 15) >> 4, TD = (D + 15) >> 4;
    const int la = (lane & 15) * TILE_RS + (lane >> 4);     // operand element (row lane&15, k lane>>4) inside a tile
    const int lc = (lane >> 4) * TILE_RS + (lane & 15);     // accumulator element (row lane>>4 (+4g), col lane&15)
    const int ntile_all = (T * (T + 1)) >> 1;
    // the panel waves are the last ones (one; two when the first panel has more than 64 rows -- K > 10); the tiles are dealt to the others exactly as chol_lookahead
    // deals them (`pre` sees the same slots), and the tile wave(s) that share a SIMD with the panel wave(s) sit out when the others have slots for every tile
    const int npw = R - 4 > 64 ? 2 : 1;
    const int NWP = NW - npw;                              // the chain wave; NWP + 1: the second panel wave
    const int n_idle = (ntile_all <= (NWP - npw) * SLOTS && NWP >= 4) ? npw : 0;
    const int NWT = NWP - n_idle;
    const bool idle_w = wave < NWP && wave >= NWP - 4 && wave < NWP - 4 + n_idle;
    const int wr = wave - (wave >= NWP - 4 + n_idle && wave < NWP ? n_idle : 0);
    d4 Creg[SLOTS]; int tIJ[SLOTS];
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
        const int g = idle_w ? ntile_all : (wave < NWP ? wr + NWT * u : NWT * SLOTS + (wave - NWP) + npw * u);
        tIJ[u] = -1;
        if (g < ntile_all) {
            const int I = s.tI[g], J = s.tJ[g];
            tIJ[u] = (I << 8) | J;
            const int cb = tl_base(I, J) + lc;
#pragma unroll
            for (int q = 0; q < 4; ++q) Creg[u][q] = A[cb + q * (4 * TILE_RS)];
        }
    }
    pre(Creg, tIJ);
    if (t == 0) { s.cok = 1; s.pad0_ = 0; }
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) if (tIJ[u] >= 0 && (tIJ[u] & 255) == 0) {
        const int cb = tl_base(tIJ[u] >> 8, 0) + lc;
#pragma unroll
        for (int q = 0; q < 4; ++q) A[cb + q * (4 * TILE_RS)] = Creg[u][q];
    }
    lds_barrier();
    auto update = [&](const int u, const int Kt) {          // tile of slot u -= X_I X_J^T over the 16 columns of tile column Kt
        const int I = tIJ[u] >> 8, J = tIJ[u] & 255;
        const int ab = tl_base(I, Kt) + la, bb = tl_base(J, Kt) + la;
        double a_[4], b_[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a_[q] = A[ab + 4 * q]; b_[q] = A[bb + 4 * q]; }
        // (two accumulators: the four matrix instructions of a tile are two dependent pairs instead of a chain of four)
        d4 c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_[0], b_[0], Creg[u], 0, 0, 0), c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_[1], b_[1], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_[2], b_[2], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a_[3], b_[3], c1, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) Creg[u][q] = c0[q] + c1[q];
    };
    for (int Kt = 0; Kt < TD; ++Kt) {
        const int o = Kt == 0 ? 4 : 0;
        const bool two = R - (Kt << 4) - o > 64;             // (more rows than the chain wave has lanes: only in full panels)
        if (wave == NWP) {
            bool ok;
            if (two) ok = Kt == 0 ? rowwave_panel_full<true, true>(A, 0, R, s, lane) : rowwave_panel_full<false, true>(A, Kt, R, s, lane);
            else ok = Kt == 0 ? rowwave_panel_full<true, false>(A, 0, R, s, lane) : (D - (Kt << 4) >= 16 ? rowwave_panel_full<false, false>(A, Kt, R, s, lane) : D - (Kt << 4) == 3 ? rowwave_panel<false, false, 3>(A, Kt, D, R, s, lane) : D - (Kt << 4) == 15 ? rowwave_panel<false, false, 15>(A, Kt, D, R, s, lane) : rowwave_panel<false, false>(A, Kt, D, R, s, lane));
            if (!ok && lane == 0) s.cok = 0;
        } else if (two && wave == NWP + 1) rowwave_panel_second(A, Kt, o, R, s, lane);
        lds_barrier();                                       // tile column Kt is L
        if (Kt + 1 >= TD) break;
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) if (tIJ[u] >= 0 && (tIJ[u] & 255) == Kt + 1) {       // the next tile column first, and to LDS
            update(u, Kt);
            const int cb = tl_base(tIJ[u] >> 8, Kt + 1) + lc;
#pragma unroll
            for (int q = 0; q < 4; ++q) A[cb + q * (4 * TILE_RS)] = Creg[u][q];
        }
        lds_barrier();                                       // the chain wave goes on; the columns right of it are due one barrier later
#pragma unroll
        for (int u = 0; u < SLOTS; ++u) if (tIJ[u] >= 0 && (tIJ[u] & 255) > Kt + 1) update(u, Kt);
    }
    return s.cok != 0;
}
// K <= 20: the row-per-lane panels; otherwise the look-ahead factorisation
template <int SLOTS, class PTR, class PRE = NoPre>
__device__ __forceinline__ bool chol_dense(PTR A, int D, StepShared& s, PRE pre = PRE()) {
    if (D >= 16 && D + 1 <= 132 && (blockDim.x >> 6) >= 8) return chol_rowwave<SLOTS>(A, D, s, pre);
    return chol_lookahead<SLOTS, false>(A, D, s, pre);
}

// back substitution L^T x = y, a COLUMN per lane: the counterpart of chol_rowwave's panels.  From the last diagonal tile up, one wave: lane (g, j) owns column j of
// tile column blk - g -- the diagonal tile's column (g = 0) and the same sixteen rows of the three tiles left of it -- as sixteen registers.  Row by row from the
// bottom, x_i = y_i / L_ii is lane i's; one v_readlane pair makes it a uniform operand, and ONE fma per lane takes l_ij x_i off the y of all 64 columns: the
// diagonal block's recurrence and the fold into the three blocks left of it are the same instruction.  No inverse of the diagonal tiles (back_subst spends
// ~5000 ticks on forming them before its first x), four instructions per row on the chain.  Tiles further left than three (one tile at K = 10) are folded by
// the other waves behind the block's barrier into a separate accumulator that a column picks up when its own block's turn comes -- never on the chain.
// A is not modified.  Result in s.y[0 .. D); s.xs is scratch.
template <class PTR>
__device__ __forceinline__ void back_subst_cols(PTR A, int D, StepShared& s) {
    const int t = vil_tid(), NT = blockDim.x, wave = t >> 6, lane = t & 63;
    const int TD = (D + 15) >> 4;
    for (int i = t; i < (TD << 4); i += NT) { s.y[i] = i < D ? A[tl_idx(D, i)] : 0.0; s.xs[i] = 0.0; }
    const int g = lane >> 4, j = lane & 15;
    // a block's columns: loaded, masked (the diagonal tile at or above its diagonal, rows past the matrix: not L) and -- the diagonal block's -- scaled by 1 / L_jj
    // (entries and y alike: x_i is then lane i's value as it stands, a row of the recurrence is readlane + fma).
    // (measured and not kept: two rows per broadcast round with the sub-diagonal entry as a uniform operand, 7.6 -> 7.9 k ticks at D = 67; block blk - 1 prepared
    //  in front of block blk's recurrence, 7.6 -> 8.3 k)
    auto prepare = [&](const int blk, double* l, double& sc) {
        const int nb = min(16, D - (blk << 4)), cb = blk - g, ccb = max(cb, 0), col = (ccb << 4) + j;
        const bool valid = cb >= 0;
        const int tb = tl_base(blk, ccb) + j;
#pragma unroll
        for (int i = 0; i < 16; ++i) l[i] = A[tb + i * TILE_RS];
        sc = g == 0 ? s.dinv[min(col, D - 1)] : 1.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { double v = l[i]; asm volatile("" : "+v"(v)); l[i] = (valid && i < nb && (g > 0 || i > j)) ? v * sc : 0.0; }
    };
    __syncthreads();
    for (int blk = TD - 1; blk >= 0; --blk) {
        const int nb = min(16, D - (blk << 4));
        if (wave == 0) {
            const int cb = blk - g, col = (max(cb, 0) << 4) + j;
            double l[16], sc;
            prepare(blk, l, sc);
            double yv = s.y[col];
            if (g == 0) yv = (yv - s.xs[col]) * sc;
            // (x_j is lane j's own value once the rows below j have been taken off: nothing is stored inside the recurrence -- a store per row would put an
            //  LDS wait into every step)
            if (nb == 16) {
#pragma unroll
                for (int i = 15; i >= 0; --i) yv = fma(-l[i], bcast_lane(yv, i), yv);
            } else {
#pragma unroll
                for (int i = 15; i >= 0; --i) if (i < nb) yv = fma(-l[i], bcast_lane(yv, i), yv);      // (wave-uniform)
            }
            if (cb >= 0) s.y[col] = yv;
        }
        __syncthreads();
        if (wave > 0 && blk >= 4) {                           // tile columns 0 .. blk - 4: a column per thread
            for (int c = t - 64; c < ((blk - 3) << 4); c += NT - 64) {
                const int cb = tl_base(blk, c >> 4) + (c & 15);
                double acc = 0.0;
                for (int i = 0; i < nb; ++i) acc = fma(A[cb + i * TILE_RS], s.y[(blk << 4) + i], acc);
                s.xs[c] += acc;
            }
        }
    }
}

// back substitution L^T x = y (y = row D of A) with 16 x 16 diagonal blocks.
//   (I)  one WAVE per diagonal tile (no workgroup barrier inside): W_t = L_tt^-1 by the column recurrence with FOUR lanes per column -- lane (j, p) holds
//        the entries L[i][4m + p] of its quarter (36 loads, one round trip) and the w_k with k = p mod 4, a row's partial sums meet in two DPP quad
//        steps -- then W_t^T REPLACES the diagonal tile (row j = zeros, 1 / L_jj, W_ij for i > j), and N_t = W_t L_{t,t-1} (four matrix-core
//        instructions) REPLACES the tile left of it.  (One thread per column, as before: 120 loads and their waits in a row; about the same ~5000 cycles for the stage, but no barrier inside it.)
//   (II) from the last tile up, ONE barrier per tile: wave 0 completes y_blk (its own share of the folds is a register), shares it through LDS, and forms
//        from the same sixteen values BOTH x_blk = W^T y_blk and N^T y_blk = L_{blk,blk-1}^T x_blk, what x_blk takes off the NEXT block's y -- the only
//        fold the next stage waits for.  Behind the barrier the other waves fold x_blk into the columns
//        left of that block, one column per thread, while wave 0 is already in the next stage; their sums are due one barrier later.
//        (Before: x, barrier, every thread folds, barrier: 1200 - 1700 cycles per tile against ~1000.  Splitting x and the fold over two groups of lanes
//        with the fold passed through LDS, operands loaded ahead of the barrier, was slower again: the tile is bound by the LDS traffic of its one wave.)
// Result in s.y[0..D).
template <class PTR>
__device__ __forceinline__ void back_subst(PTR A, int D, StepShared& s) {
    const int t = vil_tid(), NT = blockDim.x, wave = t >> 6, lane = t & 63;
    for (int i = t; i < D; i += NT) s.y[i] = A[tl_idx(D, i)];
    const int TD = (D + 15) >> 4;                           // diagonal tiles that hold rows of L
    for (int i = D + t; i < (TD << 4); i += NT) s.y[i] = 0.0;                        // padding of the last tile: its products vanish
    // ---- (I) ----------------------------------------------------------------------------------------------------
    {
        const int j = lane >> 2, p = lane & 3;
        const int lc = (lane >> 4) * TILE_RS + (lane & 15);     // accumulator element (row lane>>4 (+4g), col lane&15)
        for (int tt = wave; tt < TD; tt += NT >> 6) {           // wave-uniform
            const int n_t = min(16, D - (tt << 4));
            const int tb = tl_base(tt, tt);
            const double* dv = s.dinv + (tt << 4);
            double dr[16], lq[16][4], wq[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < 16; ++i) dr[i] = dv[i];     // (beyond the matrix: whatever the array holds -- masked below)
#pragma unroll
            for (int i = 1; i < 16; ++i)
#pragma unroll
                for (int m = 0; m < (i + 3) / 4; ++m) lq[i][m] = A[tb + i * TILE_RS + 4 * m + p];
#pragma unroll
            for (int i = 1; i < 16; ++i)
#pragma unroll
                for (int m = 0; m < (i + 3) / 4; ++m) { double v = lq[i][m]; asm volatile("" : "+v"(v)); lq[i][m] = (4 * m + p < i) ? v : 0.0; }      // (at or right of the diagonal: not L)
            const bool col = j < n_t;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int m = 0; m < (i + 3) / 4; ++m) { if (m & 1) a1 = fma(lq[i][m], wq[m], a1); else a0 = fma(lq[i][m], wq[m], a0); }
                double sum = a0 + a1;
                if (i > 0) { sum = dpp_add<0xB1, 0xf>(sum); sum = dpp_add<0x4E, 0xf>(sum); }     // quad_perm [1,0,3,2], [2,3,0,1]: the four quarters of the row, the same bits in the four lanes
                double off = -sum * dr[i], on = dr[i];
                asm volatile("" : "+v"(off), "+v"(on));        // both arms exist before the selects: nothing for the compiler to sink into a branch
                const bool live = col && i < n_t;
                const double wi = (live && i > j) ? off : ((live && i == j) ? on : 0.0);
                wq[i >> 2] = ((i & 3) == p) ? wi : wq[i >> 2];
            }
            // (a wave's LDS operations execute in order and every lane's loads above have been consumed: the tile can be overwritten)
#pragma unroll
            for (int m = 0; m < 4; ++m) A[tb + j * TILE_RS + 4 * m + p] = wq[m];      // W_ij at (j, i)
            if (tt > 0) {                                    // N = W L_{t,t-1}: N[i][c] = sum_r W[i][r] L[16 t + r][16 (t - 1) + c]
                const int nb = tl_base(tt, tt - 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                double av[4], bv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int o = (4 * q + (lane >> 4)) * TILE_RS + (lane & 15); av[q] = A[tb + o]; bv[q] = A[nb + o]; }
                d4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double b_ = bv[q]; asm volatile("" : "+v"(b_));
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], (4 * q + (lane >> 4) < n_t) ? b_ : 0.0, c, 0, 0, 0);      // (rows beyond the matrix: W is zero there, L may be anything)
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) A[nb + lc + g * (4 * TILE_RS)] = c[g];
            }
        }
    }
    lds_barrier();
#ifdef VIL_STAMPS
    if (t == 0) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); s.tacc[0] = tt_; }
#endif
    // ---- (II) ---------------------------------------------------------------------------------------------------
    double crit = 0.0;                                      // wave 0, lane tt: what x of the tile just solved takes off y[kb + tt]
    for (int blk = TD - 1; blk >= 0; --blk) {
        const int kb = blk << 4;
        if (t < 64) {                                         // wave 0 (scalar branch); lanes 16..63 mirror lanes 0..15 (sixteen active lanes only: no faster)
            const int tt = t & 15;
            const int tb = tl_base(blk, blk) + tt * TILE_RS;
            const int nb = tl_base(blk, max(blk - 1, 0)) + tt;          // column tt of N (blk = 0: unused)
            double wr[16], nn[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { wr[i] = A[tb + i]; nn[i] = A[nb + i * TILE_RS]; }      // independent of y: in flight during the round trip below
            const double yo = s.y[kb + tt] - crit;
            s.y[kb + tt] = yo;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (its LDS operations execute in order: the reads below see every lane's store)
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
#pragma unroll
            for (int i = 0; i < 16; i += 4) {                  // row tt of W^T (zeros left of the diagonal and beyond the matrix) and column tt of N, times y_blk
                const double y0 = s.y[kb + i], y1 = s.y[kb + i + 1], y2 = s.y[kb + i + 2], y3 = s.y[kb + i + 3];
                a0 = fma(wr[i], y0, a0); a1 = fma(wr[i + 1], y1, a1); a2 = fma(wr[i + 2], y2, a2); a3 = fma(wr[i + 3], y3, a3);
                c0 = fma(nn[i], y0, c0); c1 = fma(nn[i + 1], y1, c1); c2 = fma(nn[i + 2], y2, c2); c3 = fma(nn[i + 3], y3, c3);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            s.y[kb + tt] = (a0 + a1) + (a2 + a3);             // x (rows beyond the matrix: W^T is zero there, x = 0)
            crit = (c0 + c1) + (c2 + c3);
        }
        lds_barrier();
        if (t >= 64) {
            // columns left of the next block: y_c -= sum_r L[kb + r][c] x_r, due at the NEXT barrier (wave-uniform loop bounds: a wave whose columns all lie
            // right of the limit stays off the LDS pipe)
            const int lim = kb - 16;
            for (int w0 = (t & ~63) - 64; w0 < lim; w0 += NT - 64) {
                const int c = w0 + (t & 63), cc = min(c, lim - 1);
                const int base = tl_base(blk, cc >> 4) + (cc & 15);
                double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    v0 = fma(A[base + r * TILE_RS], s.y[kb + r], v0);
                    v1 = fma(A[base + (r + 1) * TILE_RS], s.y[kb + r + 1], v1);
                    v2 = fma(A[base + (r + 2) * TILE_RS], s.y[kb + r + 2], v2);
                    v3 = fma(A[base + (r + 3) * TILE_RS], s.y[kb + r + 3], v3);
                }
                double yn = s.y[cc] - ((v0 + v1) + (v2 + v3));
                asm volatile("" : "+v"(yn));
                if (c < lim) s.y[cc] = yn;
            }
        }
    }
    lds_barrier();
}

// ---- chain path (vil_chain.hpp): pack the pose part, eliminate the speed-bias chain from both ends, Schur-update the pose
//      tiles on the matrix cores, dense Cholesky + back substitution on 6K + 8 rows, back-substitute the chain.
// In: s.sc, s.dcs, s.y (= u, for the camera share of u^T H u), s.gd.  Out: solution of M x = rhs in s.y[0 .. D); qpart gets
// this thread's share of u_c^T S' u_c.  Returns false (uniformly) when a pivot is not positive.
struct ChainSrcStep {                      // the chain's view of the system inside the step kernel: S' in global memory, scales in LDS
    const double* S; int D; const double* sc_; const double* dcs_; const double* gd_; const double* u_; double mu;
    __device__ __forceinline__ double u(int i) const { return u_[i]; }
    int NP;
    __device__ __forceinline__ double raw(int i, int j) const { return S[(size_t)i * D + j]; }
    __device__ __forceinline__ double diag(int k, int i, int j) const { return raw(NP + 9 * k + i, NP + 9 * k + j); }
    __device__ __forceinline__ double sub(int k, int kn, int q, int c) const { return raw(NP + 9 * kn + q, NP + 9 * k + c); }
    __device__ __forceinline__ double prow(int r, int k, int c) const { return raw(r, NP + 9 * k + c); }
    __device__ __forceinline__ double sc(int j) const { return sc_[j]; }
    __device__ __forceinline__ double madd(int j) const { const double d = dcs_[j]; return mu * d * d; }
    __device__ __forceinline__ double rowscale(int r) const { return sc_[r]; }
    __device__ __forceinline__ double rhsraw(int j) const { return gd_[j]; }
    __device__ __forceinline__ void row_done(int, int r, double zr, double& q) const { q += 2.0 * u_[r] * zr; }
    __device__ __forceinline__ void wput(double* p, double v) const { *p = v; }
};

template <bool WLDS, class PUB, class SIDE>
__device__ __forceinline__ bool solve_chain(const DevP& P, const SysBuf& sb, StepShared& s, double* lds, const double mu, const bool cam, double& qpart, PUB pub, SIDE side) {
    const int t = vil_tid();
#ifdef VIL_STAMPS
    #define SSTAMP(k) do { if (t == 0) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); P.dbg[40 + k] = tt_; } } while (0)
#else
    #define SSTAMP(k) do {} while (0)
#endif
    SSTAMP(0);
    const int K = P.K, D = P.D, NP = P.NV, R = NP + 1, T = (R + 15) >> 4, ntile = (T * (T + 1)) >> 1;
    const int RS = P.chain_rs, NB = 9 * K, m = K >> 1;
    double* Tl = lds;
    double* Wt = WLDS ? lds + ntile * TILE_SZ : P.M;                       // W^T: column j of the chain at Wt[j * RS + row]
    const ChainLds L = chain_lds(lds + ntile * TILE_SZ + (WLDS ? (size_t)chain_wcols(K) * RS : 0), K);
    if (t < 8) chain_flag_set(L.flag + t, 0);
    __syncthreads();
    SSTAMP(1);
    if (t < 384) {
        const ChainSrcStep src{sb.S, D, s.sc, s.dcs, s.gd, s.y, mu, NP};
        double qc = 0.0;
        if (cam) chain_eliminate<true>(src, K, NP, RS, Wt, L, qc, P.dbg); else chain_eliminate<false>(src, K, NP, RS, Wt, L, qc, P.dbg);
        qpart += qc;
    } else {
        // waves 6, 7 meanwhile: the pose tiles M_pp = Sc S'_pp Sc + mu dc^2 (+ rhs row) with their share of u^T S' u; four
        // elements per round so that the loads are in flight together
        const int pl = t - 384, NE = ntile << 8;
        for (int e0 = pl; e0 < NE; e0 += 4 * 128) {
            double v[4]; int ii[4], jj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(e0 + 128 * u, NE - 1), tile = e >> 8, w = e & 255;
                ii[u] = (s.tI[tile] << 4) + (w >> 4); jj[u] = (s.tJ[tile] << 4) + (w & 15);
                v[u] = sb.S[(size_t)min(ii[u], NP - 1) * D + min(jj[u], NP - 1)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 128 * u;
                if (e >= NE) break;
                const int i = ii[u], j = jj[u];
                double mv = 0.0;
                if (i < NP && j <= i) {
                    if (cam) qpart += (i == j ? 1.0 : 2.0) * s.y[i] * v[u] * s.y[j];
                    mv = s.sc[i] * v[u] * s.sc[j];
                    if (i == j) mv += mu * s.dcs[i] * s.dcs[i];
                } else if (i == NP && j < NP) mv = s.sc[j] * s.gd[j];
                Tl[tl_phys(e)] = mv;
            }
        }
#ifdef VIL_STAMPS
        if (t == 384) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); P.dbg[57] = tt_; P.dbg[58] = tt_; }
#endif
    }
    __syncthreads();
    SSTAMP(2);
    if (chain_flag_get(L.flag + 5)) return false;
    SSTAMP(3);
    // ---- S_pp -= W W^T on the matrix cores, straight into the accumulators of the blocked Cholesky; dense part -------------
    auto schur = [&](d4* Creg, const int* tIJ) {
        const int lane = t & 63, row = lane & 15, kq = lane >> 4;
        if (P.skip_mask & 128) return;                 // (timing probe only)
        // eight k-steps (32 chain columns) per round: all sixteen operand loads are issued before the first MFMA, so the
        // LDS / L2 latency is paid once per round instead of once per MFMA.  Addresses are `lane base + wave-uniform offset`
        // (32-bit): W^T is padded to a multiple of 32 columns, nothing is clamped per lane.
        const int RS4 = 4 * RS;
#pragma unroll
        for (int u = 0; u < CH_SLOTS; ++u) {
            if (tIJ[u] < 0) continue;
            const int I = tIJ[u] >> 8, J = tIJ[u] & 255;
            const bool va = (I << 4) + row < R, vb = (J << 4) + row < R;
            const double* pa = Wt + kq * RS + (I << 4) + row;
            const double* pb = Wt + kq * RS + (J << 4) + row;
            d4 c4 = Creg[u];
            for (int kk = 0; kk < NB; kk += 32) {
                const double* qa = pa + kk * RS; const double* qb = pb + kk * RS;
                double av[8], bv[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) { av[g] = qa[g * RS4]; bv[g] = qb[g * RS4]; }
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const bool kv = kk + 4 * g + kq < NB;
                    c4 = __builtin_amdgcn_mfma_f64_16x16x4f64((va && kv) ? -av[g] : 0.0, (vb && kv) ? bv[g] : 0.0, c4, 0, 0, 0);
                }
            }
            Creg[u] = c4;
        }
    };
    if (!chol_lookahead<CH_SLOTS, false>(Tl, NP, s, schur)) return false;
    SSTAMP(4);
    back_subst(Tl, NP, s);                             // x_p in s.y[0 .. NP)
    pub();                                             // the landmark workgroups can start: they only need the pose part
    SSTAMP(5);
    // ---- chain back substitution: t = y_b - W^T x_p, then outwards from the middle block -----------------------------------
    double* tB = L.tB;
    {
        const int G = (4 * NB <= VIL_STEP_THREADS) ? 4 : 2;
        const int j = t / G, part = t - j * G;
        const int jc = min(j, NB - 1);
        double acc = 0.0;
        for (int rr = part; rr < NP; rr += G) acc += Wt[(size_t)jc * RS + rr] * s.y[rr];
        acc += __shfl_xor(acc, 1, 64);
        if (G == 4) acc += __shfl_xor(acc, 2, 64);
        if (j < NB && part == 0) tB[j] = Wt[(size_t)j * RS + NP] - acc;
    }
    __syncthreads();
    if (t < 64) chain_block_back(L.Ldg + 54 * m, nullptr, tB + 9 * m, nullptr, s.y + NP + 9 * m);
    __syncthreads();
    if (t < 64) { for (int k = m - 1; k >= 0; --k) chain_block_back(L.Ldg + 54 * k, L.Lsb + 82 * k, tB + 9 * k, s.y + NP + 9 * (k + 1), s.y + NP + 9 * k); }
    else if (t < 128) { for (int k = m + 1; k < K; ++k) chain_block_back(L.Ldg + 54 * k, L.Lsb + 82 * k, tB + 9 * k, s.y + NP + 9 * (k - 1), s.y + NP + 9 * k); }
    else if (t >= VIL_STEP_THREADS - 64) side();       // a spare wave: whatever the caller can hide behind the two chain walks
    __syncthreads();
    SSTAMP(6);
    return true;
}

// ---- chain eliminated ahead by the extra workgroup of k_sweep, W W^T contracted by k_reduce (vil_prechain.hpp): pack the pose tiles as
//      M_pp = Sc (S'_pp - W W^T) Sc + mu d^2 (the row scaling the chain workgroup deferred is applied here), dense part, chain back
//      substitution.  Same contract as solve_chain.
template <bool RW /* the row-per-lane factorisation (chol_rowwave) where it applies: the one-launch iteration */, class PUB, class SIDE>
__device__ __forceinline__ bool solve_prechain(const DevP& P, const SysBuf& sb, StepShared& s, double* lds, const double mu, const bool cam, double& qpart, const int epoch /* of this launch's flags */, PUB pub, SIDE side) {
    const int t = vil_tid();
    SSTAMP(0);
    const int K = P.K, D = P.D, NP = P.NV, R = NP + 1, T = (R + 15) >> 4, ntile = (T * (T + 1)) >> 1;
    const int RS = P.chain_rs, NB = 9 * K, m = K >> 1;
    double* Tl = lds;
    const double* Wt = P.chW;
    double* Ldg = lds + ntile * TILE_SZ; double* Lsb = Ldg + 54 * K; double* tB = Lsb + 82 * K;
    {   // pose tiles (+ rhs row) from S' alone, their share of u^T S' u -- the chain / tile workgroups of this launch are usually still at work
        const int NE = ntile << 8;
        for (int e0 = t; e0 < NE; e0 += 8 * VIL_STEP_THREADS) {
            double v[8]; int ii[8], jj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = min(e0 + VIL_STEP_THREADS * u, NE - 1), tile = e >> 8, w = e & 255;
                ii[u] = (s.tI[tile] << 4) + (w >> 4); jj[u] = (s.tJ[tile] << 4) + (w & 15);
                v[u] = ld_ag(sb.S + (size_t)min(ii[u], NP - 1) * D + min(jj[u], NP - 1));
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + VIL_STEP_THREADS * u;
                if (e >= NE) break;
                const int i = ii[u], j = jj[u];
                double mv = 0.0;
                if (i < NP && j <= i) {
                    if (cam) qpart += (i == j ? 1.0 : 2.0) * s.y[i] * v[u] * s.y[j];
                    mv = s.sc[i] * v[u] * s.sc[j];
                    if (i == j) mv += mu * s.dcs[i] * s.dcs[i];
                } else if (i == NP && j < NP) mv = s.sc[j] * s.gd[j];
                Tl[tl_phys(e)] = mv;
            }
        }
    }
    if (P.rs_merged) {   // the chain workgroup and the W W^T tile workgroups of this launch are done (a tile's flag implies the chain's)
        for (int i = t; i < P.n_ww; i += VIL_STEP_THREADS) spin_until_eq(P.wwflag + i, epoch, P.abortf);
        __syncthreads();                               // (everything they left is read at agent scope below: no fence)
    }
    if (t == 0) prof_stamp(P, epoch - 1, 9);
    SSTAMP(1);
    // (the chain workgroup's last flag -- factors for the chain back substitution, written ~4 us after W -- rides in the round trip of the W W^T loads:
    //  a thread that sees it posted here needs neither a poll nor a barrier after the dense part)
    const int f2 = (P.rs_merged || P.prechain == 2) ? ld_ag(P.chflag + 2) : 0;
    {   // M_pp -= Sc (W W^T) Sc: each thread on the very elements it packed (same index map: no barrier in between)
        const int NE = ntile << 8;
        for (int e0 = t; e0 < NE; e0 += 8 * VIL_STEP_THREADS) {
            double ww[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ww[u] = ld_ag(P.chWW + tl_phys(min(e0 + VIL_STEP_THREADS * u, NE - 1)));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + VIL_STEP_THREADS * u;
                if (e >= NE) break;
                const int tile = e >> 8, w = e & 255, i = (s.tI[tile] << 4) + (w >> 4), j = (s.tJ[tile] << 4) + (w & 15);
                if (i <= NP && j <= i && j < NP) Tl[tl_phys(e)] -= (i < NP ? s.sc[i] : 1.0) * ww[u] * s.sc[j];
            }
        }
        if (cam) {                                     // chain share of u^T S' u: chain x chain + 2 u_p . (S'_pb u_b)
            if (t < NP) qpart += 2.0 * s.y[t] * (ld_ag(P.chZ + t) + ld_ag(P.chZ + R + t));
            if (t == 0) qpart += ld_ag(P.chQ) + ld_ag(P.chQ + 1);
        }
        if (t == 0 && !ld_ag(P.chOk)) s.ok = 0;
    }
    __syncthreads();
    SSTAMP(2); SSTAMP(3);
    if (!s.ok) return false;
    // W^T for the chain back substitution (t = y_b - W^T x_p) does not depend on x_p: its loads are issued here and land under the first block steps of the
    // factorisation, whose barriers order LDS only (lds_barrier) and do not wait for them
    const int G = (4 * NB <= VIL_STEP_THREADS) ? 4 : 2;
    const int j = t / G, part = t - j * G;
    const int jc = min(j, NB - 1);
    double wv[24], wrhs = 0.0;
    const bool small = ntile <= 24;                    // K <= 12: three register tiles per wave in the factorisation instead of seven -- room for the prefetch
    if (small) {
        // (waves whose threads all sit past the last chain column -- the last two at K = 10, the panel wave of the factorisation among them -- ask for nothing: their 25
        //  requests per lane queued in front of the factorisation's first panel at the compute unit's 5 lanes per ns)
        const bool wave_has_columns = __builtin_amdgcn_readfirstlane(((t >> 6) << 6) / G) < NB;      // (a scalar branch)
#pragma unroll
        for (int q = 0; q < 24; ++q) wv[q] = 0.0;
        if (wave_has_columns) {
#pragma unroll
            for (int q = 0; q < 24; ++q) wv[q] = ld_ag(Wt + (size_t)jc * RS + min(part + q * G, NP - 1));
            wrhs = ld_ag(Wt + (size_t)jc * RS + NP);
        }
        if (!(RW ? chol_dense<3>(Tl, NP, s) : chol_lookahead<3, false>(Tl, NP, s))) return false;
    } else if (!(RW ? chol_dense<CH_SLOTS>(Tl, NP, s) : chol_lookahead<CH_SLOTS, false>(Tl, NP, s))) return false;
    SSTAMP(4);
    if (t == 0) prof_stamp(P, epoch - 1, 10);
    if (RW) back_subst_cols(Tl, NP, s); else back_subst(Tl, NP, s);
    pub();
    if (t == 0) prof_stamp(P, epoch - 1, 11);
    SSTAMP(5);
    // ---- chain back substitution.  What it needs from the chain workgroup (inverses of the factored diagonal blocks, sub-diagonal blocks, W^T) in ONE
    //      round trip: every load of a thread in flight together
    {
        if (P.rs_merged || P.prechain == 2) {          // (prechain 2: prechain_inverses, a workgroup of this launch)
            if (f2 != epoch) spin_until_eq(P.chflag + 2, epoch, P.abortf);
        }
        double acc = 0.0, pl[2], ps[2];
        const int nr = (NP - part + G - 1) / G;            // rows part, part + G, ... < NP
#pragma unroll
        for (int q = 0; q < 2; ++q) { pl[q] = ld_ag(P.chLdg + min(t + q * VIL_STEP_THREADS, 54 * K - 1)); ps[q] = ld_ag(P.chLsb + min(t + q * VIL_STEP_THREADS, 82 * K - 1)); }
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int e = t + q * VIL_STEP_THREADS; if (e < 54 * K) Ldg[e] = pl[q]; if (e < 82 * K) Lsb[e] = ps[q]; }
        for (int e = t + 2 * VIL_STEP_THREADS; e < 54 * K; e += VIL_STEP_THREADS) Ldg[e] = ld_ag(P.chLdg + e);      // (K > 18)
        for (int e = t + 2 * VIL_STEP_THREADS; e < 82 * K; e += VIL_STEP_THREADS) Lsb[e] = ld_ag(P.chLsb + e);      // (K > 12)
        // t = y_b - W^T x_p with the deferred row scaling (the right-hand-side row of W^T carries y_b)
        if (small) {
#pragma unroll
            for (int q = 0; q < 24; ++q) { const int rr = min(part + q * G, NP - 1); acc += (q < nr ? wv[q] : 0.0) * s.sc[rr] * s.y[rr]; }
            for (int q = 24; q < nr; ++q) { const int rr = part + q * G; acc += ld_ag(Wt + (size_t)jc * RS + rr) * s.sc[rr] * s.y[rr]; }
            acc += __shfl_xor(acc, 1, 64);
            if (G == 4) acc += __shfl_xor(acc, 2, 64);
            if (j < NB && part == 0) tB[j] = wrhs - acc;
        } else {
            // K > 12 (no room for a prefetch across the factorisation, 64 rows per thread in the thread-per-column map: 40 of them were one dependent L2
            // round trip each, 10 us of the 11.6 this stage took at K = 20): a WAVE per chain column -- lanes along the rows of a column of W^T (whole
            // lines; NP <= 127: two rows per lane), eight columns of loads in flight per wave, the column's dot product folded on the DPP crossbar
            const int wave = t >> 6, lane = t & 63;
            const int r0 = min(lane, NP - 1), r1 = min(lane + 64, NP - 1);
            const double x0 = lane < NP ? s.sc[r0] * s.y[r0] : 0.0, x1 = lane + 64 < NP ? s.sc[r1] * s.y[r1] : 0.0;
            for (int j0 = wave; j0 < NB; j0 += 64) {
                double w0[8], w1[8], wr[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double* col = Wt + (size_t)min(j0 + 8 * u, NB - 1) * RS;
                    w0[u] = ld_ag(col + r0); w1[u] = ld_ag(col + r1); wr[u] = ld_ag(col + NP);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double a = wave_total(w0[u] * x0 + w1[u] * x1);
                    if (lane == 0 && j0 + 8 * u < NB) tB[j0 + 8 * u] = wr[u] - a;
                }
            }
        }
    }
    __syncthreads();
    // x_k = L_kk^-T (t_k - Ls_k^T x_next) = c_k - M_k x_next with what the chain workgroup left: the INVERSE of L_kk (Ldg here holds L^-1, 45 entries per block)
    // and M_k = L_kk^-T Ls_k^T (Lsb here, row r at 9 r).  c_k for every block at once -- thread 9 k + r, a nine-term product -- into s.y; then the two
    // directions walk away from the middle block (x_m = c_m) on one wave each: lane r holds component r of the block solved before, the nine components
    // reach every lane as scalars (v_readlane), row r of the next M_k and c_k are requested a block ahead -- no LDS round trip on the recursion
    if (t < NB) {
        const int k = t / 9, r = t - 9 * k;
        const double* Li = Ldg + 54 * k; const double* tk = tB + 9 * k;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int i = 0; i < 9; ++i) { const double li = i >= r ? Li[(i * (i + 1) >> 1) + r] : 0.0; if (i & 1) a1 += li * tk[i]; else a0 += li * tk[i]; }
        s.y[NP + t] = a0 + a1;
    }
    __syncthreads();
    if (t < 128) {
        const int dir = t >> 6, lane = t & 63, r = min(lane, 8), stp = dir == 0 ? -1 : 1, kend = dir == 0 ? -1 : K;
        double xn = s.y[NP + 9 * m + r];
        int k = m + stp;
        double mrow[9], c = 0.0;
        auto fetch = [&](int kk, double* mr, double& cc) {
            const double* Mk = Lsb + 82 * kk + 9 * r;
#pragma unroll
            for (int i = 0; i < 9; ++i) mr[i] = Mk[i];
            cc = s.y[NP + 9 * kk + r];
        };
        if (k != kend) fetch(k, mrow, c);
        for (; k != kend; k += stp) {
            double mnext[9], cnext = 0.0;
#pragma unroll
            for (int i = 0; i < 9; ++i) mnext[i] = 0.0;
            if (k + stp != kend) fetch(k + stp, mnext, cnext);
            const int xlo = __double2loint(xn), xhi = __double2hiint(xn);
            double a0 = c, a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const double xi = __hiloint2double(__builtin_amdgcn_readlane(xhi, i), __builtin_amdgcn_readlane(xlo, i));
                if (i % 3 == 0) a0 = fma(-mrow[i], xi, a0); else if (i % 3 == 1) a1 = fma(-mrow[i], xi, a1); else a2 = fma(-mrow[i], xi, a2);
            }
            xn = a0 + (a1 + a2);
            if (lane < 9) s.y[NP + 9 * k + lane] = xn;
#pragma unroll
            for (int i = 0; i < 9; ++i) mrow[i] = mnext[i];
            c = cnext;
        }
    }
    else if (t >= VIL_STEP_THREADS - 64) side();
    __syncthreads();
    SSTAMP(6);
    return true;
}

}  // namespace vd

// The step kernel is the same on one GPU and on N: in the multi-GPU path the whole linear-system set has been all-reduced
// before it starts (vilsolve.hip: view), so every rank runs it on identical data.
// CHAIN 0: dense factorisation of all D columns (LDSM: tile array in LDS or global).  CHAIN 1 / 2: vil_chain.hpp, with W^T in
// LDS / in global memory (tiles always in LDS).  CHAIN 3: the chain was eliminated by the extra workgroup of k_sweep (vil_prechain.hpp).
//
// Landmarks never enter the step kernel's serial part: per landmark the pass after the solve leaves the two step directions
//   la = Sl gradient_l / dl  (Cauchy direction),  lb = Sl gn_l / dl  (Gauss-Newton direction)
// and six sums (|gn_l|^2, gn_l . g_l, |la|^2, la . lb, |lb|^2, |lambda|^2); the candidate inverse depth lambda + cg la + cn lb is
// formed by the NEXT sweep's visual workgroups from the two dogleg coefficients in Ctl, and its norm follows from the sums.
// With helper workgroups (grid = 1 + n_help) that pass runs on their CUs while the master back-substitutes the chain.
// FUSED: the role runs inside the one-launch iteration (k_iter, vil_iter.hpp) -- the sweep's workgroups are part of the SAME launch: the gather workgroups
// wait for their flags (P.sflag) and read the records at agent scope, the chain workgroup waits for the IMU / prior workgroups', master and helpers read the
// landmark arrays at agent scope, and the master counts the launch in Ctl::n_sweeps itself.  p0: the workgroup's index among the step roles.
// The judge of a candidate that is a STEP (not the first linearisation, not a re-sweep): function tolerance, relative decrease, radius update (trust_region_minimizer.cc;
// SURVEY Appendix B).  One function for the ordinary place (behind the gather) and the persistent solve's cost-first judgement (behind the sweep roles' cost partials).
__device__ __forceinline__ void judge_step(Ctl& c, const double cand_cost, const SolveOpts& O, int& need) {
    const int cand = 1 - c.cur;
    if (fabs(c.cost_cur - cand_cost) <= O.function_tolerance * c.cost_cur) { c.done = 1; c.term = 1; }
    else {
        const double rel = (c.cost_cur - cand_cost) / c.model_change;
        if (isfinite(cand_cost) && rel > O.min_relative_decrease) {
            c.cur = cand; c.cost_cur = cand_cost; c.nsucc++;
            if (rel < 0.25) c.radius *= 0.5;
            if (rel > 0.75) c.radius = fmax(c.radius, 3.0 * c.dogleg_norm);
            c.radius = fmin(O.max_radius, c.radius);
            c.reuse = 0; need = 1;
        } else { c.radius *= 0.5; c.reuse = 1; need = 0; }
    }
    if (c.iter >= 1 && c.iter <= 64) c.cost_trace[c.iter - 1] = c.cost_cur;
}
__device__ __forceinline__ void judge_caps(Ctl& c, const SolveOpts& O) {
    if (!c.done) {
        if (c.iter >= O.max_iterations) { c.done = 1; c.term = 4; }
        else if (c.radius <= 1e-32) { c.done = 1; c.term = 6; c.status = -4; }
    }
}
// would judge_step + judge_caps END the solve?  (no side effects: the persistent solve asks before the gather is complete)
__device__ __forceinline__ bool judge_would_end(const Ctl& c, const double cand_cost, const SolveOpts& O) {
    if (fabs(c.cost_cur - cand_cost) <= O.function_tolerance * c.cost_cur) return true;
    if (c.iter >= O.max_iterations) return true;
    const double rel = (c.cost_cur - cand_cost) / c.model_change;
    const bool accepted = isfinite(cand_cost) && rel > O.min_relative_decrease;
    return !accepted && c.radius * 0.5 <= 1e-32;
}
// the hand-over line of a persistent solve (vil_iter.hpp): what a sweep role reads of Ctl, as seven words {payload, epoch}; lanes 0 .. 6 of the caller
__device__ __forceinline__ void post_iter_header(const DevP& P, const Ctl& c, const int epoch) {
    const int t = vil_tid();
    if (t < 7) {
        const double dv = t <= 2 ? c.mu : (t <= 4 ? c.cg : c.cn);
        const unsigned pl = t == 0 ? (unsigned)((c.cur & 1) | ((c.done ? 1 : 0) << 1) | ((c.first ? 1 : 0) << 2) | ((c.lin_mode & 3) << 3)) : (unsigned)(((t - 1) & 1) ? __double2hiint(dv) : __double2loint(dv));
        __hip_atomic_store(P.ihdr + t, ((unsigned long long)(unsigned)epoch << 32) | pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// duty_item >= 0 (the persistent solve, k_solve): the gather item a helper or tile workgroup of a live iteration takes once it holds Ctl and the epoch, BEFORE its own
// waits -- it has nothing to do until the gather is complete / the chain is eliminated
template <bool LDSM, int CHAIN, bool FUSED>
__device__ __forceinline__ void step_body(const DevP& P, const SolveOpts& O, vd::StepShared& s, double* const Alds, const int p0, const int duty_item = -1, const long long deadline_tick = 0 /* persistent solve: the device's wall clock at which max_solver_time ends (0: no cap) */) {
    using namespace vd;
    const int t = vil_tid(), NT = blockDim.x;
    const int D = P.D, L = P.L;
    // The grid is 1 + P.n_help workgroups: the extra ones run the same judge on their own copy of Ctl and
    // do the two landmark passes of their slice on their own CU (those passes are bound by what ONE CU can pull out of L2).
    // Flags (all compared with the launch epoch): hflag[k] -- helper k has read Ctl and published its pre-pass sums;
    // xflag -- the master has published Sc x_p; xstat -- it has given up for this launch (the helpers poll both at once);
    // hflag2[8 k + w] -- wave w of helper k has left the sums of its second pass (no block reduction on that path).
    // Only the master writes Ctl / the camera candidate, after every helper has signalled hflag.
    // One GPU (P.rs_merged): the gather of the sweep's partial records rides in this launch -- grid = [master | helpers | chain workgroup |
    // W W^T tile workgroups | gather workgroups]; whoever is done posts the launch epoch in its flag, and master, helpers and tile workgroups
    // wait for what they read.  The waiting workgroups have the lowest block indices (dispatched first); everything they wait for is finite.
    const int nhelp = P.n_help;
    const bool merged = FUSED || P.rs_merged != 0;
    // roles by `bid`: 0 master, 1 .. nhelp helpers, then chain, tiles, gather.  The hardware dispatches in blockIdx order and the chain workgroup
    // is the longest path into the dense part, the gather the next: physical order [chain | gather | master | helpers | tiles]
    int bid = p0;
    if (merged) {
        const int pc = P.prechain ? 1 : 0;
        if (pc && p0 == 0) bid = 1 + nhelp;
        else if (FUSED) {
            // one-launch iteration: [chain | master | helpers | tiles | gather] -- the few workgroups that wait for others are resident from the start (their prologues
            // run under the sweep; the device holds them beside the workgroups they wait for: vilsolve.hip), the many gather workgroups take the slots the sweep roles free
            const int q = p0 - pc;
            bid = q <= nhelp ? q : (q - 1 - nhelp < P.n_ww ? 1 + nhelp + pc + (q - 1 - nhelp) : 1 + nhelp + pc + P.n_ww + (q - 1 - nhelp - P.n_ww));
        }
        else if (p0 - pc < P.n_gather) bid = 1 + nhelp + pc + P.n_ww + (p0 - pc);
        else { const int q = p0 - pc - P.n_gather; bid = q <= nhelp ? q : 1 + nhelp + pc + (q - 1 - nhelp); }
    }
#ifdef VIL_STAMPS
    if (bid == 0 && t == 0) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); P.dbg[30] = tt_; }
#endif
    const int b_chain = 1 + nhelp, b_ww = b_chain + ((merged && P.prechain) ? 1 : 0), b_gather = b_ww + (merged ? P.n_ww : 0);
    const bool helper = bid > 0 && bid <= nhelp;
    {   // Ctl (1.5 kB with its traces) into LDS: one coalesced load per lane, not 190 loads of a lone lane with everybody waiting at the barrier
        const double* src = (const double*)P.ctl; double* dst = (double*)&s.c;
        for (int i = t; i < (int)(sizeof(Ctl) / 8); i += NT) dst[i] = ldx<FUSED>(src + i);      // (FUSED: whatever the roles hand from one iteration to the next crosses at agent scope -- the persistent solve, k_solve, has no launch boundary between them)
        if (t == 0) { s.need = 0; s.was_first = 0; s.ok = 1; s.early = 0; s.judged = 0; s.hdr_posted = 0; }
    }
    for (int q = t; q < 256; q += NT) {      // triangular tile index -> (tile row, tile col)
        int Ir = (int)((sqrtf(8.f * (float)q + 1.f) - 1.f) * 0.5f);
        if (((Ir + 1) * (Ir + 2)) / 2 <= q) ++Ir;
        if ((Ir * (Ir + 1)) / 2 > q) --Ir;
        s.tI[q] = (unsigned char)Ir; s.tJ[q] = (unsigned char)(q - (Ir * (Ir + 1)) / 2);
    }
    __syncthreads();
    // (one-launch iteration: nobody else has counted this launch -- every role forms the epoch from the n_sweeps it READ, the master stores the new count)
    const int epoch = (int)((((unsigned)s.c.gen) << 12) + (unsigned)s.c.n_sweeps + 1u);
    const int lidx = epoch - 1;                                      // launch slot of the phase stamps (its low six bits)
    auto PROF = [&](int k, bool mn = false) { if (FUSED && t == 0) prof_stamp(P, lidx, k, mn); };
    __syncthreads();
    if (t == 0) { s.done_at_entry = s.c.done; s.c.swe++; if (FUSED) s.c.n_sweeps++; }      // (every path that writes Ctl back carries the new swe)
    __syncthreads();
#ifdef VIL_STAMPS
    #define STAMP(k) do { __syncthreads(); if (t == 0) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); P.dbg[k] = tt_; } } while (0)
#else
    #define STAMP(k) do {} while (0)
#endif
    if (s.c.done) return;                    // finished in an earlier launch: nobody writes anything
    // completion of a whole workgroup: what it leaves for other workgroups of the launch is stored at agent scope (st_ag), so every thread only
    // waits for its own stores (__syncthreads does not), then one flag; the readers poll relaxed and load at agent scope (ld_ag) -- no fences
    auto rs_signal = [&](int* f) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); if (t == 0) __hip_atomic_store(f, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto rs_wait = [&](const int* f, int n) {
        for (int i = t; i < n; i += NT) spin_until_eq(f + i, epoch, P.abortf);
        __syncthreads();
    };
    if (merged && bid >= b_gather) {
        if constexpr (FUSED) reduce_gather<true, VIL_STEP_THREADS / 8, true>(P, s.c, bid - b_gather, (int4*)Alds, epoch);      // (waits for the sweep workgroups' flags behind its own table staging)
        else if (P.rs_merged == 2) reduce_gather<true, VIL_STEP_THREADS / 8>(P, s.c, bid - b_gather, (int4*)Alds);      // 64 entries, all 512 threads
        else { if (t >= VIL_THREADS) return; reduce_gather<true, RED_EPW>(P, s.c, bid - b_gather, (int4*)Alds); }      // (descriptor table in the dynamic LDS this role does not use otherwise)   // 32 entries on the first four waves
        if (P.drop_role == -2 - (bid - b_gather) && s.c.n_sweeps - 1 == P.drop_launch) return;      // (test hook, vil_debug_drop_flag: this gather workgroup loses its flag in that launch of the solve)
        rs_signal(P.gflag + (bid - b_gather)); PROF(5); return;
    }
    auto duty = [&]() {
        if constexpr (FUSED) if (duty_item >= 0) {
            reduce_gather<true, VIL_STEP_THREADS / 8, true>(P, s.c, duty_item, (int4*)Alds, epoch);
            if (!(P.drop_role == -2 - duty_item && s.c.n_sweeps - 1 == P.drop_launch)) rs_signal(P.gflag + duty_item);
            PROF(5);
            __syncthreads();
        }
    };
    if (merged && bid >= b_ww) {
        duty();
        if (!FUSED && t >= VIL_THREADS) return;        // a 256-thread role: the upper waves leave before the first barrier (one-launch iteration: they idle THROUGH the barriers)
        rs_wait(P.chflag, 1); prechain_ww_tile<FUSED>(P, bid - b_ww, Alds); rs_signal(P.wwflag + (bid - b_ww)); PROF(13); return;
    }
    if (merged && P.prechain && bid == b_chain) { prechain_wg<FUSED>(P, s.c, O.jacobi_scaling, Alds, epoch, FUSED); return; }      // (posts chflag[0 .. 2] itself; one-launch iteration: behind the IMU / prior workgroups' flags)
    if (!merged && P.prechain == 2 && bid == b_chain) { prechain_inverses(P, Alds, epoch); return; }                 // (chain eliminated inside k_sweep: posts chflag[2])
    if (bid == 0) PROF(7);
    if constexpr (FUSED) {
        if (bid == 0) {      // the master collects the sweep roles' flags for the gather workgroups: the other roles first (they finish early), then the visual ones
            const int v0 = P.n_imu + 2, nv = P.n_vwg, n1 = P.n_sw - nv;
            for (int i = t; i < n1; i += NT) spin_until_eq(P.sflag + (i < v0 ? i : i + nv), epoch, P.abortf);
            __syncthreads();
            if (t == 0) st_ag(P.sall + 16, epoch);
            for (int i = t; i < nv; i += NT) spin_until_eq(P.sflag + v0 + i, epoch, P.abortf);
            __syncthreads();
            if (t == 0) st_ag(P.sall + 32, epoch);
            // ---- persistent solve, COST FIRST: every sweep role's cost partial is out.  The master has nothing to do until the gather is complete (~6 us from here): it
            //      adds the partials itself -- the gather's own code and order, the same bits -- and asks whether the judgement will end the solve (function tolerance,
            //      iteration cap, radius underflow).  If so the solve ends HERE: judge, write-out (the host is released), the `done` hand-over line; the gather
            //      workgroups, the helpers and the tile workgroups drain behind it.  What is swept into Jacobians in a solve's last iteration is then never waited for.
            if (P.persist && !s.c.first && !s.c.resweep && s.c.lin_mode == 0) {
                const double cc = gather_cost<VIL_STEP_THREADS / 8, true>(P, s.red);      // (valid in thread 0)
                if (t == 0) s.early = judge_would_end(s.c, cc, O) ? 1 : 0;
                __syncthreads();
                if (s.early) {
                    if (t == 0) { Ctl& c = s.c; c.cand_cost = cc; judge_step(c, cc, O, s.need); judge_caps(c, O); c.outd = 1; s.judged = 1; }
                    __syncthreads();
                    vd::solve_finish<true>(P.x[0], P.x[1], P.xorig, P.hstate, P.ctl, P.hctl, P.hseq, P.K, P.NS, P.gauge_on, s.c.cur, s.c.status, s.c.gen, Alds, &s.c);
                    post_iter_header(P, s.c, epoch);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
                    if (t == 0) st_ag(P.sall + 48, epoch);      // the helpers wake up, read the line and leave
                }
            }
        }
    }
    // master and helpers: the candidate's cost, gradient and diagonal (and S') are complete.  One-launch iteration: the master polls the gather workgroups' flags
    // and passes one word on to the helpers (their first pass is not on the critical path: a hop more, n_help x n_gather polling lanes fewer)
    if (FUSED && merged && bid > 0) {
        duty();
        if (t == 0) {
            spin_until_eq(P.sall + 48, epoch, P.abortf);
            if (P.persist) { const unsigned long long h0 = __hip_atomic_load(P.ihdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s.early = ((unsigned)(h0 >> 32) == (unsigned)epoch && ((unsigned)h0 & 2u)) ? 1 : 0; }      // cost first: the master has ended the solve
        }
        __syncthreads();
        if (s.early) { if (t == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __hip_atomic_store(P.hflag + (bid - 1), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } return; }
    }
    else if (merged && !s.early) { rs_wait(P.gflag, P.n_gather); if (FUSED && t == 0) st_ag(P.sall + 48, epoch); }
    if (bid == 0) PROF(8);
    // Everything the master and its helpers hand each other inside this launch (hpart, hpart2, stepc) is stored AND loaded with agent-scope
    // atomics, i.e. at the level all XCDs share, so a flag only has to be ordered after the poster's own stores (s_waitcnt).  A release
    // fence would also write back the XCD's L2 and an acquire invalidate the reader's: microseconds on the critical path, for data
    // nobody reads before the next kernel.
    auto post = [&](int* f) { if (t == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __hip_atomic_store(f, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } };
    auto wait1 = [&](int* f) { spin_until_eq(f, epoch, P.abortf); };
    auto put = [&](double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // flag-in-data words: 32 bits of payload + the launch epoch in ONE 64-bit store / load (single-copy atomic); epochs never repeat, so a word of
    // this epoch is this launch's value whatever the buffer held before
    auto put_ll = [&](unsigned long long* p, int half) { __hip_atomic_store(p, ((unsigned long long)(unsigned)epoch << 32) | (unsigned)half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ld_ll = [&](const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // wave 0 (all 64 lanes call it): Ctl back to device memory, one coalesced store per lane and round instead of a lone lane's 190.  (The host's
    // copy is written by solve_finish, vil_finish.hpp: the next sweep launch finds the solve finished and leaves Ctl + the final state there.)
    auto store_ctl = [&]() {
        static_assert(sizeof(Ctl) % 8 == 0, "Ctl is copied as doubles");
        const double* src = (const double*)&s.c;
        double* dst = (double*)P.ctl;
        for (int i = t; i < (int)(sizeof(Ctl) / 8); i += 64) stx<FUSED>(dst + i, src[i]);
    };
    // master: every helper has read Ctl (one lane per helper: the polls overlap instead of queueing behind one another)
    bool hseen = false;
    // (... and, gather + step in one launch: the chain / tile workgroups have read Ctl too -- a tile's flag implies the chain's; n_ww <= 45)
    auto wait_helpers = [&]() { if (!hseen && t < nhelp) wait1(P.hflag + t); if (!hseen && merged && P.prechain && t < P.n_ww) wait1(P.wwflag + t); hseen = true; };
    // wave 0: the end of the master's iteration on every path -- every helper (and tile workgroup) has read Ctl, then the new one goes out.  A wait of this launch
    // that gave up (vil_math.hpp: spin_until_eq) ends the solve here with a device error instead of letting numbers formed from incomplete data through
    int abort_pre = -1;            // the abort word as thread 0 read it when the helpers' sums came in (below): the load's round trip runs under the dogleg's scalars instead of at the launch's very end
    auto end_iter = [&]() {
        wait_helpers();
        if (t == 0 && P.abortf && (abort_pre >= 0 ? abort_pre : ld_ag(P.abortf)) != 0) { s.c.done = 1; s.c.term = 6; s.c.status = -2; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (!(FUSED && P.persist)) store_ctl();      // (the persistent solve: Ctl leaves through k_solve's tail, BEHIND the hand-over line of the next iteration's sweep roles)
    };
    const bool cam = true;             // (every rank holds the complete system: nothing is counted per rank any more)
    STAMP(0);
    // ---------------- judge the candidate that the sweep just linearised -------------------------
    if (t == 0 && !s.judged) {
        Ctl& c = s.c;
        const int cand = 1 - c.cur;
        const double cand_cost = merged ? ld_ag(P.sys[cand].cost) : *P.sys[cand].cost;
        c.cand_cost = cand_cost;
        if (c.first && *P.setup_stat != 0) { c.done = 1; c.term = 6; c.status = *P.setup_stat; }      // k_setup: an IMU covariance is not positive definite (resident window: or its prior is not finite)
        if (c.first || c.resweep) {
            if (c.first) { c.initial_cost = cand_cost; s.was_first = 1; }
            if (!isfinite(cand_cost)) { c.done = 1; c.term = 6; c.status = -3; }
            c.cur = cand; c.cost_cur = cand_cost; c.first = 0; c.resweep = 0; s.need = 1;
        } else judge_step(c, cand_cost, O, s.need);
        judge_caps(c, O);
    }
    __syncthreads();
    const int cur = s.c.cur;
    SysBuf sb = P.sys[cur];
    const double* x = P.x[cur];
    double* xc = P.x[1 - cur];
    // ---- the two landmark passes over [l0, l1) (helper: its slice; master without helpers: everything) -----------------------
    // pass 1 (needs u = Sc gradient_/d of the camera part in vc): dl, gradient_l, share of u^T H u, |g|^2, max |b|
    auto lm_pass1 = [&](int l0, int l1, const double* vc, double& q, double& g2, double& gm) {
        for (int l = l0 + t; l < l1; l += NT) {
            const double ip = ldx<FUSED>(sb.invp + l), Sl = ldx<FUSED>(sb.sl + l), h = ldx<FUSED>(sb.hll + l), b = ldx<FUSED>(sb.bl + l);
            const double d = sqrt(fmin(fmax(Sl * Sl * h, 1e-6), 1e32));
            const double g = ip != 0.0 ? Sl * b / d : 0.0;
            P.dl[l] = d; P.gradl[l] = g;
            if (ip != 0.0) {
                const double ul = Sl * g / d;
                const double ev = lm_dot<FUSED>(P, sb, l, vc);
                q += ip * ev * ev + 2.0 * ul * ev + h * ul * ul;
                g2 += g * g; gm = fmax(gm, fabs(b));
            }
        }
    };
    // pass 2 (needs Sc x_c of the pose part in vc): landmark back-substitution, step directions, the six sums
    auto lm_pass2 = [&](int l0, int l1, const double* vc, double* sm /*6*/) {
        for (int l = l0 + t; l < l1; l += NT) {
            const double ip = ldx<FUSED>(sb.invp + l);
            double a = 0.0, b = 0.0;
            if (ip != 0.0) {
                const double Sl = ldx<FUSED>(sb.sl + l), dl = P.dl[l], g = P.gradl[l];
                const double xl = (ldx<FUSED>(sb.bl + l) - lm_dot<FUSED>(P, sb, l, vc)) * ip / Sl;
                const double gnv = -xl * dl;
                sm[0] += gnv * gnv; sm[1] += gnv * g;
                a = Sl * g / dl; b = Sl * gnv / dl;
                sm[2] += a * a; sm[3] += a * b; sm[4] += b * b;
            }
            stx<FUSED>(P.la + l, a); stx<FUSED>(P.lb + l, b);
            if (!(P.lm_const && P.lm_const[l])) { const double lam = ldx<FUSED>(x + xo_lam(P) + l); sm[5] += lam * lam; }      // (an accepted candidate's inverse depths: the visual workgroups of this launch wrote them)
        }
    };
    // helper side of the second pass: poll "published" and "given up" together (one round trip), then every wave posts its own
    // sums -- the master's gathering wave adds the <= 8 nhelp slots with a fixed tree, no block reduction on the critical path
    auto wait_x = [&](double* dst) {           // dst[0 .. NV) = Sc x_p as soon as its words carry this launch's epoch; s.ok (1 on entry) = 0 if the master gave up
        for (int w = t; w < 2 * P.NV; w += NT) {
            const unsigned long long* p = (const unsigned long long*)P.stepc + w;
            unsigned long long tw0 = 0;
            for (int sp = 0;;) {
                const unsigned long long v = ld_ll(p);
                const int b = __hip_atomic_load(P.xstat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) == (unsigned)epoch) { ((unsigned*)dst)[w] = (unsigned)v; break; }
                if (b == epoch) { s.ok = 0; break; }
                __builtin_amdgcn_s_sleep(1);
                if ((++sp & 1023) == 0 && P.abortf && wait_expired(tw0, P.abortf)) { st_ag(P.abortf, 1); s.ok = 0; break; }
            }
        }
        __syncthreads();
    };
    auto post_wave2 = [&](double* sm) {
        bsum6(sm, s);
        const int slot = bid - 1;
        if (t == 0) { unsigned long long* hp = (unsigned long long*)P.hpart2 + 16 * slot; for (int e = 0; e < 6; ++e) { put_ll(hp + 2 * e, __double2loint(sm[e])); put_ll(hp + 2 * e + 1, __double2hiint(sm[e])); } }
    };
    // master side: one wave gathers the helper waves' sums (h[0..6)) and, when with1, the first pass's three numbers (h[6..9))
    auto gather2 = [&](double* h, bool with1) {
        const int ln = t & 63;
        for (int e = 0; e < 9; ++e) h[e] = 0.0;
        for (int s0 = 0; s0 < nhelp; s0 += 64) {
            const int slot = s0 + ln;
            if (slot < nhelp) {                    // the six sums of a helper: twelve words of this epoch, all requested together
                const unsigned long long* hp = (const unsigned long long*)P.hpart2 + 16 * slot;
                unsigned long long w[12];
                unsigned long long tw0 = 0;
                for (int sp = 0;;) {
                    bool all = true;
#pragma unroll
                    for (int e = 0; e < 12; ++e) w[e] = ld_ll(hp + e);
#pragma unroll
                    for (int e = 0; e < 12; ++e) all = all && (unsigned)(w[e] >> 32) == (unsigned)epoch;
                    if (all) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++sp & 1023) == 0 && P.abortf && wait_expired(tw0, P.abortf)) { st_ag(P.abortf, 1); break; }
                }
#pragma unroll
                for (int e = 0; e < 6; ++e) h[e] += __hiloint2double((int)(unsigned)w[2 * e + 1], (int)(unsigned)w[2 * e]);
            }
        }
        if (with1 && ln < nhelp) for (int e = 0; e < 3; ++e) h[6 + e] = __hip_atomic_load(P.hpart + 4 * ln + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int e = 0; e < 8; ++e) h[e] = wave_total_l63(h[e]);          // a fixed tree: deterministic
        h[8] = wave_max_l63(h[8]);
    };
    if (helper) {
        const int hk = bid - 1;
        if (!s.c.done && s.need) {
            // camera vectors u = Sc gradient_/d in LDS (nothing global is written here: that is the master's job)
            for (int i = t; i < P.NV; i += NT) {
                const double dg = merged ? ld_ag(sb.diag + i) : sb.diag[i], b = merged ? ld_ag(sb.bc + i) : sb.bc[i];
                const double Sc = s.was_first ? (O.jacobi_scaling ? 1.0 / (1.0 + sqrt(dg)) : 1.0) : ldx<FUSED>(P.Sc + i);
                const double d = sqrt(fmin(fmax(Sc * Sc * dg, 1e-6), 1e32));
                s.y[i] = Sc * (Sc * b / d) / d;
            }
            __syncthreads();
            const int per = (L + nhelp - 1) / nhelp, l0 = hk * per, l1 = min(L, l0 + per);
            double q = 0, g2 = 0, gm = 0;
            if (per <= NT) {
                // one landmark per thread: its rows and scalars stay in registers between the two passes, the second pass
                // (which the master waits for) reads only Sc x_p
                // (a QUAD per landmark while the helper has four threads for each -- lm_rows_quad: up to twelve observers in registers; the quad's lanes hold the same
                //  scalars, its lane 0 owns the landmark's sums and stores)
                const bool quad = 2 * per <= NT;                  // (lanes per landmark: four while the helper has them, else two, else one)
                const int G = 4 * per <= NT ? 4 : (quad ? 2 : 1);
                const int qd = t & (G - 1);
                const int l = l0 + (G == 4 ? (t >> 2) : G == 2 ? (t >> 1) : t);
                const bool have = l < l1, lead = have && qd == 0;
                LmRows r; r.fs = 0; r.fe = 0;
                double ip = 0, b = 0, d = 1, g = 0, lam2 = 0, ipS = 0, Sd = 0, a_ = 0, h_ = 0;      // ipS = invp / Sl, Sd = Sl / dl, a_ = la
                if (have) {
                    if (quad) lm_rows_quad<FUSED>(P, sb, l, qd, r, G); else lm_rows<FUSED>(P, sb, l, r);
                    ip = ldx<FUSED>(sb.invp + l); b = ldx<FUSED>(sb.bl + l);
                    const double Sl = ldx<FUSED>(sb.sl + l), h = ldx<FUSED>(sb.hll + l);
                    h_ = h;
                    if (!(P.lm_const && P.lm_const[l])) { const double lam = ldx<FUSED>(x + xo_lam(P) + l); lam2 = lam * lam; }
                    d = sqrt(fmin(fmax(Sl * Sl * h, 1e-6), 1e32));
                    g = ip != 0.0 ? Sl * b / d : 0.0;
                    if (lead) { P.dl[l] = d; P.gradl[l] = g; }
                    if (ip != 0.0) { Sd = Sl / d; ipS = ip / Sl; a_ = Sd * g; }               // the divides of the second pass, done while the master solves
                }
                {
                    double ev = (have && ip != 0.0) ? lm_dot_rows<FUSED>(P, sb, r, s.y) : 0.0;
                    if (G == 4) ev = quad_total(ev); else if (G == 2) ev = pair_total(ev);
                    if (lead && ip != 0.0) {
                        q += ip * ev * ev + 2.0 * a_ * ev + h_ * a_ * a_;
                        g2 += g * g; gm = fmax(gm, fabs(b));
                    }
                }
                bsum3<true>(g2, q, gm, s);
                if (t == 0) { double* hp = P.hpart + 4 * hk; put(hp, q); put(hp + 1, g2); put(hp + 2, gm); }
                post(P.hflag + hk);
                wait_x(s.gn);
                if (s.ok) {
                    double sm[6] = {0, 0, 0, 0, 0, 0};
                    double b_ = 0.0;
                    double ev2 = (have && ip != 0.0) ? lm_dot_rows<FUSED>(P, sb, r, s.gn) : 0.0;
                    if (G == 4) ev2 = quad_total(ev2); else if (G == 2) ev2 = pair_total(ev2);
                    if (lead) {
                        if (ip != 0.0) {
                            const double xl = (b - ev2) * ipS;
                            const double gnv = -xl * d;
                            sm[0] = gnv * gnv; sm[1] = gnv * g;
                            b_ = Sd * gnv;
                            sm[2] = a_ * a_; sm[3] = a_ * b_; sm[4] = b_ * b_;
                        }
                        sm[5] = lam2;
                    }
                    post_wave2(sm);
                    if (lead) { stx<FUSED>(P.la + l, a_); stx<FUSED>(P.lb + l, b_); }      // (for the next sweep: not part of what the master waits for -- k_solve's helpers post hflag2 behind these stores, its master collects that flag before it opens the next iteration)
                }
                return;
            }
            lm_pass1(l0, l1, s.y, q, g2, gm);
            bsum3<true>(g2, q, gm, s);
            if (t == 0) { double* hp = P.hpart + 4 * hk; put(hp, q); put(hp + 1, g2); put(hp + 2, gm); }
            post(P.hflag + hk);
            // second pass once the master has the pose part of the solution
            wait_x(s.y);
            if (s.ok) {
                double sm[6] = {0, 0, 0, 0, 0, 0};
                lm_pass2(l0, l1, s.y, sm);
                post_wave2(sm);
            }
        } else post(P.hflag + hk);
        return;
    }
    if (s.c.done) { if (t < 64) end_iter(); return; }
    for (int i = t; i < 16 * P.K + 8; i += NT) s.x0[i] = ldx<FUSED>(x + i);          // (read after several barriers)
    for (int k = t; k < 2 * P.K; k += NT) s.cst[k] = k < P.K ? (P.pose_const ? P.pose_const[k] : 0) : (P.sb_const ? P.sb_const[k - P.K] : 0);
    bool xpub = false;                                 // the master owes the waiting helpers an xflag on every path through the need branch
    auto publish_xp = [&](int okk) {                   // called by all threads; s.y[0 .. NV) = x_p when okk
        if (nhelp) {
            // Sc x_p goes out as 64-bit words {half of a value, launch epoch}: a helper that reads a word of this epoch has the value -- no flag
            // behind the data, no wait for the stores, and on the other side ONE round trip instead of a poll followed by the loads
            if (okk) { for (int w = t; w < 2 * P.NV; w += NT) { const double v = s.sc[w >> 1] * s.y[w >> 1]; put_ll((unsigned long long*)P.stepc + w, (w & 1) ? __double2hiint(v) : __double2loint(v)); } }
            else { __syncthreads(); post(P.xstat); }
        }
        xpub = true;
    };
    // prefetch this thread's share of S' (tiled order) so that the global latency hides behind the vector passes
    double pf[PF_N];
    if (CHAIN == 0 && s.need) {
        const int R = D + 1, T = (R + 15) >> 4, NTL = (tri_off(T)) << 8;
        const int half = __builtin_amdgcn_readfirstlane(t >> 8), w = t & 255;
        static_for<PF_N>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int I0 = tri_row_c(2 * u), J0 = tri_col_c(2 * u), I1 = tri_row_c(2 * u + 1), J1 = tri_col_c(2 * u + 1);
            const int e = t + u * VIL_STEP_THREADS;
            pf[u] = 0.0;
            if (e < NTL) {
                const int I = half ? I1 : I0, J = half ? J1 : J0;
                const int i = (I << 4) + (w >> 4), j = (J << 4) + (w & 15);
                if (i < D && j <= i) pf[u] = sb.S[(size_t)i * D + j];
            }
        });
    }
    STAMP(1);
    double gn2 = 0, g2 = 0, gg = 0;
    // The candidate x_cur (+) step for the FULL Gauss-Newton step (cg = 0, cn = 1: what the dogleg takes whenever the step fits the trust region -- every iteration after the
    // first few) is formed SPECULATIVELY by waves 0 and 1 while the last wave collects the helpers' sums: a camera block is a lane, which forms the
    // step entries of its own block itself (no step vector in LDS, no barrier); the two waves' totals are the block sum of the norms.  When the
    // dogleg then does take the full step, the ordinary path's step loop, barrier, pose_plus and block sum are skipped; (0 * gradient + 1 * gn) * rt has the value gn * rt.
    bool spec = false;
    double* const xcs_spec = Alds; double* const sums_spec = Alds + 336;      // (the factor's tiles: dead once the back substitutions are through)
    auto cand_blocks = [&](auto&& stepf, double* xcs, double& xn, double& sn) {
        const int K = P.K;
        for (int k = t; k < 2 * K + 2; k += NT) {
            if (k < K) {
                const double* in = s.x0 + xo_pose(P, k); double* o = xcs + xo_pose(P, k);
                if (s.cst[k]) { for (int q = 0; q < 7; ++q) o[q] = in[q]; }
                else { double st[6]; for (int q = 0; q < 6; ++q) st[q] = stepf(col_pose(P, k) + q); pose_plus(in, st, o); for (int q = 0; q < 7; ++q) { xn += in[q] * in[q]; sn += (in[q] - o[q]) * (in[q] - o[q]); } }
            } else if (k < 2 * K) {
                const int kk = k - K;
                const double* in = s.x0 + xo_sb(P, kk); double* o = xcs + xo_sb(P, kk);
                const bool cst = s.cst[K + kk] != 0;
                for (int q = 0; q < 9; ++q) { const double d = cst ? 0.0 : stepf(col_sb(P, kk) + q); o[q] = in[q] + d; if (!cst) { xn += in[q] * in[q]; sn += d * d; } }
            } else if (k == 2 * K) {
                const double* in = s.x0 + xo_ex(P); double* o = xcs + xo_ex(P);
                if (P.ex_const) { for (int q = 0; q < 7; ++q) o[q] = in[q]; }
                else { double st[6]; for (int q = 0; q < 6; ++q) st[q] = stepf(col_ex(P) + q); pose_plus(in, st, o); for (int q = 0; q < 7; ++q) { xn += in[q] * in[q]; sn += (in[q] - o[q]) * (in[q] - o[q]); } }
            } else {
                const double in = s.x0[xo_td(P)];
                const double d = P.td_free ? stepf(col_td(P)) : 0.0;
                xcs[xo_td(P)] = in + d;
                if (P.td_free) { xn += in * in; sn += d * d; }
            }
        }
    };
    if (s.need) {
        // ---- camera vectors: Jacobi scaling (first linearisation), dogleg diagonal, gradient_, u = Sc gradient_/d
        double gm = 0;
        if constexpr (CHAIN == 3) { if (merged) { if (t == 0) wait1(P.chflag + 1); __syncthreads(); } }      // the chain workgroup's scales of the chain columns are out
        for (int i = t; i < D; i += NT) {
            const double dg = merged ? ld_ag(sb.diag + i) : sb.diag[i], b = merged ? ld_ag(sb.bc + i) : sb.bc[i];
            double Sc;
            if (s.was_first) { Sc = O.jacobi_scaling ? 1.0 / (1.0 + sqrt(dg)) : 1.0; if (CHAIN == 3 && i >= P.NV) Sc = ld_ag(P.chSc + (i - P.NV)); stx<FUSED>(P.Sc + i, Sc); } else Sc = ldx<FUSED>(P.Sc + i);
            double d = sqrt(fmin(fmax(Sc * Sc * dg, 1e-6), 1e32));
            if (CHAIN == 3 && i >= P.NV) d = ld_ag(P.chDc + (i - P.NV));      // the very numbers the chain workgroup scaled M_bb with
            const double g = Sc * b / d;
            s.sc[i] = Sc; s.dcs[i] = d; s.gr[i] = g; s.y[i] = Sc * g / d; s.gd[i] = merged ? ld_ag(sb.gred + i) : sb.gred[i]; s.rt[i] = Sc / d;
            P.dc[i] = d; P.gradc[i] = g;
            if (cam) { g2 += g * g; gm = fmax(gm, fabs(b)); }
        }
        __syncthreads();
        STAMP(9);
        double q = 0;
        if (!nhelp) lm_pass1(0, L, s.y, q, g2, gm);        // with helper workgroups this pass runs on their CUs
        STAMP(10);
        // ---- camera share of u^T H u, fused with packing M = Sc S' Sc + mu dc^2 (+ rhs row) into LDS ------------
        const double mu = s.c.mu;
        bool ok = true;
        if constexpr (CHAIN != 0) {
            __syncthreads();
            auto pub = [&]() { publish_xp(1); };
            auto side = [&]() {};          // (the helpers' sums are collected after the step vectors below: their round trip outlasts the chain walks -- also with a quad of threads per landmark: fetched by the spare wave beside the walks, the walks' barrier waits for the answer instead, 46.2 -> 47.2 us, and the sums are in at the same 49.4)
            if constexpr (CHAIN == 3) ok = solve_prechain<FUSED>(P, sb, s, Alds, mu, cam, q, epoch, pub, side);
            else ok = solve_chain<CHAIN == 1>(P, sb, s, Alds, mu, cam, q, pub, side);      // packing, chain, Schur update, dense part, back substitution
        } else {
        // tiled storage: element e of the tile array -> (i, j); S entries were prefetched into registers at kernel start
        double* Ag = P.M;
        const int R = D + 1, T = (R + 15) >> 4, NTL = (tri_off(T)) << 8;
        const int half = __builtin_amdgcn_readfirstlane(t >> 8), w = t & 255;
#ifdef VIL_STAMPS
        long long tpk0 = 0; if (t == 0) { asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tpk0) :: "memory"); }
#endif
        static_for<PF_N>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int I0 = tri_row_c(2 * u), J0 = tri_col_c(2 * u), I1 = tri_row_c(2 * u + 1), J1 = tri_col_c(2 * u + 1);
            const int e = t + u * VIL_STEP_THREADS;
            if (e < NTL) {
                const int I = half ? I1 : I0, J = half ? J1 : J0;
                const int i = (I << 4) + (w >> 4), j = (J << 4) + (w & 15);
                double m = 0.0;
                if (i < D && j <= i) {
                    const double v = pf[u];
                    if (cam) q += (i == j ? 1.0 : 2.0) * s.y[i] * v * s.y[j];
                    m = s.sc[i] * v * s.sc[j];
                    if (i == j) m += mu * s.dcs[i] * s.dcs[i];
                } else if (i == D && j < D) m = s.sc[j] * s.gd[j];
                if constexpr (LDSM) Alds[tl_phys(e)] = m; else Ag[tl_phys(e)] = m;
            }
        });
#ifdef VIL_STAMPS
        if (t == 0) { long long tpk1; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tpk1) :: "memory"); P.dbg[27] = tpk1 - tpk0; }
#endif
        for (int e = t + PF_N * VIL_STEP_THREADS; e < NTL; e += VIL_STEP_THREADS) {     // large windows (K > 10): remainder, direct loads
            const int tile = e >> 8, w = e & 255;
            const int I = s.tI[tile], J = s.tJ[tile];
            const int i = (I << 4) + (w >> 4), j = (J << 4) + (w & 15);
            double m = 0.0;
            if (i < D && j <= i) {
                const double v = sb.S[(size_t)i * D + j];
                if (cam) q += (i == j ? 1.0 : 2.0) * s.y[i] * v * s.y[j];
                m = s.sc[i] * v * s.sc[j];
                if (i == j) m += mu * s.dcs[i] * s.dcs[i];
            } else if (i == D && j < D) m = s.sc[j] * s.gd[j];
            if constexpr (LDSM) Alds[tl_phys(e)] = m; else Ag[tl_phys(e)] = m;
        }
        }
        STAMP(11);
        // chain path with helpers: the solve is already behind us, so the helpers' first sums are collected together with
        // their second ones (one block reduction, one poll, one round trip instead of two of each)
        const bool defer = CHAIN != 0 && nhelp > 0 && ok;
        if (!defer) bsum3<true>(g2, q, gm, s);
        if (nhelp && !defer) {                // the helpers' landmark sums, added in workgroup order (deterministic)
            if (t < 64) {                     // lane k fetches helper k's three numbers, lane 0 adds them in order
                wait_helpers();
                double h0 = 0, h1 = 0, h2 = 0;
                if (t < nhelp) {
                    h0 = __hip_atomic_load(P.hpart + 4 * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    h1 = __hip_atomic_load(P.hpart + 4 * t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    h2 = __hip_atomic_load(P.hpart + 4 * t + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                for (int k = 0; k < nhelp; ++k) { q += __shfl(h0, k); g2 += __shfl(h1, k); gm = fmax(gm, __shfl(h2, k)); }
                if (t == 0) { s.red[0] = q; s.red[1] = g2; s.red[2] = gm; }
            }
            __syncthreads();
            q = s.red[0]; g2 = s.red[1]; gm = s.red[2];
            __syncthreads();
        }
        STAMP(2);
        if (!defer && gm <= O.gradient_tolerance) {
            if (!xpub) publish_xp(0);
            if (t == 0) { s.c.done = 1; s.c.term = 2; }
            if (t < 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); end_iter(); }
            return;
        }
        if constexpr (CHAIN == 0) { if constexpr (LDSM) ok = chol_lookahead<CH_SLOTS, false>(Alds, D, s); else ok = chol_blocked<false>(P.M, D, s, Alds); }      // Alds = staging of the active tile column
        STAMP(3);
        PROF(21);
#ifdef VIL_STAMPS
        if (t == 0) { P.dbg[20] = s.tacc[0]; P.dbg[21] = s.tacc[1]; P.dbg[22] = s.tacc[2]; P.dbg[24] = s.tacc[3]; P.dbg[25] = s.tacc[4]; P.dbg[26] = s.tacc[5]; }
#endif
        if (!ok) {
            // dogleg_strategy.cc: mu *= 10 and retry; the Schur pivots depend on mu, so re-sweep at x_cur
            if (!xpub) publish_xp(0);
            if (t == 0) {
                Ctl& c = s.c;
                c.mu *= 10.0;
                if (!(c.mu < O.max_mu)) { c.iter++; c.invalid_run++; c.reuse = 0; if (c.invalid_run >= 5) { c.done = 1; c.term = 6; c.status = -4; } }
                c.resweep = 1; c.cg = 0.0; c.cn = 0.0;
            }
            for (int i = t; i < P.NS; i += NT) stx<FUSED>(xc + i, i >= xo_lam(P) ? ldx<FUSED>(x + i) : s.x0[i]);
            if (FUSED && P.persist) for (int w = t; w < 2 * (16 * P.K + 8); w += NT) put_ll(P.xtag + w, (w & 1) ? __double2hiint(s.x0[w >> 1]) : __double2loint(s.x0[w >> 1]));      // (the sweep roles of a persistent solve poll these)
            __syncthreads();
            if (t < 64) end_iter();      // (a late helper may still be copying Ctl into its LDS)
            return;
        }
        if constexpr (CHAIN == 0) { if constexpr (LDSM) back_subst(Alds, D, s); else back_subst(P.M, D, s); publish_xp(1); __syncthreads(); }      // (the publishing threads read s.y across the thread map of the loop below)
        STAMP(4);
#ifdef VIL_STAMPS
        if (t == 0) P.dbg[23] = s.tacc[0];
#endif
        // ---- gauss-newton step in dogleg space (camera part); landmark back-substitution + sums -----------------------------
        for (int i = t; i < D; i += NT) {
            const double xi = s.y[i];
            const double gnv = -xi * s.dcs[i];
            s.gn[i] = gnv; P.gnc[i] = gnv;
            if (cam) { gn2 += gnv * gnv; gg += gnv * s.gr[i]; }
            s.y[i] = s.sc[i] * xi;     // Sc x_c for the landmark back-substitution
        }
        __syncthreads();
        // chain path with helpers: the last wave collects the helpers' sums of both passes (-> s.hs) HERE -- the helpers received x_p before the chain back
        // substitution and answer ~5 us later, the master's chain walks take half of that: the step vectors above no longer wait for the answer
        if (defer && LDSM && NT >= 192 && P.K + 1 <= 64) {
            // TWO waves, one code path each (the blocks of one kind in one wave: 2 K + 2 lanes of one wave walked four divergent paths one after the other, 2.3 us -- the
            // longest item between the chain's back substitution and the dogleg): wave 0 the K poses and the extrinsic, wave 1 the K speed-bias blocks and td
            auto stepf = [&](const int i) { return (0.0 * s.gr[i] + 1.0 * s.gn[i]) * s.rt[i]; };
            const int K = P.K;
            double xs_ = 0, ss_ = 0;
            if (t < 64) {
                if (t <= K) {
                    const bool isex = t == K;
                    const int xo = isex ? xo_ex(P) : xo_pose(P, t), col = isex ? col_ex(P) : col_pose(P, t);
                    const bool cst = isex ? P.ex_const != 0 : s.cst[min(t, K - 1)] != 0;
                    const double* in = s.x0 + xo; double* o = xcs_spec + xo;
                    double st[6], ov[7];
#pragma unroll
                    for (int q = 0; q < 6; ++q) st[q] = stepf(col + q);
                    pose_plus(in, st, ov);
#pragma unroll
                    for (int q = 0; q < 7; ++q) { const double iv = in[q]; o[q] = cst ? iv : ov[q]; if (!cst) { xs_ += iv * iv; ss_ += (iv - ov[q]) * (iv - ov[q]); } }
                }
            } else if (t < 128) {
                const int l = t - 64;
                if (l < K) {
                    const double* in = s.x0 + xo_sb(P, l); double* o = xcs_spec + xo_sb(P, l);
                    const bool cst = s.cst[K + l] != 0;
#pragma unroll
                    for (int q = 0; q < 9; ++q) { const double iv = in[q], d = cst ? 0.0 : stepf(col_sb(P, l) + q); o[q] = iv + d; if (!cst) { xs_ += iv * iv; ss_ += d * d; } }
                } else if (l == K) {
                    const double iv = s.x0[xo_td(P)], d = P.td_free ? stepf(col_td(P)) : 0.0;
                    xcs_spec[xo_td(P)] = iv + d;
                    if (P.td_free) { xs_ += iv * iv; ss_ += d * d; }
                }
            }
            if (t < 128) {
                xs_ = wave_total_l63(xs_); ss_ = wave_total_l63(ss_);
                if ((t & 63) == 63) { sums_spec[2 * (t >> 6)] = xs_; sums_spec[2 * (t >> 6) + 1] = ss_; }
            }
            spec = true;      // (what waves 0 and 1 wrote is read behind the barriers of the block sums below)
        }
        if (defer && t >= NT - 64) { double h[9]; gather2(h, true); if ((t & 63) == 63) for (int e = 0; e < 9; ++e) s.hs[e] = h[e]; }
        double sm[6] = {0, 0, 0, 0, 0, 0};
        if (!nhelp) lm_pass2(0, L, s.y, sm);
        sm[0] += gn2; sm[1] += gg;
        if (defer) bsum5(g2, q, gm, sm[0], sm[1], s);
        else { bsum3(sm[0], sm[1], sm[2], s); if (!nhelp) bsum3(sm[3], sm[4], sm[5], s); }      // (with helpers the master's share of the last three is zero)
        if (defer) {                              // the helpers' nine numbers are already in LDS (side(), above)
            hseen = true;
            for (int e = 0; e < 6; ++e) sm[e] += s.hs[e];
            q += s.hs[6]; g2 += s.hs[7]; gm = fmax(gm, s.hs[8]);
        }
        if (nhelp) {
            if (!defer) {
            if (t < 64) {
                double h[9];
                gather2(h, false);
                if (t == 63) for (int e = 0; e < 6; ++e) s.red[e] = sm[e] + h[e];
            }
            __syncthreads();
            for (int e = 0; e < 6; ++e) sm[e] = s.red[e];
            __syncthreads();
            }
            if (defer && gm <= O.gradient_tolerance) {          // (the check the other paths make before the factorisation)
                if (t == 0) { s.c.done = 1; s.c.term = 2; }
                if (t < 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); hseen = false; end_iter(); }
                return;
            }
        }
        gn2 = sm[0]; gg = sm[1];
        if (t == 0) {
            Ctl& c = s.c;
            c.alpha = g2 / q; c.mu_used = c.mu; c.gn2 = gn2; c.g2 = g2; c.gg = gg;
            c.saa = sm[2]; c.sab = sm[3]; c.sbb = sm[4]; c.xnl = sm[5];
            c.mu = fmax(O.min_mu, 2.0 * c.mu / 10.0);
        }
        __syncthreads();
    } else {
        for (int i = t; i < D; i += NT) { s.sc[i] = ldx<FUSED>(P.Sc + i); s.dcs[i] = P.dc[i]; s.gr[i] = P.gradc[i]; s.gn[i] = P.gnc[i]; s.rt[i] = s.sc[i] / s.dcs[i]; }
        __syncthreads();
    }
    STAMP(5);
    PROF(22);
    // (every wait of the master is behind it here: a workgroup that gives up later than this is a helper waiting for the master, and the next launch's waits see its word)
    if (t == 0 && P.abortf) abort_pre = ld_ag(P.abortf);
    // persistent solve: the helpers' la / lb (read by the visual roles of the NEXT iteration) are out -- their flags are collected by the spare wave under the dogleg's scalars
    if (FUSED && P.persist && t >= NT - 64) { if ((t & 63) < nhelp) spin_until_eq(P.hflag2 + (t & 63), epoch, P.abortf); __builtin_amdgcn_wave_barrier(); if (t == NT - 64) s.pad0_ = epoch; }
    // ---------------- traditional dogleg in dogleg space (scalars saved with the linearisation) ----------------
    gn2 = s.c.gn2; g2 = s.c.g2; gg = s.c.gg;
    const double radius = s.c.radius, alpha = s.c.alpha, mu_u = s.c.mu_used;
    const double gn_norm = sqrt(gn2), g_norm = sqrt(g2);
    double cg, cn, dnorm;
    if (gn_norm <= radius) { cg = 0; cn = 1; dnorm = gn_norm; }
    else if (g_norm * alpha >= radius) { cg = -(radius / g_norm); cn = 0; dnorm = radius; }
    else {
        const double b_dot_a = -alpha * gg;
        const double a2 = (alpha * g_norm) * (alpha * g_norm);
        const double bma2 = a2 - 2 * b_dot_a + gn2;
        const double cc = b_dot_a - a2;
        const double dd = sqrt(cc * cc + bma2 * (radius * radius - a2));
        const double beta = (cc <= 0) ? (dd - cc) / bma2 : (radius * radius - a2) / (dd + cc);
        cg = -alpha * (1 - beta); cn = beta; dnorm = radius;
    }
    // model decrease from the linear-algebra identities of the solved system (u = Sc gradient_/d, w = GN step):
    //   u^T H u = g2/alpha ; H w = -g - mu (d/S)^2 w  =>  u^T H w = -g2 - mu gg ,  w^T H w = -gg - mu gn2 ;  g^T u = g2 , g^T w = gg
    const double qd = cg * cg * (g2 / alpha) + 2.0 * cg * cn * (-g2 - mu_u * gg) + cn * cn * (-gg - mu_u * gn2);
    const double gd = cg * g2 + cn * gg;
    const double model_change = -(0.5 * qd + gd);
    // ---------------- candidate state x_cur (+) step: camera blocks here, inverse depths in the next sweep ----------------------
    double* xcs = s.gr;                              // the candidate's camera part is formed in LDS (gr | gn: 640 doubles, dead from here on -- P.gradc / P.gnc keep them) and leaves in one pass
    static_assert(offsetof(StepShared, gn) == offsetof(StepShared, gr) + 320 * sizeof(double), "the candidate spills from gr into gn");
    double xn = 0, sn = 0;
    if (spec && cg == 0.0 && cn == 1.0) { xcs = xcs_spec; xn = sums_spec[0] + sums_spec[2]; sn = sums_spec[1] + sums_spec[3]; }      // (uniform: every thread computed the same scalars)
    else {
        for (int i = t; i < D; i += NT) s.y[i] = (cg * s.gr[i] + cn * s.gn[i]) * s.rt[i];
        __syncthreads();
        cand_blocks([&](const int i) { return s.y[i]; }, xcs, xn, sn);
        double dummy2 = 0;
        bsum3(xn, sn, dummy2, s);
    }
    // landmark share of the norms from the sums of the pass
    xn += s.c.xnl; sn += cg * cg * s.c.saa + 2.0 * cg * cn * s.c.sab + cn * cn * s.c.sbb;
    STAMP(6);
    PROF(23);
    if (t == 0) {
        Ctl& c = s.c;
        c.iter++;
        if (c.iter <= 64) { c.radius_trace[c.iter - 1] = c.radius; c.cost_trace[c.iter - 1] = c.cost_cur; }
        c.dogleg_norm = dnorm;
        c.model_change = model_change;
        c.cg = cg; c.cn = cn;               // the sweep forms lambda + cg la + cn lb
        if (!(model_change > 0.0)) {
            // invalid step (trust_region_minimizer.cc HandleInvalidStep): mu *= 10, same linearisation, new pivots
            c.invalid_run++;
            if (c.invalid_run >= 5) { c.done = 1; c.term = 6; c.status = -4; }
            c.mu *= 10.0; c.reuse = 0; c.resweep = 1; c.cg = 0.0; c.cn = 0.0;
        } else {
            c.invalid_run = 0;
            if (sqrt(sn) <= O.parameter_tolerance * (sqrt(xn) + O.parameter_tolerance)) { c.done = 1; c.term = 3; }
        }
        // persistent solve: the time cap is looked at HERE (k_solve's tail looked at it a microsecond later), so that what the next iteration's sweep roles wait for can
        // leave with the candidate: the step just formed was never judged -- not an iteration of the summary
        if (FUSED && deadline_tick > 0 && !c.done && (long long)wall_clock64() > deadline_tick) { c.done = 1; c.term = 5; if (c.iter > 0 && !c.resweep) c.iter--; s.hdr_posted = 2; }
    }
    __syncthreads();
    // persistent solve: the hand-over line of the next iteration's sweep roles goes out with (in front of) the candidate's tagged words -- they are all a sweep role reads of this
    // iteration (Ctl itself, and the word the step roles wait for, still leave through k_solve's tail) -- once the helpers' la / lb are known to be out (s.pad0_: their
    // flags were collected under the dogleg's scalars; otherwise the tail collects them and posts the line as before).  1.5 us earlier than behind the tail's barriers.
    if constexpr (FUSED) {
        if (P.persist && bid == 0 && !s.c.done && s.pad0_ == epoch) {
            post_iter_header(P, s.c, epoch);
            if (t == 0) { s.hdr_posted = 1; prof_stamp(P, lidx, 25); }
        }
    }
    {   // the candidate's camera part leaves in one pass (a re-sweep: the current state again)
        const bool back = s.c.resweep && !s.c.done;
        for (int i = t; i < 16 * P.K + 8; i += NT) stx<FUSED>(xc + i, back ? s.x0[i] : xcs[i]);
        // persistent solve: the same values as 64-bit words {half of a value, epoch} -- the sweep roles of the next iteration poll THEM (data and flag in one round trip)
        if (FUSED && P.persist) for (int w = t; w < 2 * (16 * P.K + 8); w += NT) { const double v = back ? s.x0[w >> 1] : xcs[w >> 1]; put_ll(P.xtag + w, (w & 1) ? __double2hiint(v) : __double2loint(v)); }
    }
    STAMP(7);
    if (t < 64) end_iter();      // (no second poll when the sums were already collected)
    PROF(12);
}

#ifndef VIL_PERSIST_TU
template <bool LDSM, int CHAIN = 0>
__global__ __launch_bounds__(VIL_STEP_THREADS) void k_step(DevP P, SolveOpts O) {
    __shared__ vd::StepShared s;
    extern __shared__ double Alds[];
    step_body<LDSM, CHAIN, false>(P, O, s, Alds, (int)blockIdx.x);
}
#endif
