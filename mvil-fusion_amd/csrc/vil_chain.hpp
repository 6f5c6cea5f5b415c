// Structure-exploiting solve of the reduced system (the dense-solve half of ceres' DENSE_SCHUR, estimator.cpp:1400-1414).
//
// In the reduced ordering [pose 6K | ex 6 | td 1 | speed-bias 9K] the speed-bias part of M = Sc S' Sc + mu dc^2 is a CHAIN:
// speed-bias k is coupled to speed-bias k +- 1 only (IMUFactor couples (i, i+1), estimator.cpp:1179-1186; the prior holds
// one speed-bias block), so M_bb is block-tridiagonal with 9 x 9 blocks.  Instead of a dense Cholesky over all D = 15K + 7
// columns (157 pivots at K = 10, the single-CU critical path of round 1):
//   1. the chain is eliminated FIRST, from BOTH ends at once ("twisted" block factorisation: blocks 0 .. m-1 forwards and
//      K-1 .. m+1 backwards by two independent groups of waves, meeting in block m = K/2): ceil(K/2) + 1 sequential 9 x 9 steps.
//      A lane owns a row of the panel [next speed-bias block (9 rows) | pose part incl. the right-hand-side row (6K + 8)];
//      the 9 x 9 diagonal block is factored redundantly in every lane's registers, the row is solved against it, and the
//      fill that the elimination creates in the following chain block is carried in registers (9 doubles per row);
//   2. the Schur complement of the pose part, S_pp -= W W^T (W: (6K+8) x 9K), is ONE dense contraction on the fp64 matrix
//      cores with the result left in the accumulators the blocked Cholesky starts from;
//   3. the dense blocked Cholesky / back substitution of vil_step.hpp run on 6K + 8 rows (67 pivots instead of 157);
//   4. the chain is back-substituted outwards from the middle block, both directions concurrently.
// Same solution as the dense factorisation up to rounding (a different elimination order of the same SPD matrix).
// Windows whose speed-bias coupling is not a chain (checked on the host at upload) keep the dense path.
#pragma once
#include "vil_dev.hpp"
#include "vil_math.hpp"

namespace vd {

__host__ __device__ inline int chain_rs(int K) { int rs = ((6 * K + 8 + 15) >> 4) << 4; if ((rs & 31) != 16) rs += 16; return rs; }   // row stride of W^T: >= 16 T, = 16 mod 32 (LDS banks)
__host__ __device__ inline int even_up(int v) { return (v + 1) & ~1; }
__host__ __device__ inline int chain_wcols(int K) { return (9 * K + 31) & ~31; }     // columns of W^T incl. padding: the Schur contraction reads 32 at a time
// Chain scratch in LDS behind the tile array (and W^T), in doubles:
//   Dk 2 x 82 | L_kk (45 + 9 reciprocal pivots) per block | sub-diagonal block per block (82 each) | carry of the backward
//   direction R x 9 | t (9K) | 8 ints of flags
struct ChainLds {
    double* Dk; double* Ldg; double* Lsb; double* cB; double* tB;
    volatile int* flag;       // [0], [1]: blocks published by the recursion wave of direction d; [2]: middle factor published;
                              // [3], [4]: carry of the backward direction published by its two row waves; [5]: a pivot was not positive
};
__host__ __device__ inline size_t chain_scratch_doubles(int K) { const int R = 6 * K + 8; return 164 + (size_t)54 * K + (size_t)82 * K + (size_t)9 * R + even_up(9 * K) + 8; }
__device__ __forceinline__ ChainLds chain_lds(double* cs, int K) {
    ChainLds L; const int R = 6 * K + 8;
    L.Dk = cs; L.Ldg = cs + 164; L.Lsb = L.Ldg + 54 * K; L.cB = L.Lsb + 82 * K; L.tB = L.cB + 9 * R; L.flag = (volatile int*)(L.tB + even_up(9 * K));
    return L;
}

struct L9 { double l[45]; double r[9]; };     // lower factor, l[i(i+1)/2 + j], and reciprocal pivots

// Cholesky of the 9 x 9 block whose lower triangle sits at Dk[i * 9 + j] (LDS, same address in every lane: broadcast reads).
// Fully unrolled, register-resident; returns false on a non-positive pivot.
__device__ __forceinline__ bool chol9(const double* Dk, L9& o) {
    double a[45];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) a[(i * (i + 1) >> 1) + j] = Dk[i * 9 + j];
    double rsum = 0.0;        // (a pivot that is not positive and finite makes its reciprocal root NaN or inf, and every later one with it)
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        const double dpp = a[(p * (p + 1) >> 1) + p];
        const double rs = rsqrt_1(dpp);
        rsum += rs;
        o.l[(p * (p + 1) >> 1) + p] = rs; o.r[p] = rs;      // (the diagonal of L is never read: row solves, inverses and back substitutions use r)
#pragma unroll
        for (int i = p + 1; i < 9; ++i) o.l[(i * (i + 1) >> 1) + p] = a[(i * (i + 1) >> 1) + p] * rs;
#pragma unroll
        for (int j = p + 1; j < 9; ++j)
#pragma unroll
            for (int i = j; i < 9; ++i) a[(i * (i + 1) >> 1) + j] -= o.l[(i * (i + 1) >> 1) + p] * o.l[(j * (j + 1) >> 1) + p];
    }
    return rsum < 1.7976931348623157e308;
}

#define CHAIN_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// (relaxed workgroup-scope atomics, not volatile accesses: the address-space inference skips volatile loads and stores, and these flags then travel as FLAT
//  instructions -- each poll behind an s_waitcnt vmcnt(0) lgkmcnt(0), i.e. behind the completion of the row waves' own global stores of W^T, a microsecond per block)
__device__ __forceinline__ int chain_flag_get(volatile int* f) { return __hip_atomic_load(const_cast<int*>(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void chain_flag_set(volatile int* f, int v) { __hip_atomic_store(const_cast<int*>(f), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void chain_wait(volatile int* f, int v) { while (__hip_atomic_load(const_cast<int*>(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < v) __builtin_amdgcn_s_sleep(1); CHAIN_FENCE(); }
__device__ __forceinline__ void chain_post(volatile int* f, int v) { CHAIN_FENCE(); if ((vil_tid() & 63) == 0) __hip_atomic_store(const_cast<int*>(f), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // LDS operations of a wave execute in order

// solve one panel row against the factored 9 x 9 block: w = a L^-T
__device__ __forceinline__ void row_solve9(const double* l, const double* r, const double* a, double* w) {
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        double acc = a[p];
#pragma unroll
        for (int c = 0; c < p; ++c) acc -= w[c] * l[(p * (p + 1) >> 1) + c];
        w[p] = acc * r[p];
    }
}

// Two-sided elimination of the speed-bias chain, barrier-free inside the workgroup.  Waves 0 / 1 run the 9 x 9 recursion of the
// forward / backward direction (D_k -> L_kk -> L_{k+-1,k} -> D_{k+-1}) and publish every factored block through LDS + a flag;
// waves 2,3 / 4,5 own the pose-part rows (incl. the right-hand side) of the forward / backward direction and follow the flags:
// row solve against L_kk, W^T column block to Wt, fill carried into the next block.  Wave 0 finally factors the middle block,
// waves 2,3 finish its rows.  Other waves return at once (the caller gives them the tile packing).  The caller zeroes the flags
// before and puts a workgroup barrier after.
// SRC: raw S' entries -- diag(k, i, j): entry (i, j <= i) of diagonal block k; sub(k, kn, q, c): row q of block kn = k +- 1, column c
// of block k; prow(r, k, c): pose row r, column c of block k --; sc(j) scale of reduced column j; madd(j) = mu dc_j^2; rowscale(r) scale applied to pose row r
// (1 when the row scaling is deferred); rhsraw(j) reduced gradient of column j; u(j) (WITHQ) the vector of the quadratic form
// on the chain columns; row_done(d, r, z, q): z = (S'_pb u_b)[r] over the blocks of direction d (the step kernel adds 2 u_r z);
// wput(p, v): store of an entry of W^T (plain, or at agent scope when another workgroup of the launch reads it).
// WITHQ: qacc receives this lane's share of u^T S' u over every entry of S' with a row or a column in the chain part (each
// raw entry passes through exactly one lane here: pose row x chain block in the row waves, diagonal and sub-diagonal blocks in
// the recursion waves).
template <bool WITHQ, class SRC>
__device__ __forceinline__ void chain_eliminate(const SRC& src, const int K, const int NP, const int RS, double* Wt, const ChainLds& L, double& qacc, long long* dbg = nullptr) {
    const int t = vil_tid(), wave = t >> 6, lane = t & 63;
#ifdef VIL_STAMPS
    #define CSTMP(k) do { if (lane == 0 && dbg) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); dbg[k] = tt_; } } while (0)
#else
    #define CSTMP(k) do {} while (0)
#endif
    if (wave >= 6) return;
    const int R = NP + 1, m = K >> 1, nf = m, nb = K - 1 - m;
    if (wave < 2) {
        // ---------------- recursion wave of direction d ----------------------------------------------------------------------
        const int d = wave, nd = d == 0 ? nf : nb;
        int di = 0, dj = 0;
        { const int e = lane < 45 ? lane : 0; while ((di + 1) * (di + 2) / 2 <= e) ++di; dj = e - di * (di + 1) / 2; }
        bool ok = true;
        auto diag_entry = [&](int k) { return src.diag(k, di, dj); };
        auto diag_scaled = [&](int k, double v) { const int gi = NP + 9 * k + di, gj = NP + 9 * k + dj; double mv = src.sc(gi) * v * src.sc(gj); if (di == dj) mv += src.madd(gi); return mv; };
        auto sub_rows = [&](int k, int kn, double* a) {           // row `lane` of block kn against the columns of block k
            const int kc = min(max(kn, 0), K - 1), q = min(lane, 8);
#pragma unroll
            for (int c = 0; c < 9; ++c) a[c] = src.sub(k, kc, q, c);
        };
        double dv = 0.0, an[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) an[c] = 0.0;
        const int k0 = d == 0 ? 0 : K - 1;
        if (nd > 0) { dv = diag_entry(k0); sub_rows(k0, d == 0 ? 1 : K - 2, an); }
        else if (d == 0) dv = diag_entry(m);
        for (int st = 0; st < nd; ++st) {
            const int k = d == 0 ? st : K - 1 - st, kn = d == 0 ? k + 1 : k - 1, kp = d == 0 ? k - 1 : k + 1;
            double a[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) a[c] = an[c];
            const double dcur = dv;
            // next step's raw entries (or the middle block's diagonal) while this one computes
            if (st + 1 < nd) { dv = diag_entry(kn); sub_rows(kn, d == 0 ? kn + 1 : kn - 1, an); }
            else if (d == 0) dv = diag_entry(m);
            if (lane < 45) {
                double v = diag_scaled(k, dcur);
                if (WITHQ) qacc += (di == dj ? 1.0 : 2.0) * src.u(NP + 9 * k + di) * dcur * src.u(NP + 9 * k + dj);
                if (st > 0) { const double* Lp = L.Lsb + 82 * kp;
#pragma unroll
                    for (int c = 0; c < 9; ++c) v -= Lp[di * 9 + c] * Lp[dj * 9 + c]; }
                L.Dk[82 * d + di * 9 + dj] = v;
            }
            CHAIN_FENCE();
            L9 Lf;
            ok = chol9(L.Dk + 82 * d, Lf) && ok;
            if (lane < 9) {
                const double rsc = src.sc(NP + 9 * kn + lane);
                double w[9];
                if (WITHQ) {
                    double qs = 0.0;
#pragma unroll
                    for (int c = 0; c < 9; ++c) qs += a[c] * src.u(NP + 9 * k + c);
                    qacc += 2.0 * src.u(NP + 9 * kn + lane) * qs;
                }
#pragma unroll
                for (int c = 0; c < 9; ++c) a[c] = rsc * a[c] * src.sc(NP + 9 * k + c);
                row_solve9(Lf.l, Lf.r, a, w);
#pragma unroll
                for (int c = 0; c < 9; ++c) L.Lsb[82 * k + lane * 9 + c] = w[c];
            }
            if (lane == 0) {
#pragma unroll
                for (int e = 0; e < 45; ++e) L.Ldg[54 * k + e] = Lf.l[e];
#pragma unroll
                for (int e = 0; e < 9; ++e) L.Ldg[54 * k + 45 + e] = Lf.r[e];
            }
            chain_post(L.flag + d, st + 1);
            if (d == 0 && st < 6) CSTMP(48 + st);
        }
        if (d == 0) {                                  // middle block: both directions meet
            if (nb > 0) chain_wait(L.flag + 1, nb);
            if (lane < 45) {
                double v = diag_scaled(m, dv);
                if (WITHQ) qacc += (di == dj ? 1.0 : 2.0) * src.u(NP + 9 * m + di) * dv * src.u(NP + 9 * m + dj);
                if (nf > 0) { const double* Lp = L.Lsb + 82 * (m - 1);
#pragma unroll
                    for (int c = 0; c < 9; ++c) v -= Lp[di * 9 + c] * Lp[dj * 9 + c]; }
                if (nb > 0) { const double* Lp = L.Lsb + 82 * (m + 1);
#pragma unroll
                    for (int c = 0; c < 9; ++c) v -= Lp[di * 9 + c] * Lp[dj * 9 + c]; }
                L.Dk[di * 9 + dj] = v;
            }
            CHAIN_FENCE();
            L9 Lf;
            ok = chol9(L.Dk, Lf) && ok;
            if (lane == 0) {
#pragma unroll
                for (int e = 0; e < 45; ++e) L.Ldg[54 * m + e] = Lf.l[e];
#pragma unroll
                for (int e = 0; e < 9; ++e) L.Ldg[54 * m + 45 + e] = Lf.r[e];
            }
            chain_post(L.flag + 2, 1);
            CSTMP(54);
        }
        if (!ok) chain_flag_set(L.flag + 5, 1);
        return;
    }
    // ---------------- row waves: pose-part row r (r == NP: right-hand side) of direction d -------------------------------------
    const int d = (wave - 2) >> 1, half = (wave - 2) & 1, r = half * 64 + lane, nd = d == 0 ? nf : nb;
    const bool valid = r < R;
    const int rc = min(r, NP - 1);
    const double rsc = r < NP ? src.rowscale(rc) : 1.0;
    double zr = 0.0;                                   // (S'_pb u_b)[r], accumulated over the blocks of this direction
    auto fetch = [&](int k, double* a) {
#pragma unroll
        for (int c = 0; c < 9; ++c) a[c] = src.prow(rc, k, c);
        if (r >= NP) {
#pragma unroll
            for (int c = 0; c < 9; ++c) a[c] = src.rhsraw(NP + 9 * k + c);
        }
    };
    auto load_factor = [&](int k, double* l, double* rv) {
        const double* p = L.Ldg + 54 * k;
#pragma unroll
        for (int e = 0; e < 45; ++e) l[e] = p[e];
#pragma unroll
        for (int e = 0; e < 9; ++e) rv[e] = p[45 + e];
    };
    double carry[9], an[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) { carry[c] = 0.0; an[c] = 0.0; }
    if (nd > 0) fetch(d == 0 ? 0 : K - 1, an);
    else if (d == 0) fetch(m, an);
    for (int st = 0; st < nd; ++st) {
        const int k = d == 0 ? st : K - 1 - st, kn = d == 0 ? k + 1 : k - 1;
        double a[9], w[9], l[45], rv[9];
        if (WITHQ) {
            double qs = 0.0;
#pragma unroll
            for (int c = 0; c < 9; ++c) qs += an[c] * src.u(NP + 9 * k + c);
            zr += qs;
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) a[c] = rsc * an[c] * src.sc(NP + 9 * k + c) - carry[c];
        if (st + 1 < nd) fetch(kn, an);
        else if (d == 0) fetch(m, an);
        chain_wait(L.flag + d, st + 1);
        load_factor(k, l, rv);
        row_solve9(l, rv, a, w);
        if (valid) {
#pragma unroll
            for (int c = 0; c < 9; ++c) src.wput(Wt + (size_t)(9 * k + c) * RS + r, w[c]);
        }
        const double* L1 = L.Lsb + 82 * k;
#pragma unroll
        for (int cn = 0; cn < 9; ++cn) {
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 9; ++c) acc += w[c] * L1[cn * 9 + c];
            carry[cn] = acc;
        }
    }
    if (d == 1) {
        if (nb > 0) {
            if (valid) {
#pragma unroll
                for (int c = 0; c < 9; ++c) L.cB[r * 9 + c] = carry[c];
            }
            chain_post(L.flag + 3 + half, 1);
        }
        if (WITHQ && r < NP) src.row_done(1, r, zr, qacc);       // (not the right-hand-side row)
        if (half == 0) CSTMP(56);
        return;
    }
    chain_wait(L.flag + 2, 1);
    if (nb > 0) { chain_wait(L.flag + 3, 1); chain_wait(L.flag + 4, 1); }
    {
        double a[9], w[9], l[45], rv[9];
        if (WITHQ) {
            double qs = 0.0;
#pragma unroll
            for (int c = 0; c < 9; ++c) qs += an[c] * src.u(NP + 9 * m + c);
            zr += qs;
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) a[c] = rsc * an[c] * src.sc(NP + 9 * m + c) - carry[c] - ((nb > 0 && valid) ? L.cB[min(r, R - 1) * 9 + c] : 0.0);
        load_factor(m, l, rv);
        row_solve9(l, rv, a, w);
        if (valid) {
#pragma unroll
            for (int c = 0; c < 9; ++c) src.wput(Wt + (size_t)(9 * m + c) * RS + r, w[c]);
        }
    }
    if (WITHQ && r < NP) src.row_done(0, r, zr, qacc);
    if (half == 0) CSTMP(55);
}

// One block of the chain back substitution, executed by ONE wave:  x_k = L_kk^-T (t_k - Ls_k^T x_next).
// Ldg: 45 + 9 doubles of block k; Lsb: its sub-diagonal block (rows of the neighbouring block that was eliminated after it)
// or nullptr for the middle block; tk / xn / xo: LDS.  Lanes 0..8 form the right-hand side, lane 0 solves.
__device__ __forceinline__ void chain_block_back(const double* Ldg, const double* Lsb, double* tk, const double* xn, double* xo) {
    const int lane = vil_tid() & 63;
    if (lane < 9 && Lsb) {
        double v = tk[lane];
#pragma unroll
        for (int i = 0; i < 9; ++i) v -= Lsb[i * 9 + lane] * xn[i];
        tk[lane] = v;
    }
    CHAIN_FENCE();
    if (lane == 0) {
        double v[9], x[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) v[i] = tk[i];
#pragma unroll
        for (int p = 8; p >= 0; --p) {
            double acc = v[p];
#pragma unroll
            for (int i = 8; i > p; --i) acc -= Ldg[(i * (i + 1) >> 1) + p] * x[i];
            x[p] = acc * Ldg[45 + p];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) xo[i] = x[i];
    }
    CHAIN_FENCE();
}

}  // namespace vd
