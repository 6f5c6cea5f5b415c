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
//   direction R x 9 | t (9K) | 8 ints of flags.  The diagonal block a recursion wave factors sits where its factor will (Ldg: a packed lower triangle, read before the factor is
//   stored); the inverses of the factors (fw: entry (p, c) of block k at LI[LIs k + 9 p + c]) live in memory of the CALLER'S that is dead by then (prechain_wg: the slab's
//   sub-diagonal blocks) -- at K = 20 the workgroup has no LDS to spare
struct ChainLds {
    double* Dk; double* Ldg; double* Lsb; double* cB; double* tB; double* LI; int LIs;
    volatile int* flag;       // [0], [1]: blocks published by the recursion wave of direction d; [2]: middle factor published;
                              // [3], [4]: carry of the backward direction published by its two row waves; [5]: a pivot was not positive;
                              // [6], [7]: spare
};
__host__ __device__ inline size_t chain_scratch_doubles(int K) { const int R = 6 * K + 8; return 164 + (size_t)54 * K + (size_t)82 * K + (size_t)9 * R + even_up(9 * K) + 8; }
__device__ __forceinline__ ChainLds chain_lds(double* cs, int K) {
    ChainLds L; const int R = 6 * K + 8;
    L.Dk = cs; L.Ldg = cs + 164; L.Lsb = L.Ldg + 54 * K; L.cB = L.Lsb + 82 * K; L.tB = L.cB + 9 * R; L.LI = nullptr; L.LIs = 0; L.flag = (volatile int*)(L.tB + even_up(9 * K));
    return L;
}

struct L9 { double l[45]; double r[9]; };     // lower factor, l[i(i+1)/2 + j], and reciprocal pivots

// Cholesky of the 9 x 9 block whose lower triangle sits at Dk[i * 9 + j] (LDS, same address in every lane: broadcast reads).
// Fully unrolled, register-resident; returns false on a non-positive pivot.
template <bool PACKED = false>      // PACKED: the lower triangle at Dk[i (i + 1) / 2 + j]
__device__ __forceinline__ bool chol9(const double* Dk, L9& o) {
    double a[45];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) a[(i * (i + 1) >> 1) + j] = Dk[PACKED ? (i * (i + 1) >> 1) + j : i * 9 + j];
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

// column cc of the inverse of a factored 9 x 9 block (x = L^-1 e_cc by forward substitution; the arithmetic of chain_inverse_block, vil_prechain.hpp)
__device__ __forceinline__ void inv9_col(const L9& f, const int cc, double* x) {
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        double acc = p == cc ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < p; ++q) acc -= f.l[(p * (p + 1) >> 1) + q] * x[q];
        x[p] = p < cc ? 0.0 : acc * f.r[p];
    }
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
// waves 2,3 finish its rows.  fw (the caller's waves 5 .. 7 are free and L.LI is set): the recursion waves publish the INVERSE of every factored block and the rows are
// carried by chain_rows_mfma on the matrix cores (waves 2, 3, 5 forwards, 6, 7 backwards); otherwise a row per lane on waves 2 .. 5 against the published factors.  Other waves return at once (the caller gives them the tile packing).  The caller zeroes the flags
// before and puts a workgroup barrier after.
// SRC: raw S' entries -- diag(k, i, j): entry (i, j <= i) of diagonal block k; sub(k, kn, q, c): row q of block kn = k +- 1, column c
// of block k; prow(r, k, c): pose row r, column c of block k --; sc(j) scale of reduced column j; madd(j) = mu dc_j^2; rowscale(r) scale applied to pose row r
// (1 when the row scaling is deferred); rhsraw(j) reduced gradient of column j; u(j) (WITHQ) the vector of the quadratic form
// on the chain columns; row_done(d, r, z, q): z = (S'_pb u_b)[r] over the blocks of direction d (the step kernel adds 2 u_r z);
// wput(p, v): store of an entry of W^T (plain, or at agent scope when another workgroup of the launch reads it).
// WITHQ: qacc receives this lane's share of u^T S' u over every entry of S' with a row or a column in the chain part (each
// raw entry passes through exactly one lane here: pose row x chain block in the row waves, diagonal and sub-diagonal blocks in
// the recursion waves).
// The row waves of chain_eliminate on the matrix cores (see there).  MT: tiles of 16 rows a wave carries -- tiles half, half + 2, .. of its direction; a tile past the
// last one (K = 10: the third of wave `half` = 1) is skipped by a scalar branch.  Otherwise straight-line code: every load is unconditional from a clamped address and
// selected afterwards (a per-lane predicated load is an exec-mask branch with its own wait: the first version of this function spent 57 waits per block on them).
template <bool WITHQ, int MT, class SRC>
__device__ __forceinline__ void chain_rows_mfma(const SRC& src, const int K, const int NP, const int RS, double* Wt, const ChainLds& L, double& qacc, long long* dbg, const int d, const int half,
                                                const int t0, const int ntl /* this wave's tiles: t0, t0 + 2, .. (ntl of them, those below the direction's tile count) */) {
    typedef double d4v __attribute__((ext_vector_type(4)));
    const int lane = vil_tid() & 63;
    const int R = NP + 1, m = K >> 1, nf = m, nb = K - 1 - m, nd = d == 0 ? nf : nb, ntile = (R + 15) >> 4;
    const int rl = lane & 15, kq = lane >> 4;
    volatile int* const fflag = L.flag + d;
    int rr[MT], rcl[MT]; double rsc[MT]; bool act[MT], isrhs[MT];
#pragma unroll
    for (int u = 0; u < MT; ++u) {
        const int T = t0 + 2 * u;
        act[u] = u < ntl && T < ntile;                   // (wave-uniform)
        rr[u] = 16 * T + rl; rcl[u] = min(rr[u], NP - 1); isrhs[u] = rr[u] >= NP;
        rsc[u] = src.rowscale(rcl[u]); if (isrhs[u]) rsc[u] = 1.0;
    }
    const int c0 = kq, c1 = 4 + kq, c2 = 8;              // this lane's columns of a block (the third counts in lanes with kq = 0 only)
    const bool h2 = kq == 0;
    auto fetch = [&](const int k, double (*x)[3]) {
        const double g0 = src.rhsraw(NP + 9 * k + c0), g1 = src.rhsraw(NP + 9 * k + c1), g2 = src.rhsraw(NP + 9 * k + c2);
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            if (!act[u]) continue;                       // (wave-uniform: a scalar branch; the tile's entries stay zero)
            const double v0 = src.prow(rcl[u], k, c0), v1 = src.prow(rcl[u], k, c1), v2 = src.prow(rcl[u], k, c2);
            x[u][0] = isrhs[u] ? g0 : v0; x[u][1] = isrhs[u] ? g1 : v1; x[u][2] = h2 ? (isrhs[u] ? g2 : v2) : 0.0;
        }
    };
    // the 9 x 9 operand of a product: entry (row rl, column 4 s + kq) of M (row stride 9), zero outside
    auto operand = [&](const double* M, double* o) {
        const int mr = min(rl, 8);
        const double v0 = M[9 * mr + c0], v1 = M[9 * mr + c1], v2 = M[9 * mr + c2];
        const bool in = rl < 9;
        o[0] = in ? v0 : 0.0; o[1] = in ? v1 : 0.0; o[2] = (in && h2) ? v2 : 0.0;
    };
    double an[MT][3], zr[MT], cy[MT][3];                 // raw rows of the block at hand | a row's share of S'_pb u_b | carry of the block before (this lane's elements)
#pragma unroll
    for (int u = 0; u < MT; ++u) { zr[u] = 0.0; cy[u][0] = cy[u][1] = cy[u][2] = 0.0; an[u][0] = an[u][1] = an[u][2] = 0.0; }
    if (nd > 0) fetch(d == 0 ? 0 : K - 1, an);
    else if (d == 0) fetch(m, an);
    auto scaled = [&](const int k, double (*ax)[3]) {                 // A' of block k in this lane's elements; the rows' share of S'_pb u_b on the way
        const double s0 = src.sc(NP + 9 * k + c0), s1 = src.sc(NP + 9 * k + c1), s2 = src.sc(NP + 9 * k + c2);
        const double u0 = WITHQ ? src.u(NP + 9 * k + c0) : 0.0, u1 = WITHQ ? src.u(NP + 9 * k + c1) : 0.0, u2 = WITHQ ? src.u(NP + 9 * k + c2) : 0.0;
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            if (WITHQ) zr[u] += an[u][0] * u0 + an[u][1] * u1 + an[u][2] * u2;
            ax[u][0] = rsc[u] * an[u][0] * s0 - cy[u][0]; ax[u][1] = rsc[u] * an[u][1] * s1 - cy[u][1]; ax[u][2] = rsc[u] * an[u][2] * s2 - cy[u][2];
        }
    };
    // W_k^T = L_kk^-1 A'^T of one tile, its columns out (rows past R - 1 of the last tile land in the padding of W^T's columns: RS = 16 ntile)
    auto solve_store = [&](const int k, const int u, const double* li, const double* ax) {
        d4v acc = d4v{0.0, 0.0, 0.0, 0.0};
        if (act[u]) {
#pragma unroll
            for (int q = 0; q < 3; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(li[q], ax[q], acc, 0, 0, 0);
            double* const col = Wt + (size_t)(9 * k) * RS + rr[u];
            src.wput(col + (size_t)c0 * RS, acc[0]);
            src.wput(col + (size_t)c1 * RS, acc[1]);
            if (h2) src.wput(col + (size_t)8 * RS, acc[2]);
        }
        return acc;
    };
    for (int st = 0; st < nd; ++st) {
        const int k = d == 0 ? st : K - 1 - st, kn = d == 0 ? k + 1 : k - 1;
        double ax[MT][3];
        scaled(k, ax);
        const int k2 = st + 1 < nd ? kn : (d == 0 ? m : k);      // (the backward direction's last step asks for its own block again: no branch)
        fetch(k2, an);
        chain_wait(fflag, st + 1);
        double li[3], l1[3];
        operand(L.LI + L.LIs * k, li);
        operand(L.Lsb + 82 * k, l1);
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            if (!act[u]) continue;
            const d4v w = solve_store(k, u, li, ax[u]);
            d4v acc = d4v{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 3; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(l1[q], w[q], acc, 0, 0, 0);
            cy[u][0] = acc[0]; cy[u][1] = acc[1]; cy[u][2] = acc[2];
        }
    }
    // a row's S'_pb u_b: the four lanes of the row hold a share each
    auto row_sums = [&]() {
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            double z = zr[u];
            z += __shfl_xor(z, 16, 64); z += __shfl_xor(z, 32, 64);
            if (WITHQ && act[u] && kq == 0 && rr[u] < NP) src.row_done(d, rr[u], z, qacc);
        }
    };
#ifdef VIL_STAMPS
    #define RSTMP(k) do { if (lane == 0 && dbg) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); dbg[k] = tt_; } } while (0)
#else
    #define RSTMP(k) do {} while (0)
#endif
    if (d == 1) {
        if (nb > 0) {
#pragma unroll
            for (int u = 0; u < MT; ++u) if (act[u] && rr[u] < R) {
                L.cB[rr[u] * 9 + c0] = cy[u][0]; L.cB[rr[u] * 9 + c1] = cy[u][1];
                if (h2) L.cB[rr[u] * 9 + 8] = cy[u][2];
            }
            chain_post(L.flag + 3 + half, 1);
        }
        row_sums();
        if (half == 0) RSTMP(56);
        return;
    }
    chain_wait(L.flag + 2, 1);
    if (nb > 0) { chain_wait(L.flag + 3, 1); chain_wait(L.flag + 4, 1); }
    {
        double ax[MT][3], li[3];
        scaled(m, ax);
        if (nb > 0) {
#pragma unroll
            for (int u = 0; u < MT; ++u) {
                const int rc = min(rr[u], R - 1);
                const double b0 = L.cB[rc * 9 + c0], b1 = L.cB[rc * 9 + c1], b2 = L.cB[rc * 9 + c2];
                ax[u][0] -= b0; ax[u][1] -= b1; ax[u][2] -= h2 ? b2 : 0.0;
            }
        }
        operand(L.LI + L.LIs * m, li);
#pragma unroll
        for (int u = 0; u < MT; ++u) solve_store(m, u, li, ax[u]);
    }
    row_sums();
    if (half == 0 && t0 == 0) RSTMP(55);
}

template <bool WITHQ, class SRC>
__device__ __forceinline__ void chain_eliminate(const SRC& src, const int K, const int NP, const int RS, double* Wt, const ChainLds& L, double& qacc, long long* dbg = nullptr, const bool fw = false /* rows on the matrix cores (chain_rows_mfma): the caller's waves 5 .. 7 are free, L.LI is set */) {
    const int t = vil_tid(), wave = t >> 6, lane = t & 63;
#ifdef VIL_STAMPS
    #define CSTMP(k) do { if (lane == 0 && dbg) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); dbg[k] = tt_; } } while (0)
#else
    #define CSTMP(k) do {} while (0)
#endif
    if (fw ? wave == 4 : wave >= 6) return;      // (fw: row waves 2, 3 forwards and 6, 7 backwards -- SIMDs 2 and 3; the recursion waves keep SIMDs 0 and 1 to themselves)
    const int R = NP + 1, m = K >> 1, nf = m, nb = K - 1 - m;
    if (wave < 2) {
        // ---------------- recursion wave of direction d ----------------------------------------------------------------------
        // One basic block per step.  What does not depend on the block factored before -- the raw entries, their scaling, the two u^T S' u terms -- is formed a step
        // AHEAD (prep), and the sub-diagonal rows are solved in the factorisation's own block by every lane (lanes past 8 repeat row 8, lanes past 44 entry (0, 0): the
        // same values to the same addresses, so no store is predicated and nothing waits behind a branch for the scales' LDS round trip).
        const int d = wave, nd = d == 0 ? nf : nb;
        int di = 0, dj = 0;
        { const int e = lane < 45 ? lane : 0; while ((di + 1) * (di + 2) / 2 <= e) ++di; dj = e - di * (di + 1) / 2; }
        const int dq = (di * (di + 1) >> 1) + dj, ql = min(lane, 8);
        bool ok = true;
        auto diag_entry = [&](int k) { return src.diag(k, di, dj); };
        auto diag_scaled = [&](int k, double v) { const int gi = NP + 9 * k + di, gj = NP + 9 * k + dj; double mv = src.sc(gi) * v * src.sc(gj); if (di == dj) mv += src.madd(gi); return mv; };
        auto sub_rows = [&](int k, int kn, double* a) {           // row `lane` of block kn against the columns of block k
            const int kc = min(max(kn, 0), K - 1);
#pragma unroll
            for (int c = 0; c < 9; ++c) a[c] = src.sub(k, kc, ql, c);
        };
        // scaled diagonal entry, scaled sub-diagonal row and the two terms of u^T S' u of block k (next block kn) from their raw entries
        auto prep = [&](const int k, const int kn, const double dvr, const double* ar, double& vs, double* as, double& qd, double& qs2) {
            vs = diag_scaled(k, dvr);
            qd = 0.0; qs2 = 0.0;
            if (WITHQ) qd = (di == dj ? 1.0 : 2.0) * src.u(NP + 9 * k + di) * dvr * src.u(NP + 9 * k + dj);
            const double rsc = src.sc(NP + 9 * kn + ql);
            if (WITHQ) {
                double qs = 0.0;
#pragma unroll
                for (int c = 0; c < 9; ++c) qs += ar[c] * src.u(NP + 9 * k + c);
                qs2 = 2.0 * src.u(NP + 9 * kn + ql) * qs;
            }
#pragma unroll
            for (int c = 0; c < 9; ++c) as[c] = rsc * ar[c] * src.sc(NP + 9 * k + c);
        };
        double dv = 0.0, an[9], vs = 0.0, as_[9], qd = 0.0, qs2 = 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c) { an[c] = 0.0; as_[c] = 0.0; }
        const int k0 = d == 0 ? 0 : K - 1;
        if (nd > 0) { dv = diag_entry(k0); sub_rows(k0, d == 0 ? 1 : K - 2, an); prep(k0, d == 0 ? 1 : K - 2, dv, an, vs, as_, qd, qs2); }
        if (nd == 0 && d == 0) dv = diag_entry(m);
        for (int st = 0; st < nd; ++st) {
            const int k = d == 0 ? st : K - 1 - st, kn = d == 0 ? k + 1 : k - 1, kp = d == 0 ? k - 1 : k + 1;
            double v = vs;
            if (st > 0) { const double* Lp = L.Lsb + 82 * kp;
#pragma unroll
                for (int c = 0; c < 9; ++c) v -= Lp[di * 9 + c] * Lp[dj * 9 + c]; }
            L.Ldg[54 * k + dq] = v;
            double a[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) a[c] = as_[c];
            const double qdc = qd, qsc = qs2;
            // the next step's raw entries while this one computes (the last step asks for its own again -- no branch -- and, forwards, for the middle block's diagonal)
            const bool more = st + 1 < nd;
            const int k2 = more ? kn : k, kn2 = more ? (d == 0 ? kn + 1 : kn - 1) : kn;
            const double dvn = diag_entry((more || d != 0) ? k2 : m);
            sub_rows(k2, kn2, an);
            CHAIN_FENCE();
            L9 Lf;
            ok = chol9<true>(L.Ldg + 54 * k, Lf) && ok;
            double w[9];
            row_solve9(Lf.l, Lf.r, a, w);
#pragma unroll
            for (int c = 0; c < 9; ++c) L.Lsb[82 * k + ql * 9 + c] = w[c];
            if (fw) {
                // what the row waves multiply with is the INVERSE of the factor (they work on the matrix cores, below): column min(lane, 8) by this lane, nine stores
                // instead of the factor's 54 (which cost this wave 1200 of its 5250 ticks per step: LDS stores queueing among the row waves' 650 LDS reads per step)
                double x[9];
                inv9_col(Lf, ql, x);
#pragma unroll
                for (int p = 0; p < 9; ++p) L.LI[L.LIs * k + 9 * p + ql] = x[p];      // (lanes past 8: column 8 again, the same values to the same addresses)
            } else {
#pragma unroll
                for (int e = 0; e < 45; ++e) L.Ldg[54 * k + e] = Lf.l[e];
#pragma unroll
                for (int e = 0; e < 9; ++e) L.Ldg[54 * k + 45 + e] = Lf.r[e];
            }
            if (WITHQ) { qacc += lane < 45 ? qdc : 0.0; qacc += lane < 9 ? qsc : 0.0; }
            prep(k2, kn2, dvn, an, vs, as_, qd, qs2);
            dv = dvn;
            chain_post(L.flag + d, st + 1);
            if (d == 0 && st < 6) CSTMP(48 + st);
        }
        if (d == 0) {                                  // middle block: both directions meet
            if (nb > 0) chain_wait(L.flag + 1, nb);
            {
                double v = diag_scaled(m, dv);
                if (WITHQ) qacc += lane < 45 ? (di == dj ? 1.0 : 2.0) * src.u(NP + 9 * m + di) * dv * src.u(NP + 9 * m + dj) : 0.0;
                if (nf > 0) { const double* Lp = L.Lsb + 82 * (m - 1);
#pragma unroll
                    for (int c = 0; c < 9; ++c) v -= Lp[di * 9 + c] * Lp[dj * 9 + c]; }
                if (nb > 0) { const double* Lp = L.Lsb + 82 * (m + 1);
#pragma unroll
                    for (int c = 0; c < 9; ++c) v -= Lp[di * 9 + c] * Lp[dj * 9 + c]; }
                L.Ldg[54 * m + dq] = v;
            }
            CHAIN_FENCE();
            L9 Lf;
            ok = chol9<true>(L.Ldg + 54 * m, Lf) && ok;
            if (fw) {
                const int cc = min(lane, 8);
                double x[9];
                inv9_col(Lf, cc, x);
#pragma unroll
                for (int p = 0; p < 9; ++p) L.LI[L.LIs * m + 9 * p + cc] = x[p];
            } else {
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
    if (fw) {
        // ---------------- row waves on the matrix cores (fw) --------------------------------------------------------------------------------------------------------
        // A row per lane costs every row wave ~150 fp64 instructions, ~180 LDS reads (the factor and the sub-diagonal block, broadcast) and ~150 integer instructions per
        // block, four waves of them -- with the recursion waves that SATURATES the compute unit (2600 vector instructions per block step on four SIMDs, one LDS
        // pipe): with the factor's store off the recursion wave the recursion got 5 k ticks shorter and the row waves ended where they had before.  Here the rows of a block are two small matrix products on
        // v_mfma_f64_16x16x4, sixteen rows per tile:   W_k^T = L_kk^-1 A'_k^T   and   carry^T = L_{kn,k} W_k^T   (A' = scaled rows - carry of the block before)
        // with the 9 x 9 matrices as the A operand (one load of three values per lane and block, whatever the number of tiles) and the rows as the B operand.  A lane
        // holds element (row lane & 15, column 4 s + (lane >> 4)), s = 0 .. 2, of a tile -- the layout the instruction wants for B and ALSO the layout it returns the
        // product in (accumulator g of lane (j, q) is entry (q + 4 g, j)): W^T feeds the second product, and the carry the next block's A', without leaving the registers.
        // Tiles half, half + 2, .. of a direction belong to its row wave `half` (three and two at K = 10, four each at K = 20): six instructions per tile and block.  The
        // row waves are 2, 3 (forwards) and 6, 7 (backwards), i.e. SIMDs 2 and 3: an fp64 matrix instruction holds its SIMD's fp64 pipe for 64 cycles, and the recursion
        // waves (0, 1), whose steps are the elimination's length, keep SIMDs 0 and 1 to themselves.
        // The factor's inverse comes from the recursion wave (L.LI): a product with L^-1 instead of a substitution -- the blocks are Jacobi-scaled 9 x 9, condition ~1e3.
        const int ntile = (R + 15) >> 4;
        // waves 2, 3: forwards; 6, 7: backwards, with the halves swapped -- an odd number of tiles (five at K = 10) then loads SIMDs 2 and 3 alike (3 + 2 tiles each);
        // the forward direction has a block more and ends the elimination: its waves issue first
        // Up to six tiles the forward waves carry two each and wave 5 the rest (K = 10: the fifth tile, four rows): it shares SIMD 1 with the BACKWARD recursion wave, which
        // has a block less to do -- a forward row step must not be longer than a recursion step (3750 ticks), or the rows end that much later, block after block.
        const bool small = ntile <= 6;
        if (wave == 5) {
            if (!small || ntile <= 4) return;
            __builtin_amdgcn_s_setprio(2);
            chain_rows_mfma<WITHQ, 3>(src, K, NP, RS, Wt, L, qacc, dbg, 0, 1, 4, 1);      // tile 4
            __builtin_amdgcn_s_setprio(0);
            return;
        }
        const int rd = wave >> 2, rh = rd == 0 ? (wave & 1) : 1 - (wave & 1);
        if (rd == 0) __builtin_amdgcn_s_setprio(2);
        if (small) chain_rows_mfma<WITHQ, 3>(src, K, NP, RS, Wt, L, qacc, dbg, rd, rh, rh, (rd == 0 && rh == 0) ? 2 : 3);      // (forwards: tiles 0, 2 | 1, 3, (5) | wave 5: 4)
        else chain_rows_mfma<WITHQ, 4>(src, K, NP, RS, Wt, L, qacc, dbg, rd, rh, rh, 4);
        __builtin_amdgcn_s_setprio(0);
        return;
    }
    const int d = (wave - 2) >> 1, half = (wave - 2) & 1, r = half * 64 + lane, nd = d == 0 ? nf : nb;
    volatile int* const fflag = L.flag + d;      // the recursion wave of this direction publishes its factored blocks
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
        chain_wait(fflag, st + 1);
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
