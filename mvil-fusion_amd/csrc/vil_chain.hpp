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

namespace vd {

__host__ __device__ inline int chain_rs(int K) { int rs = ((6 * K + 8 + 15) >> 4) << 4; if ((rs & 31) != 16) rs += 16; return rs; }   // row stride of W^T: >= 16 T, = 16 mod 32 (LDS banks)
__host__ __device__ inline int even_up(int v) { return (v + 1) & ~1; }
// doubles of chain scratch behind the tile array (and W^T): Dk 2x81 | Ls 2x81 | carry of the backward direction R x 9 |
// L_kk (45 + 9 reciprocal pivots) per block | sub-diagonal block per block | t (9K)
__host__ __device__ inline size_t chain_scratch_doubles(int K) { const int R = 6 * K + 8; return 162 + 162 + (size_t)9 * R + even_up(54 * K) + even_up(81 * K) + even_up(9 * K) + 16; }

struct L9 { double l[45]; double r[9]; };     // lower factor, l[i(i+1)/2 + j], and reciprocal pivots

// Cholesky of the 9 x 9 block whose lower triangle sits at Dk[i * 9 + j] (LDS, same address in every lane: broadcast reads).
// Fully unrolled, register-resident; returns false on a non-positive pivot.
__device__ __forceinline__ bool chol9(const double* Dk, L9& o) {
    double a[45];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) a[(i * (i + 1) >> 1) + j] = Dk[i * 9 + j];
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        const double dpp = a[(p * (p + 1) >> 1) + p];
        ok = ok && dpp > 0.0 && isfinite(dpp);
        double sq, rs;
        sqrt_rsqrt(dpp, sq, rs);
        o.l[(p * (p + 1) >> 1) + p] = sq; o.r[p] = rs;
#pragma unroll
        for (int i = p + 1; i < 9; ++i) o.l[(i * (i + 1) >> 1) + p] = a[(i * (i + 1) >> 1) + p] * rs;
#pragma unroll
        for (int j = p + 1; j < 9; ++j)
#pragma unroll
            for (int i = j; i < 9; ++i) a[(i * (i + 1) >> 1) + j] -= o.l[(i * (i + 1) >> 1) + p] * o.l[(j * (j + 1) >> 1) + p];
    }
    return ok;
}

#define CHAIN_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// One block of the chain back substitution, executed by ONE wave:  x_k = L_kk^-T (t_k - Ls_k^T x_next).
// Ldg: 45 + 9 doubles of block k; Lsb: its sub-diagonal block (rows of the neighbouring block that was eliminated after it)
// or nullptr for the middle block; tk / xn / xo: LDS.  Lanes 0..8 form the right-hand side, lane 0 solves.
__device__ __forceinline__ void chain_block_back(const double* Ldg, const double* Lsb, double* tk, const double* xn, double* xo) {
    const int lane = threadIdx.x & 63;
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
