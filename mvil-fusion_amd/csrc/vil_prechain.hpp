// The speed-bias chain eliminated AHEAD of the step kernel, off the iteration's critical path (single GPU).
//
// Every entry of S' with a row or a column in the chain part comes from the IMU factors and the prior alone -- visual, LiDAR, ICP and
// LPS factors never touch a speed-bias block (estimator.cpp:1179-1186, 1189-1242, 1298-1396) -- so the chain does not have to wait for the
// gather of S'.  The gather and the step kernel are ONE launch (vil_step.hpp: k_step, rs_merged); beside the gather workgroups and the
// master, one more workgroup
//   1. gathers the chain part of S' into LDS through a table the host built at upload (destination, up to two IMU-record sources, one
//      prior source per entry: a fixed summation order, no index arithmetic on the device),
//   2. runs the two-sided chain elimination of vil_chain.hpp on it (13 us at K = 10) while the gather workgroups sum the partial records
//      (11 us) and the master judges the candidate and forms its vectors (7 us),
//   3. leaves W^T (pose rows unscaled: the Jacobi scale of a pose column needs the visual / LiDAR diagonal, which is not known here, and row
//      scaling commutes with the column elimination), the factored blocks, the chain columns' scales and the chain share of u^T S' u.
// Further workgroups then contract W W^T on the matrix cores, one 16 x 16 tile each, and the master starts its dense part from
// M_pp = Sc (S'_pp - W W^T) Sc + mu d^2: the chain (14.6 us) and its Schur contraction (5.3 us) are off the master's critical path.
// (Measured first inside k_sweep, behind the IMU / prior workgroups' flags: the chain workgroup needs 31 us there -- 8 us of it waiting for
//  the records -- against 19 us of visual work: the sweep got 12.6 us longer, the step kernel 13.7 us shorter.)
#pragma once
#include "vil_chain.hpp"

namespace vd {

// Raw chain entries in LDS, as the chain consumes them:
//   dg[k][45]   lower triangle of diagonal block k         sub[k][81]  rows of block k+1 x columns of block k (k < K-1)
//   pbc[k][3][6][9]  pose rows of frames k-1, k, k+1 x the columns of block k (what the IMU factors touch: compact, 162 per block)
//   pp[q][row]  every pose / extrinsic / td row x the q-th chain column the PRIOR holds (<= 18: the prior's speed-bias blocks are neighbours), stride NPs
//   rhs[9K]     pq[9K] (int)  chain column -> its index among the prior's columns, or -1
#define CHAIN_NPC_MAX 18
struct ChainSlab { double* dg; double* sub; double* pbc; double* pp; double* rhs; int* pq; int NPs; };
__host__ __device__ inline int chain_slab_nps(int K) { return (6 * K + 7 + 1) & ~1; }
__host__ __device__ inline size_t chain_slab_fp(int K) { return even_up(45 * K) + even_up(81 * K) + (size_t)162 * K + (size_t)CHAIN_NPC_MAX * chain_slab_nps(K) + even_up(9 * K); }      // gather targets (zeroed first)
__host__ __device__ inline size_t chain_slab_doubles(int K) { return chain_slab_fp(K) + (size_t)even_up(9 * K) / 2 + 2; }
__host__ __device__ inline ChainSlab chain_slab(double* p, int K) {
    ChainSlab S; S.NPs = chain_slab_nps(K);
    S.dg = p; S.sub = S.dg + even_up(45 * K); S.pbc = S.sub + even_up(81 * K); S.pp = S.pbc + (size_t)162 * K; S.rhs = S.pp + (size_t)CHAIN_NPC_MAX * S.NPs;
    S.pq = (int*)(S.rhs + even_up(9 * K));
    return S;
}
// LDS of the chain workgroup (doubles): chain scratch | sc, dc, u of the chain columns | slab
__host__ __device__ inline size_t prechain_lds_doubles(int K) { return chain_scratch_doubles(K) + 3 * (size_t)even_up(9 * K) + chain_slab_doubles(K) + 8; }

struct ChainSrcSlab {
    const DevP& P; const ChainSlab& B; const double* scB; const double* dcB; const double* uB; double mu;
    __device__ __forceinline__ double diag(int k, int i, int j) const { return B.dg[45 * k + (i * (i + 1) >> 1) + j]; }
    __device__ __forceinline__ double sub(int k, int kn, int q, int c) const { return kn > k ? B.sub[81 * k + q * 9 + c] : B.sub[81 * kn + c * 9 + q]; }
    // unconditional reads (clamped indices) + selects: no per-lane predicated loads
    __device__ __forceinline__ double prow(int r, int k, int c) const {
        const int fr = r < 6 * P.K ? r / 6 : 1 << 20, lr = r - 6 * fr, d = fr - (k - 1);
        const bool in = (unsigned)d <= 2u;
        const double vi = B.pbc[(size_t)((k * 3 + min(max(d, 0), 2)) * 6 + min(max(lr, 0), 5)) * 9 + c];
        const int q = B.pq[9 * k + c];
        const double vp = B.pp[(size_t)max(q, 0) * B.NPs + r];
        return (in ? vi : 0.0) + (q >= 0 ? vp : 0.0);
    }
    __device__ __forceinline__ double rhsraw(int j) const { return B.rhs[j - P.NV]; }
    __device__ __forceinline__ double sc(int j) const { return scB[j - P.NV]; }
    __device__ __forceinline__ double madd(int j) const { const double d = dcB[j - P.NV]; return mu * d * d; }
    __device__ __forceinline__ double rowscale(int) const { return 1.0; }
    __device__ __forceinline__ double u(int j) const { return uB[j - P.NV]; }
    __device__ __forceinline__ void row_done(int d, int r, double zr, double&) const { st_ag(P.chZ + (size_t)d * (P.NV + 1) + r, zr); }
    __device__ __forceinline__ void wput(double* p, double v) const { st_ag(p, v); }      // read by the tile workgroups and the master of this launch
};

// Column cc of the INVERSE of the factored diagonal block k (x = L^-1 e_c by forward substitution) and row cc of M_k = L_kk^-T Ls_k^T (the lane that
// holds column cc of the inverse forms it): the master's back substitution is x_k = L_kk^-T t_k - M_k x_next, ONE nine-term product per block instead of
// two with an LDS round trip in between.  (The middle block has no Ls: its row is never read.)  Ldg / Lsb: the chain's factors (LDS or global).
__device__ __forceinline__ void chain_inverse_block(const DevP& P, const double* Ldg, const double* Lsb, const int k, const int cc, const double* LI = nullptr /* the inverses are there already (chain_eliminate, fw: ChainLds::LI) */, const int LIs = 0) {
    const double* l = Ldg + 54 * k; const double* r = l + 45;
    double x[9];
    if (LI) {
#pragma unroll
        for (int p = 0; p < 9; ++p) x[p] = LI[LIs * k + 9 * p + cc];
    } else {
#pragma unroll
    for (int p = 0; p < 9; ++p) {
        double acc = p == cc ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < p; ++q) acc -= l[(p * (p + 1) >> 1) + q] * x[q];
        x[p] = p < cc ? 0.0 : acc * r[p];
    }
    }
#pragma unroll
    for (int p = 0; p < 9; ++p) if (p >= cc) st_ag(P.chLdg + 54 * k + (p * (p + 1) >> 1) + cc, x[p]);
    const double* ls = Lsb + 82 * k;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int p = 0; p < 9; ++p) { if (p & 1) a1 += x[p] * ls[i * 9 + p]; else a0 += x[p] * ls[i * 9 + p]; }
        st_ag(P.chLsb + 82 * k + 9 * cc + i, k == (P.K >> 1) ? 0.0 : a0 + a1);
    }
}
// the same for every block from the raw factors the in-sweep chain workgroup left (windows the merged launch cannot hold): one extra workgroup of the STEP kernel's launch, beside the
// master -- which reads the result 40 us after its entry, behind chflag[2].  lds: 136 K doubles (the factors are staged with whole-line loads: read in
// place every lane would walk 135 loads at memory latency).  (As a workgroup of the gather launch it was that launch's longest.)
__device__ __forceinline__ void prechain_inverses(const DevP& P, double* lds, const int epoch) {
    const int n = 136 * P.K;
    for (int e = vil_tid(); e < n; e += blockDim.x) lds[e] = P.chLraw[e];
    __syncthreads();
    for (int it = vil_tid(); it < 9 * P.K; it += blockDim.x) chain_inverse_block(P, lds, lds + 54 * P.K, it / 9, it % 9);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (vil_tid() == 0) st_ag(P.chflag + 2, epoch);
}

// the chain workgroup (all threads of the block enter; dynamic LDS >= prechain_lds_doubles(K)); the IMU / prior records are complete
// epoch: the launch's flag value; P.chflag[1] is posted as soon as the chain columns' scales are out (the master's vector pass reads them).
// wait_records: the workgroup rides in k_sweep (windows too large for the merged launch) and spins until the IMU / prior workgroups of
// that launch have published their records (swflag).
// FUSED: the one-launch iteration (vil_iter.hpp) -- the IMU / prior workgroups are workgroups of THIS launch too (wait_records = true, their flags are P.sflag
// with the launch epoch), but the launch's longest path is the master's, not this workgroup's: it forms the inverses of its diagonal blocks itself.
template <bool FUSED = false>
__device__ __forceinline__ void prechain_wg(const DevP& P, const Ctl& ctl, const int jacobi, double* lds, const int epoch, const bool wait_records = false) {
    const int t = vil_tid(), K = P.K, NP = P.NV, NB = 9 * K, NT = blockDim.x;
#ifdef VIL_STAMPS
    #define PSTAMP(k) do { if (t == 0) { long long tt_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory"); P.dbg[k] = tt_; } } while (0)
#else
    #define PSTAMP(k) do {} while (0)
#endif
    PSTAMP(12);
    ChainLds L = chain_lds(lds, K);
    double* scB = lds + chain_scratch_doubles(K); double* dcB = scB + even_up(NB); double* uB = dcB + even_up(NB);
    const ChainSlab B = chain_slab(uB + even_up(NB), K);
    L.LI = B.sub; L.LIs = 81;                          // the inverses of the factored blocks take the place of the raw sub-diagonal blocks: block k's is written when the
                                                       // elimination of block k ends, its raw sub-diagonal block (k, k + 1) was read a step before (forwards: by that very
                                                       // step's look-ahead; backwards: when block k + 1 was eliminated)
    if (t < 8) chain_flag_set(L.flag + t, 0);
    // ---- gather of the chain part of S' out of the IMU / prior records through the host-built table: eight entries per thread and round (one round at
    //      K = 10, two at K = 20).  Two dependent round trips -- entries, then their sources -- and NOTHING before them: the zeroing of the slab (a gather target: entries
    //      without a source, pose rows of far frames, stay zero) and the small copies run while the sources are in flight; only the stores wait for the barrier
    const int4* tab = (const int4*)P.chtab;
    const int n = P.n_chtab;
    double* slab = B.dg;
    constexpr int QN = 8;                              // entries per thread and round (K = 10: one round; K = 20, 6840 entries: two)
    int4 q[QN], q2[QN]; double a[QN], b[QN], c[QN];
    // (IMU sources of an entry: index into the 931-double records + 1 in the low half, index into the compact tagged records of a one-launch iteration + 1 in the high half)
    auto isrc = [&](const int v) { return FUSED ? (v >> 16) - 1 : (v & 0xffff) - 1; };      // (decoded where it is used: the entries stay as loaded -- sixteen int4 per thread)
#pragma unroll
    for (int u = 0; u < QN; ++u) { q[u] = tab[min(t + u * NT, n - 1)]; q2[u] = tab[min(t + (QN + u) * NT, n - 1)]; }      // (the second round's entries too: no table round trip behind the flags)      // (the second round's entries too: no table round trip behind the flags)
    const double sc_prev = (t < NB && !ctl.first) ? ldx<FUSED>(P.Sc + NP + t) : 0.0;      // (written by the master workgroup of an earlier iteration)
    const int pqv = P.chpq[min(t, NB - 1)];
    const double* const pHs = P.pn > 0 ? P.pH : P.mpart;
    // (IMU / prior records: agent-scope loads -- inside k_sweep they were written by workgroups of this launch; unconditional loads + selects)
    auto sources = [&](const int e0) {
        if constexpr (FUSED) {
            // one-launch iteration.  This workgroup is ONE compute unit pulling its sources across the device with agent-scope loads, which cost per LANE (measured:
            // ~5 lanes per ns, whatever the lines they fall into), so
            //  - only the sources an entry has are loaded (fewer than half of the three are there: the prior's gradient in 9 K entries of ~3000);
            //  - the prior's constant share (J0^T J0 entries) is read from the matrix by the first iteration of a solve only, which leaves it in table order for the
            //    later ones -- plain, coalesced loads;
            //  - the IMU roles leave what is gathered here as compact records (chain_rec_index, vil_dev.hpp) and the table is sorted by source.
#pragma unroll
            for (int u = 0; u < QN; ++u) {
                a[u] = 0.0; b[u] = 0.0; c[u] = 0.0;
                const int ya = isrc(q[u].y), za = isrc(q[u].z);
                if (ya >= 0) a[u] = ld_ag(P.irec + ya);
                if (za >= 0) b[u] = ld_ag(P.irec + za);
                const int cw = q[u].w;
                if (cw < -1) c[u] = ld_ag(P.mpart - cw - 2);
                else if (cw >= 0) c[u] = ctl.first ? ld_ag(pHs + cw) : P.chc[min(e0 + u * NT, n - 1)];
            }
            if (ctl.first) {
#pragma unroll
                for (int u = 0; u < QN; ++u) if (e0 + u * NT < n && q[u].w >= 0) P.chc[e0 + u * NT] = c[u];
            }
        } else {
#pragma unroll
        for (int u = 0; u < QN; ++u) {
            a[u] = ld_ag(P.ipart + max(isrc(q[u].y), 0)); b[u] = ld_ag(P.ipart + max(isrc(q[u].z), 0));
            const int cw = q[u].w;
            c[u] = ld_ag(cw >= 0 ? pHs + cw : P.mpart + max(-cw - 2, 0));      // (no prior: cw is never >= 0)
        }
        }
    };
    if (!wait_records) sources(t);
    { double* z = B.dg; const int nz = (int)chain_slab_fp(K); for (int e = t; e < nz; e += NT) z[e] = 0.0; }
    if (t < NB) B.pq[t] = pqv;
    for (int e = t + NT; e < NB; e += NT) B.pq[e] = P.chpq[e];
    if (wait_records) {
        const int ep = FUSED ? epoch : (int)((((unsigned)ctl.gen) << 12) + (unsigned)ctl.swe + 1u);       // (sweep_signal, vil_sweep.hpp)
        const int* const fl = FUSED ? P.sflag : P.swflag;
        if (t <= P.n_imu && (t < P.n_imu || P.pn > 0)) spin_until_eq((FUSED && t < P.n_imu) ? P.cflag + t : fl + t, ep, P.abortf);      // (one-launch iteration: the IMU roles' compact records have their own, earlier flag)
    }
    __syncthreads();
    PSTAMP(31);
    if (FUSED && t == 0) prof_stamp(P, epoch - 1, 3);
    if (wait_records) sources(t);                           // (inside k_sweep the records are complete only behind the flags)
    for (int e0 = t; e0 < n; e0 += QN * NT) {
        if (e0 != t) {
#pragma unroll
            for (int u = 0; u < QN; ++u) q[u] = e0 == t + QN * NT ? q2[u] : tab[min(e0 + u * NT, n - 1)];
            sources(e0);
        }
#pragma unroll
        for (int u = 0; u < QN; ++u) if (e0 + u * NT < n) {
            const double v = (isrc(q[u].y) >= 0 ? a[u] : 0.0) + (isrc(q[u].z) >= 0 ? b[u] : 0.0) + (q[u].w != -1 ? c[u] : 0.0);
            slab[q[u].x] = v;
        }
    }
    __syncthreads();
    PSTAMP(13);
    if (FUSED && t == 0) prof_stamp(P, epoch - 1, 14);
    const ChainSrcSlab src{P, B, scB, dcB, uB, ctl.mu};
    if (t < NB) {
        const int j = NP + t;
        const double dg = src.diag(t / 9, t % 9, t % 9), b = src.rhsraw(j);
        const double Sc = ctl.first ? (jacobi ? 1.0 / (1.0 + sqrt(dg)) : 1.0) : sc_prev;
        const double d = sqrt(fmin(fmax(Sc * Sc * dg, 1e-6), 1e32));
        scB[t] = Sc; dcB[t] = d; uB[t] = Sc * (Sc * b / d) / d;
    }
    __syncthreads();
    // the scales leave for the master on a wave the elimination does not use (chain_eliminate runs on waves 0 .. 5): the round trip of the stores
    // and the flag behind them stay off the chain's path
    const bool fw = (FUSED || !wait_records) && 6 * K + 8 <= 128 && !(P.skip_mask & 2048);      // (rows of the elimination on the matrix cores, waves 2, 3, 6, 7: up to eight row tiles; waves 6 / 7 are free unless this workgroup rides in k_sweep, where they carry the raw factors out, below)
    const int tsc = fw ? 256 : 384;                     // a wave the elimination does not use
    if (t >= tsc && t < tsc + 64) {
        for (int i = t - tsc; i < NB; i += 64) { st_ag(P.chSc + i, scB[i]); st_ag(P.chDc + i, dcB[i]); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == tsc) st_ag(P.chflag + 1, epoch);
    }
    PSTAMP(14);
    // in-sweep chain (fallback launch structure): the factored blocks leave for prechain_inverses (a workgroup of the gather launch) as they are published, on the two waves
    // the elimination leaves idle -- 54 + 82 doubles per block, three stores per lane; from the recursion wave itself (one lane, 54 stores in a row) they
    // cost it 1.7 us per block, as a pass after the elimination 5 us at the end of the launch's longest workgroup
    if (wait_records && !FUSED && t >= 384) {
        const int d = (t >> 6) - 6, lane = t & 63, m = K >> 1, nd = d == 0 ? m : K - 1 - m;
        for (int st = 0; st <= nd; ++st) {
            int k;
            if (st < nd) { chain_wait(L.flag + d, st + 1); k = d == 0 ? st : K - 1 - st; }
            else { if (d != 0) break; chain_wait(L.flag + 2, 1); k = m; }           // the middle block: factored last
            if (lane < 54) st_ag(P.chLraw + 54 * k + lane, L.Ldg[54 * k + lane]);
            for (int e = lane; e < 82; e += 64) st_ag(P.chLraw + 54 * K + 82 * k + e, L.Lsb[82 * k + e]);
        }
    }
    double qc = 0.0;
    chain_eliminate<true>(src, K, NP, P.chain_rs, P.chW, L, qc, P.dbg, fw);      // (P.dbg: stamps of the VIL_STAMPS build)
    PSTAMP(15);
    if (t < 128) {                                     // the two recursion waves hold the chain x chain share of u^T S' u
        qc = wave_total(qc);
        if ((t & 63) == 0) st_ag(P.chQ + (t >> 6), qc);
    }
    if (t == 0) st_ag(P.chOk, chain_flag_get(L.flag + 5) ? 0 : 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every thread's W^T / chZ / chQ stores are out ...
    __syncthreads();
    if (t == 0) { st_ag(P.chflag, epoch); if (FUSED) prof_stamp(P, epoch - 1, 4); }                  // ... W^T is complete: the tile workgroups start
    // ---- off the critical path: what the master needs for the chain BACK substitution, 30 us from now (chain_inverse_block above).  Beside the gather
    //      (merged launch) the 9 K columns are formed here, one per lane, and are ready long before they are read.  Inside k_sweep (fallback launch structure) this workgroup
    //      is the launch's longest and 7 us of dependent fp64 chains at its end would be 7 us of the iteration: the raw factors go out instead and a
    //      workgroup of the step launch forms the columns (prechain_inverses above).
    //      (waves 6 / 7 stored them as they were published, above)
    if (!wait_records || FUSED) {
        for (int it = t; it < 9 * K; it += NT) chain_inverse_block(P, L.Ldg, L.Lsb, it / 9, it % 9, fw ? L.LI : nullptr, L.LIs);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) st_ag(P.chflag + 2, epoch);
    }
    PSTAMP(16);
}

// One 16 x 16 tile (I, J), J <= I, of W W^T per workgroup (its first 256 threads: the four waves split the chain columns, their
// accumulators are added through LDS).  W^T: column j of the chain at chW[j * RS + row], rows = pose part + the right-hand-side row.
// Output in the tiled lower layout of the step kernel (TILE_RS = 17).
// FUSED: the four waves' accumulators meet in the workgroup's DYNAMIC LDS (`lds`, >= 1024 doubles) -- the one-launch iteration carries no static LDS, so
// that its dynamic size is the larger of the sweep roles' and the step roles' needs, not their sum
template <bool FUSED = false>
__device__ __forceinline__ void prechain_ww_tile(const DevP& P, const int tile, double* const lds = nullptr) {
    typedef double d4_ __attribute__((ext_vector_type(4)));
    double (*acc_s)[256];
    if constexpr (FUSED) acc_s = reinterpret_cast<double (*)[256]>(lds);
    else { __shared__ double acc_st[4][256]; acc_s = acc_st; }
    if (!FUSED && vil_tid() >= 256) return;
    const bool act = vil_tid() < 256;                  // (one-launch iteration: the upper waves idle through the barrier instead of leaving)
    const int t = vil_tid() & 255, wave = t >> 6, lane = t & 63, row = lane & 15, kq = lane >> 4;
    const int NB = 9 * P.K, RS = P.chain_rs, R = P.NV + 1;
    int I = 0; while ((I + 1) * (I + 2) / 2 <= tile) ++I;
    const int J = tile - I * (I + 1) / 2;
    const bool va = (I << 4) + row < R, vb = (J << 4) + row < R;
    const double* pa = P.chW + (size_t)kq * RS + (I << 4) + row;
    const double* pb = P.chW + (size_t)kq * RS + (J << 4) + row;
    d4_ c4 = {0.0, 0.0, 0.0, 0.0};
    // k-steps of four chain columns, dealt round-robin to the waves; rows beyond R / columns beyond NB are masked (chW is padded, never read out of bounds)
    const int nk = act ? (NB + 3) >> 2 : 0;
    // (eight k-steps per round with all sixteen operand loads in flight before the first MFMA: one L2 round trip per round -- a load, a wait and an
    //  MFMA per k-step was six round trips in a row at K = 10, on the path between the end of the chain and the start of the dense factorisation)
    for (int ks0 = wave; ks0 < nk; ks0 += 32) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ks = min(ks0 + 4 * u, nk - 1);
            av[u] = ld_ag(pa + (size_t)4 * ks * RS); bv[u] = ld_ag(pb + (size_t)4 * ks * RS);      // written by the chain workgroup of this launch
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ks = ks0 + 4 * u;
            const bool kv = ks < nk && 4 * ks + kq < NB;
            c4 = __builtin_amdgcn_mfma_f64_16x16x4f64((va && kv) ? av[u] : 0.0, (vb && kv) ? bv[u] : 0.0, c4, 0, 0, 0);
        }
    }
    // accumulator element g of a lane = (row (lane >> 4) + 4 g, column lane & 15) of the tile
    if (act) {
#pragma unroll
        for (int g = 0; g < 4; ++g) acc_s[wave][(((lane >> 4) + 4 * g) << 4) + (lane & 15)] = c4[g];
    }
    __syncthreads();
    if (act) {
        const int r = t >> 4, c = t & 15;
        const double v = (acc_s[0][t] + acc_s[1][t]) + (acc_s[2][t] + acc_s[3][t]);
        st_ag(P.chWW + (size_t)tile * (16 * 17) + r * 17 + c, v);
    }
}

}  // namespace vd
