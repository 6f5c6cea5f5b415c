// One trust-region iteration in ONE launch (single GPU, chain windows): the factor sweep, the gather of its records, the elimination of the speed-bias chain
// and the step (judge, dense solve, dogleg, candidate) as roles of a single grid
//   [ imu x n_imu | prior | rel ]  [ chain ]  [ visual x n_vwg | plane | edge ]  [ master | helpers x n_help | W W^T tiles x n_ww ]  [ gather x n_gather ]
// instead of k_sweep followed by the merged gather + step launch (k_step, rs_merged).  What the second launch could not overlap is what this one buys:
//   * the chain workgroup starts behind the IMU / prior workgroups' flags (~5 us into the launch) and eliminates the chain UNDER the visual workgroups and
//     the gather -- in the two-launch structure it was one of the two ~20 us legs the master waited for before its dense factorisation;
//   * the gather workgroups are resident and staged when the last visual record lands (no launch ramp between them);
//   * one launch boundary per iteration instead of two.
// The sweep roles and the chain workgroup wait for LOWER block indices only (the hardware dispatches in index order: what a resident workgroup waits for is
// resident or done).  Master, helpers and tile workgroups wait for one another and for the gather workgroups BEHIND them in the grid: they are few, and the launch
// is only taken when the device holds all of them plus two more workgroups at once (vil_coop.hpp), so the gather workgroups always find compute units to run through.  Everything that crosses workgroups inside
// the launch is stored and loaded at agent scope (st_ag / ld_ag; template parameters AG / FUSED of the roles), flags carry the launch epoch
// (solve generation, launches so far) -- no fences, no atomics on data.
// The kernel has NO static LDS: StepShared and the scratch arrays of the gather / tile roles are carved from the dynamic allocation, whose size is the LARGER
// of what the sweep roles and the step roles need (vilsolve.hip: lds_iter), not their sum.
#pragma once
#include "vil_sweep.hpp"
#include "vil_step.hpp"

#define VIL_LC_DOUBLES 160      // k_solve: a sweep role's copy of Ctl at the front of its dynamic LDS
#define VIL_XL_DOUBLES 352      // ... and its copy of the state's camera part behind it (16 K + 8 doubles, K <= 15: one polling thread per 32-bit half)
#define VIL_SS_DOUBLES ((sizeof(vd::StepShared) + 15) / 16 * 2)      // StepShared at the front of the step roles' dynamic LDS (16-byte granules)
static_assert(VIL_SWEEP_THREADS == VIL_STEP_THREADS, "one block size for every role of k_iter");

#ifndef VIL_PERSIST_TU
template <int TS>      // accumulator tiles per wave of the visual role (k_sweep<TS>)
__global__ __launch_bounds__(VIL_STEP_THREADS) void k_iter(DevP P, SolveOpts O) {
    extern __shared__ double dyn[];
    // grid order = dispatch order: [imu | prior | rel] [chain] [visual | plane | edge] [master | helpers | tiles] [gather].  The chain workgroup waits for the first group
    // only and is the head of the longest path into the dense factorisation: it must not queue behind hundreds of visual / LiDAR workgroups (configs[2]: 600 sweep
    // roles on 256 compute units: its records were seen at 20 us instead of 10).  Master, helpers and tiles stay BEHIND the sweep roles: 24 - 45 waiting workgroups
    // ahead of them cost the visual roles a second dispatch round at K = 20 (sweep phase 26 -> 35 us, measured); ahead of the gather workgroups their prologues
    // still run under the sweep.  sweep role index (= its flag in P.sflag): the order of k_sweep
    // profiling (vil_profile_workgroups): every workgroup of ONE chosen launch leaves its entry and exit time -- when a role is dispatched and how long it stays is
    // what the phase stamps cannot say (it found the gather workgroups of K = 20 queueing in three rounds behind the launch's LDS footprint)
    const unsigned long long wg_t_in = P.prof ? wall_clock64() : 0ull;
    auto wg_times = [&](const int launch) {
        if (P.prof && launch == P.wg_launch && threadIdx.x == 0 && blockIdx.x < VIL_PROF_WGS) {
            unsigned long long* d = (unsigned long long*)P.prof + 64 * VIL_PROF_SLOTS + 2 * blockIdx.x;
            d[0] = wg_t_in; d[1] = wall_clock64();
        }
    };
    const int b = (int)blockIdx.x, n_early = P.n_imu + 2, n_wait = 1 + P.n_help + P.n_ww;
    int sw = -1, p0 = -1;
    if (b < n_early) sw = b;
    else if (b == n_early) p0 = 0;                                  // chain
    else if (b <= P.n_sw) sw = b - 1;                               // visual | plane | edge
    else p0 = b - P.n_sw;                                           // 1 .. n_wait: master | helpers | tiles; then the gather workgroups
    (void)n_wait;
    if (sw >= 0) {
        const Ctl ctl = *P.ctl;
        if (ctl.done) return;
        sweep_body<TS, true>(P, O, ctl, dyn, sw);
        wg_times(ctl.n_sweeps);
        return;
    }
    vd::StepShared& s = *reinterpret_cast<vd::StepShared*>(dyn);
    step_body<true, 3, true>(P, O, s, dyn + VIL_SS_DOUBLES, p0);
    wg_times(s.c.n_sweeps - 1);      // (the step roles have counted this launch already)
    // The first launch that finds the solve FINISHED writes the result out (accepted state -> x[0], gauge fix, Ctl + state into the host mirror, the sequence word the
    // host polls): its master workgroup, which has nothing else to do -- every role returned at once.  A chunk of enqueued iterations may then be sized generously:
    // the host is released behind the first dead launch (~5 us), not behind the chunk's no-op tail and k_finish (which still covers a solve that ends in the last
    // launch of its chunk).  ONE call site, outside the step roles: inlined at the step's five exits the parameter block went to scratch memory (round 4).
    if (p0 == 1) {
        __syncthreads();
        if (s.done_at_entry && s.c.outd == 0 && s.c.lin_mode == 0)
            vd::solve_finish(P.x[0], P.x[1], P.xorig, P.hstate, P.ctl, P.hctl, P.hseq, P.K, P.NS, P.gauge_on, s.c.cur, s.c.status, s.c.gen, dyn + VIL_SS_DOUBLES);
    }
}

#endif
#ifdef VIL_PERSIST_TU
// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
// The WHOLE SOLVE in one launch (round 6): the roles of k_iter as a RESIDENT grid that loops over the trust-region iterations.  What a launch boundary cost per
// iteration -- the dispatch ramp, the end-of-kernel write-back, the gap to the next dispatch: ~10 us of 61 at configs[1], measured as (host clock per iteration) -
// (first workgroup in .. master done) -- becomes one 64-byte line: the master posts what a sweep role reads of Ctl, tagged with the iteration's epoch, behind its Ctl /
// candidate stores (P.ihdr); every other workgroup polls it.
//   grid = [imu x n_imu | prior | rel] [chain] [visual x n_vwg | plane | edge] [master | helpers x n_help | W W^T tiles x n_ww]        (no gather workgroups)
// * EVERY workgroup must be resident at once (a workgroup that is not never runs: nobody leaves before the solve ends): the launch is only taken when the grid is
//   smaller than what the device holds (vilsolve.hip: c->persist), i.e. configs[1]-sized windows (186 workgroups) -- K = 20 and configs[2] keep k_iter.
// * The gather items are therefore DUTIES of workgroups that would otherwise idle: tile workgroups (idle until the chain is eliminated), helpers (idle until the
//   gather is complete), then the sweep roles once their own record is out -- the short roles (IMU, prior, ICP / LPS, LiDAR) before the visual ones; one item each,
//   so no item queues behind another (the launch is not taken when there are more items than such workgroups).
// * What crosses from one iteration to the next crosses at agent scope (the FUSED loads / stores of the roles: Ctl, the candidate's camera part, la / lb, Sc);
//   flags carry the epoch (generation, iteration) and are never reset, as in k_iter.
// * Time cap (ceres max_solver_time_in_seconds, estimator.cpp:1411): the MASTER reads the device's wall clock at the end of an iteration -- where ceres reads its own,
//   at the top of the next -- and ends the solve with the accepted state; the step it has just formed is not counted.
// * A wait that gives up (vil_math.hpp) ends every role's loop; the host re-runs the solve with one launch per iteration (vil_solve_resident).
// * The parameter block is read through an OPAQUE pointer to the kernarg segment, renewed every iteration (as the thread index is, vil_math.hpp): nothing a role derives
//   from it can be hoisted in front of the iteration loop -- with the by-value block LLVM hoisted every address of every role and the allocator spilled them.
struct KSolveArgs { DevP P; SolveOpts O; long long budget_ticks; /* 100 MHz; <= 0: no time cap */ };
__device__ __forceinline__ const KSolveArgs& ksolve_args() {
    typedef const __attribute__((address_space(4))) KSolveArgs* KP;
    KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    return *(const KSolveArgs*)kp;
}
template <int TS>
__global__ __launch_bounds__(VIL_STEP_THREADS) void k_solve(KSolveArgs kernarg_block /* read through ksolve_args() */) {
    using vd::ld_ag; using vd::st_ag; using vd::spin_until_eq;
    extern __shared__ double dyn[];
    const KSolveArgs& A0 = ksolve_args();
    const DevP& P = A0.P;
    const long long budget_ticks = A0.budget_ticks;
    const int b = (int)blockIdx.x, t = (int)threadIdx.x, n_early = P.n_imu + 2;
    const unsigned long long t_start = wall_clock64();
    int sw = -1, p0 = -1;
    if (b < n_early) sw = b;
    else if (b == n_early) p0 = 0;                                  // chain
    else if (b <= P.n_sw) sw = b - 1;                               // visual | plane | edge
    else p0 = b - P.n_sw;                                           // 1: master; 2 .. 1 + n_help: helpers; then the tile workgroups
    // gather duty of this workgroup (-1: none): candidates in the order [tiles | helpers | short sweep roles | visual roles]
    const int n_nv = P.n_sw - P.n_vwg;
    int ci = -1;
    if (sw >= 0) ci = P.n_ww + P.n_help + (sw < n_early ? sw : (sw >= n_early + P.n_vwg ? sw - P.n_vwg : n_nv + (sw - n_early)));
    else if (p0 >= 2 + P.n_help) ci = p0 - 2 - P.n_help;
    else if (p0 >= 2) ci = P.n_ww + (p0 - 2);
    const int item = (ci >= 0 && ci < P.n_gather) ? ci : -1;
    Ctl* const lc = reinterpret_cast<Ctl*>(dyn);                    // a sweep role's copy of Ctl (the step roles keep theirs in StepShared, at the same place)
    static_assert(sizeof(Ctl) % 8 == 0 && sizeof(Ctl) / 8 <= VIL_LC_DOUBLES, "Ctl copy of the sweep roles");
    // ---- the hand-over between two iterations: ONE 64-byte line of seven tagged words {payload 32 bits, epoch of the iteration that ended} -- what a sweep role reads of
    //      Ctl (cur / done / first / lin_mode, mu, cg, cn; gen and the iteration count are the epoch itself).  The master writes it behind its Ctl and candidate
    //      stores; a sweep role polls it with seven lanes and is in its next iteration one round trip after the master's store -- no flag followed by a load of Ctl.
    //      The step roles poll word 0 and then read Ctl itself (they have slack: none of them is the start of an iteration's longest path).
    auto hdr_ld = [&](const int k) { return __hip_atomic_load(P.ihdr + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (sw >= 0) {
        // a sweep role's dynamic LDS: [its copy of Ctl | its copy of the state's camera part | the role's own arrays (IMU roles: + the gather duty's scratch behind them)]
        double* const xl = dyn + VIL_LC_DOUBLES; double* const sm = xl + VIL_XL_DOUBLES;
        const int NC = 16 * A0.P.K + 8;
        { const double* src = (const double*)P.ctl; double* dst = (double*)lc; for (int i = t; i < (int)(sizeof(Ctl) / 8); i += VIL_STEP_THREADS) dst[i] = ld_ag(src + i); }      // (the first iteration: as the init launch left them)
        __syncthreads();
        { const double* x0 = A0.P.x[1 - lc->cur]; for (int i = t; i < NC; i += VIL_STEP_THREADS) xl[i] = ld_ag(x0 + i); }
        __syncthreads();
        bool resident = false;                                      // IMU roles: constants, sqrt-information and constancy flags stay in LDS from the first iteration on
        for (;;) {
            const KSolveArgs& A = ksolve_args();
            const DevP& P = A.P; const SolveOpts& O = A.O;
            const Ctl& ctl = *lc;
            if (ctl.done) return;
            const int epoch = (int)((((unsigned)ctl.gen) << 12) + (unsigned)ctl.n_sweeps + 1u);
            sweep_body<TS, true>(P, O, ctl, sm, sw, xl, resident && !(P.skip_mask & 256));
            resident = true;
            if (item >= 0) {
                int4* const scratch = (int4*)(sm + (sw < P.n_imu ? 2048 : 0));
                __syncthreads();
                reduce_gather<true, VIL_STEP_THREADS / 8, true>(P, ctl, item, scratch, epoch);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
                if (t == 0 && !(P.drop_role == -2 - item && ctl.n_sweeps == P.drop_launch)) { st_ag(P.gflag + item, epoch); prof_stamp(P, epoch - 1, 5); }      // (drop: test hook, vil_debug_drop_flag)
            }
            __syncthreads();                                        // every reader of lc / xl is through
            // the hand-over: lanes 0 .. 6 poll the header words, threads 8 .. 8 + 2 NC the candidate's tagged halves -- ONE round trip behind the master's stores brings
            // the flag AND the data (a `done` header releases the threads whose words will never come)
            if (t < 7 || (t >= 8 && t < 8 + 2 * NC)) {
                const unsigned long long* const wp = t < 7 ? P.ihdr + t : P.xtag + (t - 8);
                unsigned long long w = 0, tw0 = 0; bool ok = true, have = false;
                for (int sp = 1;; ++sp) {
                    w = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned long long h0 = t < 7 ? w : hdr_ld(0);
                    if ((unsigned)(w >> 32) == (unsigned)epoch) { have = true; break; }
                    if ((unsigned)(h0 >> 32) == (unsigned)epoch && ((unsigned)h0 & 2u)) break;      // the solve has ended
                    __builtin_amdgcn_s_sleep(4);
                    if ((sp & 1023) == 0 && vd::wait_expired(tw0, P.abortf)) { st_ag(P.abortf, 1); ok = false; break; }
                }
                const unsigned pl = (unsigned)w;
                if (!ok) lc->pad_ = 1;
                else if (!have) {}
                else if (t == 0) { lc->cur = pl & 1; lc->done = (pl >> 1) & 1; lc->first = (pl >> 2) & 1; lc->lin_mode = (pl >> 3) & 3; lc->n_sweeps = (int)((unsigned)epoch & 0xfffu); }
                else if (t < 7) { unsigned* d = (unsigned*)(t <= 2 ? &lc->mu : (t <= 4 ? &lc->cg : &lc->cn)); d[(t - 1) & 1] = pl; }      // (little endian: low half first)
                else ((unsigned*)xl)[t - 8] = pl;
            }
            if (sw == 0 && t == 0) prof_stamp(P, epoch - 1, 27);
            __syncthreads();
            if (sw == 0 && t == 0) prof_stamp(P, epoch - 1, 28);
            if (lc->pad_) return;                                   // a wait gave up: the host re-runs the solve (vil_solve_resident)
            if (P.skip_mask & 512) { const double* xg = P.x[1 - lc->cur]; for (int i = t; i < NC; i += VIL_STEP_THREADS) xl[i] = ld_ag(xg + i); __syncthreads(); }      // (debug: the candidate from memory)
        }
    }
    vd::StepShared& s = *reinterpret_cast<vd::StepShared*>(dyn);
    double* const Alds = dyn + VIL_SS_DOUBLES;
    for (;;) {
        const KSolveArgs& A = ksolve_args();
        const DevP& P = A.P; const SolveOpts& O = A.O;
        __syncthreads();
        step_body<true, 3, true>(P, O, s, Alds, p0, item, budget_ticks > 0 ? (long long)t_start + budget_ticks : 0ll);
        __syncthreads();
        if (s.done_at_entry) return;                                // (chain, helpers, tiles: the master has ended the solve -- it does not come back here itself)
        const int epoch = (int)((((unsigned)s.c.gen) << 12) + (unsigned)s.c.n_sweeps);      // (step_body has counted this iteration in the workgroup's copy of Ctl)
        if (p0 >= 2 && p0 < 2 + P.n_help) {                        // helper: its la / lb stores are out (the visual roles of the next iteration read them)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
            if (t == 0) st_ag(P.hflag2 + (p0 - 2), epoch);
        }
        if (p0 == 1) {
            // master.  The candidate is out (waited for here); Ctl is still in LDS only (step_body's end_iter leaves it to this tail).  Order: time cap -> the hand-over line
            // of the sweep roles (they read nothing else of Ctl) -> Ctl itself -> word 8 of the line, which the step roles (and nobody on an iteration's longest path) wait for.
            // On paths where the helpers ran no second pass the spare wave's poll inside step_body did not happen: collected here (a helper posts hflag2 on every path).
            if (s.pad0_ != epoch && t < P.n_help) spin_until_eq(P.hflag2 + t, epoch, P.abortf);
            __syncthreads();                                        // (no wait for the candidate's stores: the sweep roles take it from the tagged words, P.xtag)
            if (t == 0) prof_stamp(P, epoch - 1, 24);
            if (t == 0 && !s.c.done && !s.hdr_posted && budget_ticks > 0 && (long long)(wall_clock64() - t_start) > budget_ticks) { s.c.done = 1; s.c.term = 5; if (s.c.iter > 0 && !s.c.resweep) s.c.iter--; }      // (the step just formed was never judged: not an iteration of the summary)
            __syncthreads();
            const bool done = s.c.done != 0;
            auto post_hdr = [&]() { post_iter_header(P, s.c, epoch); };
            if (!done && s.hdr_posted != 1) { post_hdr(); if (t == 0) prof_stamp(P, epoch - 1, 25); }      // the next iteration's sweep roles start from this (unless step_body has posted it with the candidate)
            const bool written = s.c.outd != 0;                    // (cost first: the master wrote the result out inside step_body)
            __syncthreads();
            if (done && s.c.lin_mode == 0 && t == 0) s.c.outd = 1;
            __syncthreads();
            if (t < 64) { const double* src = (const double*)&s.c; double* dst = (double*)P.ctl; for (int i = t; i < (int)(sizeof(Ctl) / 8); i += 64) st_ag(dst + i, src[i]); }
            if (done) {
                if (s.c.lin_mode == 0 && !written) vd::solve_finish<true>(P.x[0], P.x[1], P.xorig, P.hstate, P.ctl, P.hctl, P.hseq, P.K, P.NS, P.gauge_on, s.c.cur, s.c.status, s.c.gen, Alds, &s.c);      // (the host is released here)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
                post_hdr();                                        // everybody reads `done` and leaves
                if (t == 0) __hip_atomic_store(P.ihdr + 8, (unsigned long long)(unsigned)epoch << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads();
            if (t == 0) { __hip_atomic_store(P.ihdr + 8, (unsigned long long)(unsigned)epoch << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); prof_stamp(P, epoch - 1, 26); }
            continue;
        }
        if (t == 0) {
            unsigned long long tw0 = 0; s.red[1] = 0.0;
            for (int sp = 1; (unsigned)(hdr_ld(8) >> 32) != (unsigned)epoch; ++sp) {      // (word 8: Ctl is complete)
                __builtin_amdgcn_s_sleep(8);
                if ((sp & 1023) == 0 && vd::wait_expired(tw0, P.abortf)) { st_ag(P.abortf, 1); s.red[1] = 1.0; break; }
            }
        }
        __syncthreads();
        if (s.red[1] != 0.0) return;
    }
}
#endif
