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

#define VIL_SS_DOUBLES ((sizeof(vd::StepShared) + 15) / 16 * 2)      // StepShared at the front of the step roles' dynamic LDS (16-byte granules)
static_assert(VIL_SWEEP_THREADS == VIL_STEP_THREADS, "one block size for every role of k_iter");

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
