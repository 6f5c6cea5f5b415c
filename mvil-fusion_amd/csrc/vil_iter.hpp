// One trust-region iteration in ONE launch (single GPU, chain windows): the factor sweep, the gather of its records, the elimination of the speed-bias chain
// and the step (judge, dense solve, dogleg, candidate) as roles of a single grid
//   [ sweep roles: imu x n_imu | prior | rel | visual x n_vwg | plane | edge ]  [ chain | master | helpers x n_help | W W^T tiles x n_ww | gather x n_gather ]
// instead of k_sweep followed by the merged gather + step launch (k_step, rs_merged).  What the second launch could not overlap is what this one buys:
//   * the chain workgroup starts behind the IMU / prior workgroups' flags (~5 us into the launch) and eliminates the chain UNDER the visual workgroups and
//     the gather -- in the two-launch structure it was one of the two ~20 us legs the master waited for before its dense factorisation;
//   * the gather workgroups are resident and staged when the last visual record lands (no launch ramp between them);
//   * one launch boundary per iteration instead of two.
// The sweep roles and the chain workgroup wait for LOWER block indices only (the hardware dispatches in index order: what a resident workgroup waits for is
// resident or done).  Master, helpers and tile workgroups wait for one another and for the gather workgroups BEHIND them in the grid: they are few, resident from
// the start (their prologues run under the sweep), and the launch is only taken when the device holds all of them plus two more workgroups at once
// (vil_coop.hpp), so the gather workgroups always find compute units to run through.  Everything that crosses workgroups inside
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
    if ((int)blockIdx.x < P.n_sw) {
        const Ctl ctl = *P.ctl;
        if (ctl.done) return;
        sweep_body<TS, true>(P, O, ctl, dyn, (int)blockIdx.x);
        return;
    }
    vd::StepShared& s = *reinterpret_cast<vd::StepShared*>(dyn);
    step_body<true, 3, true>(P, O, s, dyn + VIL_SS_DOUBLES, (int)blockIdx.x - P.n_sw);
}

// ---- The whole SOLVE in one launch -----------------------------------------------------------------------------------------------------------------------------
// What still separated two one-launch iterations was the launch boundary itself: ~12 us per iteration at configs[1] between the master's last store and the first
// workgroup of the next launch (kernel teardown, the dispatch of 400 workgroups with 100 kB of LDS each, the graph's node-to-node hand-off) -- a sixth of the
// iteration.  k_solve keeps every workgroup RESIDENT for the whole solve:
//   [ chain | master | helpers x n_help | W W^T tiles x n_ww ]   dedicated workgroups, the roles of k_iter, one iteration after the other
//   [ workers ]                                                     each takes sweep role `w` (then tickets from Ctl-indexed counters while there are more roles than
//                                                                   workers), then gather item `w` (likewise)
// and the master ends iteration n by posting the epoch of iteration n + 1 in P.goflag -- behind its own stores, the helpers' la / lb (hdone) and Ctl.  Every flag of
// the iteration carries that epoch, so nothing is ever reset; everything that crosses workgroups is stored and loaded at agent scope, now including what used to
// cross a launch boundary (Ctl, the candidate state, la / lb, the Jacobi scales).  Taken when the device holds the dedicated workgroups plus a worker for every
// sweep role or gather item of the longer phase... or fewer: a worker then walks several (vilsolve.hip).  Every wait is bounded (spin_until_eq): a launch that cannot
// finish raises P.abortf and ends; the host reports VIL_ERR_DEVICE.
template <int TS>
__global__ __launch_bounds__(VIL_STEP_THREADS) void k_solve(DevP P, SolveOpts O) {
    extern __shared__ double dyn[];
    const int t = threadIdx.x, b = (int)blockIdx.x;
    const int nded = 2 + P.n_help + P.n_ww, nwk = (int)gridDim.x - nded, w = b - nded;
    double* const tail = dyn + P.tail_off;                  // [Ctl head 32 | camera part of the candidate 336 | ticket]
    double* const xs = tail + 32; int* const tick = (int*)(tail + 368);
    const int gen = vd::ld_ag(&P.ctl->gen);                 // (written by the init launch)
    constexpr int HEAD = (int)(offsetof(Ctl, cost_trace) / 8);
    static_assert(offsetof(Ctl, cost_trace) % 8 == 0 && HEAD <= 32, "the workers copy the head of Ctl as doubles");
    for (int n = 0; n <= O.max_iterations + 24; ++n) {
        const int epoch = (int)((((unsigned)gen) << 12) + (unsigned)n + 1u);
        if (t == 0) { vd::spin_until_eq(P.goflag, epoch, P.abortf); *tick = vd::ld_ag(P.abortf); }
        __syncthreads();
        if (*tick != 0) return;                               // somebody gave up waiting: the launch ends (uniform: one thread's reading)
        __syncthreads();
        if (b < nded) {
            vd::StepShared& s = *reinterpret_cast<vd::StepShared*>(dyn);
            step_body<true, 3, true>(P, O, s, dyn + VIL_SS_DOUBLES, b);
            __syncthreads();
            if (s.done_at_entry) return;
            continue;
        }
        if (t < HEAD) tail[t] = vd::ld_ag((const double*)P.ctl + t);
        __syncthreads();
        Ctl ctl;
        { double* cd = (double*)&ctl; for (int i = 0; i < HEAD; ++i) cd[i] = tail[i]; }
        if (ctl.done) return;
        {   // the candidate's camera part (written by the master of the previous iteration) into LDS: the factor roles read poses, speeds / biases, extrinsic and td from there
            const double* xg = P.x[1 - ctl.cur];
            for (int i = t; i < 16 * P.K + 8; i += VIL_STEP_THREADS) xs[i] = vd::ld_ag(xg + i);
        }
        __syncthreads();
        const bool qs = P.n_sw > nwk, qg = P.n_gather > nwk;      // more roles than workers: tickets (uniform)
        for (int item = w; item < P.n_sw;) {
            sweep_body<TS, true>(P, O, ctl, dyn, item, xs);
            if (!qs) break;
            __syncthreads();
            if (t == 0) *tick = nwk + atomicAdd(P.qsweep + (n & 63), 1);
            __syncthreads();
            item = *tick;
        }
        __syncthreads();
        for (int item = w; item < P.n_gather;) {
            reduce_gather<true, VIL_STEP_THREADS / 8, true>(P, ctl, item, (int4*)dyn, epoch);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) { vd::st_ag(P.gflag + item, epoch); prof_stamp(P, epoch - 1, 5); }
            if (!qg) break;
            if (t == 0) *tick = nwk + atomicAdd(P.qgather + (n & 63), 1);
            __syncthreads();
            item = *tick;
        }
        __syncthreads();
    }
}
