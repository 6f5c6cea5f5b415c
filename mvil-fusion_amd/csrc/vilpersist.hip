// The persistent solve kernel (k_solve, vil_iter.hpp) as a translation unit of its own: the roles of the one-launch iteration are compiled a second time here
// with an OPAQUE thread index (VIL_OPAQUE_TID, vil_math.hpp) -- inlined into k_solve's iteration loop they would otherwise have every per-thread address hoisted
// in front of the loop and spilled.  vilsolve.hip launches the kernel through the three functions at the end; everything else of the library stays in vilsolve.hip.
#include <hip/hip_runtime.h>

#define VIL_OPAQUE_TID 1
#define VIL_PERSIST_TU 1
#include "vil_tuning.hpp"
#include "vil_dev.hpp"
#include "vil_finish.hpp"
#include "vil_sweep.hpp"
#include "vil_step.hpp"
#include "vil_iter.hpp"
#include "vil_internal.h"

const void* vil_k_solve_fn(int vis_ts) { return vis_ts == 2 ? (const void*)k_solve<2> : (const void*)k_solve<5>; }
size_t vil_k_solve_ss_bytes() { return 8 * (size_t)VIL_SS_DOUBLES; }
void vil_k_solve_launch(int vis_ts, unsigned grid, size_t lds, hipStream_t stream, const DevP& P, const SolveOpts& O, long long budget_ticks) {
    KSolveArgs A; A.P = P; A.O = O; A.budget_ticks = budget_ticks;
    if (vis_ts == 2) hipLaunchKernelGGL(k_solve<2>, dim3(grid), dim3(VIL_STEP_THREADS), lds, stream, A);
    else hipLaunchKernelGGL(k_solve<5>, dim3(grid), dim3(VIL_STEP_THREADS), lds, stream, A);
}
