// Marginalisation on device (estimator.cpp:1484-1683, marginalization_factor.cpp:176-316).
#pragma once
#include "../../include/vilsolve.h"
#include "vil_dev.hpp"

struct MargWork { int dummy = 0; };
static inline void marg_free(MargWork&) {}
static inline int marg_run(int, hipStream_t, const DevP&, MargWork&, const vil_problem*, const vil_state*, const SolveOpts&, const vil_marg_spec*, vil_prior_out*) { return VIL_ERR_UNSUPPORTED; }
