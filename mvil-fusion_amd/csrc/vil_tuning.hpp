// Development knobs.  Environment variables that change chunking, pick kernel variants or print debug output exist only in a build
// with -DVIL_TUNING (`make tuning` -> libvilsolve_tuning.so, loaded by tools/ through VIL_LIB); the shipping library contains none of
// them -- tests/test_abi.py greps the binary.  What the test-suite needs to steer is a documented entry point instead
// (vil_debug_set_split, vmap_set_fused_max, vgicp_set_knn_grid); VIL_NO_POLL / VIL_DEBUG are documented run-time switches.
#pragma once
#include <cstdlib>
#ifdef VIL_TUNING
#define VIL_TUNE_ENV(name) getenv(name)
#else
#define VIL_TUNE_ENV(name) ((const char*)nullptr)
#endif
