"""What the launch-structure ladder does when the device is smaller than the occupancy query says.  Run under a CU mask:
    HSA_CU_MASK=0:0-31 python tools/cu_mask_check.py        (32 compute units; ROC_GLOBAL_CU_MASK is the hex form)
Per BASELINE window: the structure the upload chose, then 12 solves -- status, agreement with the CPU oracle, time per solve, how many were re-run one rung down."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__ as g; g.load_package()
import numpy as np
from mvil_fusion_amd import abi, lib, synth
import oracle_lib
orc = oracle_lib.open_oracle()
import torch
print("mask: HSA_CU_MASK=%s ROC_GLOBAL_CU_MASK=%s; multiprocessor count reported: %d" % (os.environ.get("HSA_CU_MASK"), os.environ.get("ROC_GLOBAL_CU_MASK"), torch.cuda.get_device_properties(0).multi_processor_count), flush=True)
for cfg in (2, 3, 4):
    w = synth.make_config(cfg, prior_fn=lambda pre: orc.marginalize(pre).to_prior())
    wo = synth.make_config(cfg, prior_fn=lambda pre: orc.marginalize(pre).to_prior()); so = orc.solve(wo)
    be = lib.open_vilsolve(); be.upload(w)
    n, one = C.c_int32(-1), C.c_int32(0); be.lib.vil_debug_get_launch_structure(be.ctx, C.byref(n), C.byref(one))
    times, ok = [], True
    for k in range(12):
        be.reset_state(); t0 = time.perf_counter()
        try:
            s = be.solve_resident()
        except lib.VilError as e:
            print("  cfg %d solve %d: ERROR status %d" % (cfg, k, e.status), flush=True); ok = False; break
        times.append(1e3 * (time.perf_counter() - t0))
        ok = ok and (s.iterations, s.termination) == (so.iterations, so.termination) and abs(s.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    rec, fail = C.c_int64(0), C.c_int64(0); be.lib.vil_recovery_counts(be.ctx, C.byref(rec), C.byref(fail))
    print("cfg %d: launches per iteration chosen at upload %d (0 = persistent solve); agrees with the oracle: %s; ms per solve: %s; re-run one rung down: %d, failed: %d"
          % (cfg, n.value, ok, " ".join("%.2f" % t for t in times), rec.value, fail.value), flush=True)
    be.close()
