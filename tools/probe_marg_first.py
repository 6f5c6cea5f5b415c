"""First vil_marginalize of a context vs the later ones (9 ms since round 3).  Under rocprofv3 --hip-trace --stats the HIP calls of the first one show up."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
be = lib.open_vilsolve()
w = synth.make_config(1)
be.solve(w)                                    # code objects loaded, the solve's own allocations done
for k in range(4):
    t0 = time.perf_counter(); be.marginalize(w); print("marginalize call %d: %.3f ms" % (k, 1e3 * (time.perf_counter() - t0)), flush=True)
be2 = lib.open_vilsolve(); w2 = synth.make_config(2); be2.solve(w2)
for k in range(3):
    t0 = time.perf_counter(); be2.marginalize(w2); print("second context, K = 10 window, call %d: %.3f ms" % (k, 1e3 * (time.perf_counter() - t0)), flush=True)
