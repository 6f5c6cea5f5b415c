"""A/B of step-kernel variants on ONE box (boxes of the pool differ by several percent): it/s of repeated resident solves, no events.
Run with VIL_LIB=.../libvilsolve_tuning.so and the knob under test (VIL_NO_MERGE, VIL_NO_PRECHAIN, VIL_GATHER32, ...)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
cfg = int(os.environ.get("CFG", "2"))
be = lib.open_vilsolve()
w = synth.make_config(cfg, prior_fn=lambda pre: be.marginalize(pre).to_prior())
be.upload(w)
opts = abi.default_options()
for _ in range(5): be.reset_state(); be.solve_resident(opts)
best = 0.0
for rep in range(5):
    t0 = time.perf_counter(); its = 0
    for _ in range(20): be.reset_state(); s = be.solve_resident(opts); its += s.iterations
    best = max(best, its / (time.perf_counter() - t0))
print("cfg %d %-40s best of 5 x 20 solves: %.0f it/s (%d iterations, final cost %.6f)" % (cfg, " ".join(k for k in os.environ if k.startswith("VIL_") and k != "VIL_LIB") or "default", best, s.iterations, s.final_cost))
