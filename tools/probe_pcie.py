"""Where a PCIe-inclusive vil_solve of the configs[1] window spends its time: Python harness, upload phases (tuning build,
VIL_UPLOAD_TRACE=1), iterate, read-back.  Run with VIL_LIB=mvil-fusion_amd/csrc/libvilsolve_tuning.so."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
be = lib.open_vilsolve()
w = synth.make_config(2, prior_fn=lambda pre: be.marginalize(pre).to_prior())
opts = abi.default_options()
saved = w.state_copy()
rows = []
for it in range(12):
    t0 = time.perf_counter(); w.set_state(saved); t1 = time.perf_counter()
    p, s = w.c_problem(), w.c_state(); t2 = time.perf_counter()
    if it == 11: os.environ["VIL_UPLOAD_TRACE"] = "1"
    sg = be.solve(w, opts); t3 = time.perf_counter()
    rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), sg.t_prepare_ms, sg.t_solve_ms, sg.t_readback_ms, sg.iterations))
a = np.median(np.array(rows[3:]), axis=0)
print("median ms: set_state %.3f | struct build (done twice per be.solve) %.3f | be.solve %.3f = prepare %.3f + iterate %.3f + readback %.3f (%d it)" % tuple(a))
