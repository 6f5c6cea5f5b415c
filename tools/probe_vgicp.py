"""Where an alignment's time goes: one-launch vgicp_align with 0 .. N iterations allowed (16-ring x 1800 pair)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.load_package()
import numpy as np
from mvil_fusion_amd import lib, vgicp
tx, tc, sx, sc, T_true = vgicp.make_pair(seed=20240606, rings=16, az=1800)
v = vgicp.Vgicp(lib.load_vilsolve(), "vgicp_")
v.set_target(tx, None, 0.5); v.set_source(sx, None)
for mi in (0, 1, 2, 3, 64):
    o = v.default_options(max_iterations=mi)
    for _ in range(5): v.align(np.eye(4), o)
    t0 = time.perf_counter(); n = 200
    for _ in range(n): T, s = v.align(np.eye(4), o)
    el = (time.perf_counter() - t0) / n
    print("max_iterations %2d: %.1f us per alignment, %d iterations done" % (mi, 1e6 * el, s.iterations))
