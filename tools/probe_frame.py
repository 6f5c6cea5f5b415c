"""Per-frame phase timing of vil_solve on the synthetic replay (prepare = pack + upload + setup, solve, readback)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, replay
be = lib.open_vilsolve()
rp = replay.Replay(K=10, n_frames=50, L=1000, n_plane=24000, n_edge=6000, seed=20240605, max_iterations=8)
rows = []
for step in range(30):
    w = rp.window(); flag = rp.margin_flag(); p0 = w.pose[0].copy()
    t0 = time.perf_counter(); p, s = w.c_problem(), w.c_state(); t1 = time.perf_counter()
    sg = be.solve(w, rp.opts); t2 = time.perf_counter()
    be.gauge_fix(p0, w)
    t3 = time.perf_counter(); pg = be.marginalize(w, flag, w._icp_marg, w._lps_marg, rp.opts); t4 = time.perf_counter()
    rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), sg.t_prepare_ms, sg.t_solve_ms, sg.t_readback_ms, sg.iterations, 1e3 * (t4 - t3)))
    rp.absorb(w, pg, flag)
a = np.array(rows[5:])
print("median ms: ctypes struct build %.3f | be.solve total %.3f = prepare %.3f + solve %.3f (%.1f it) + readback %.3f | marginalize %.3f" % tuple(np.median(a, axis=0)[[0, 1, 2, 3, 5, 4, 6]]))
