"""Per-image time split of the resident replay: library-internal phases (vil_summary) vs the Python harness around them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, replay
be = lib.open_vilsolve()
rp = replay.Replay(K=10, n_frames=130, L=1000, n_plane=24000, n_edge=6000, seed=20240605, max_iterations=8)
K = rp.K
be.set_gauge_fix(True); be.lidar_reset()
for k in range(K): be.lidar_push(rp.lidar[k][0], rp.lidar[k][1])
rows = []
for step in range(120):
    flag = rp.margin_flag()
    w = rp.window(with_lidar=False)
    t0 = time.perf_counter(); sg = be.solve(w, rp.opts); t1 = time.perf_counter()
    pg = be.marginalize_resident(w, flag, w._icp_marg, w._lps_marg, rp.opts); t2 = time.perf_counter()
    rows.append((1e3 * (t1 - t0), sg.t_prepare_ms, sg.t_solve_ms, sg.t_readback_ms, 1e3 * (t2 - t1), sg.iterations))
    be.lidar_drop(0 if flag == abi.MARGIN_OLD else K - 2)
    if not rp.absorb(w, pg, flag): break
    be.lidar_push(rp.lidar[K - 1][0], rp.lidar[K - 1][1])
r = np.median(np.array(rows[10:]), axis=0)
print("median per image: solve call %.3f ms = prepare %.3f + iterate %.3f + readback %.3f + harness %.3f | marginalize_resident %.3f ms | iterations %.1f" % (r[0], r[1], r[2], r[3], r[0] - r[1] - r[2] - r[3], r[4], r[5]))
