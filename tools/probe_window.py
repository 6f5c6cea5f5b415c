"""Per-image time split of the fully resident replay (vil_win_*): library-internal phases (vil_summary) vs the Python harness around them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, replay
be = lib.open_vilsolve()
rp = replay.Replay(K=10, n_frames=130, L=1000, n_plane=24000, n_edge=6000, seed=20240605, max_iterations=8)
K = rp.K
be.set_gauge_fix(True); be.win_open(**rp.win_open_args())
for k in range(K): be.win_push_frame(rp.win_frame(k))
rows = []
for step in range(110):
    flag = rp.margin_flag()
    w = rp.win_window()
    if step == 100: os.environ["VIL_UPLOAD_TRACE"] = "1"
    t0 = time.perf_counter(); sg = be.win_solve(w, rp.opts); t1 = time.perf_counter()
    os.environ.pop("VIL_UPLOAD_TRACE", None)
    be.win_marginalize(flag, w._icp_marg, w._lps_marg, rp.opts); t2 = time.perf_counter()
    be.win_drop_frame(flag); t3 = time.perf_counter()
    if not rp.absorb(w, None, flag): break
    fr = rp.win_frame(K - 1)
    t4 = time.perf_counter(); be.win_push_frame(fr); t5 = time.perf_counter()
    rows.append((1e3 * (t1 - t0), sg.t_prepare_ms, sg.t_solve_ms, sg.t_readback_ms, 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t5 - t4), sg.iterations))
r = np.median(np.array(rows[10:]), axis=0)
print("median per image: win_solve call %.3f ms = prepare %.3f + iterate %.3f + readback %.3f + harness %.3f | marginalize call %.3f | drop %.3f | push %.3f | iterations %.1f" % (r[0], r[1], r[2], r[3], r[0] - r[1] - r[2] - r[3], r[4], r[5], r[6], r[7]))
