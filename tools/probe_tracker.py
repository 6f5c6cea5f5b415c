"""Host-side breakdown of the tracker chain (vil_win_*): per image push_frame / solve / marginalize / drop, and (under rocprofv3 --kernel-trace) the kernels of one image."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, replay
be = lib.open_vilsolve()
N = int(os.environ.get("N", "60"))
rp = replay.Replay(K=10, n_frames=N + 24, L=1000, n_plane=24000, n_edge=6000, seed=20240605, max_iterations=30, max_time_s=0.05)
K = rp.K
be.set_gauge_fix(True); be.win_open(**rp.win_open_args())
for k in range(K): be.win_push_frame(rp.win_frame(k))
T = {k: [] for k in ("solve", "marg_call", "drop", "push", "its", "prepare", "t_solve", "readback")}
for step in range(N):
    flag = rp.margin_flag(); w = rp.win_window()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); sg = be.win_solve(w, rp.opts); t1 = time.perf_counter()
    be.win_marginalize(flag, w._icp_marg, w._lps_marg, rp.opts); t2 = time.perf_counter()
    torch.cuda.synchronize(); t2b = time.perf_counter()
    be.win_drop_frame(flag); t3 = time.perf_counter()
    assert rp.absorb(w, None, flag)
    fr = rp.win_frame(K - 1)
    t4 = time.perf_counter(); be.win_push_frame(fr); t5 = time.perf_counter()
    if step >= 8:
        T["solve"].append(1e6 * (t1 - t0)); T["marg_call"].append(1e6 * (t2 - t1)); T["drop"].append(1e6 * (t3 - t2b)); T["push"].append(1e6 * (t5 - t4)); T["its"].append(sg.iterations)
        T.setdefault("marg_wait", []).append(1e6 * (t2b - t2))
        T["prepare"].append(1e3 * sg.t_prepare_ms); T["t_solve"].append(1e3 * sg.t_solve_ms); T["readback"].append(1e3 * sg.t_readback_ms)
print({k: round(float(np.median(v)), 1) for k, v in T.items()}, "its/s timed like bench: %.0f" % (sum(T["its"]) / (1e-6 * (sum(T["solve"]) + sum(T["drop"]) + sum(T["push"])))))
print("solve us per iteration: %.1f ; fixed per solve (intercept of a line through (iterations, solve us)): %s" % (np.median(np.array(T["solve"]) / np.array(T["its"])), np.polyfit(T["its"], T["solve"], 1).round(1).tolist()))
be.close()
