"""Full-size BASELINE configs 1-4: GPU vs oracle parity + timing (not a bench line; see bench.py for configs[1])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
import oracle_lib
orc = oracle_lib.open_oracle(); be = lib.open_vilsolve()
for cid in (1, 2, 3, 4):
    pf = lambda pre: be.marginalize(pre).to_prior()
    wg = synth.make_config(cid, prior_fn=pf); wo = synth.make_config(cid, prior_fn=pf)
    p0 = wg.pose[0].copy()
    be.upload(wg)
    for _ in range(2): be.reset_state(); be.solve_resident()
    t0 = time.perf_counter(); n = 10
    for _ in range(n): be.reset_state(); sg = be.solve_resident()
    tg = (time.perf_counter() - t0) / n
    be.download_state(wg)
    t0 = time.perf_counter(); so = orc.solve(wo); to = time.perf_counter() - t0
    be.gauge_fix(p0, wg); orc.gauge_fix(p0, wo)
    dp = np.abs(wg.pose[:, :3] - wo.pose[:, :3]).max(); dq = np.abs(wg.pose[:, 3:] - wo.pose[:, 3:]).max()
    t0 = time.perf_counter(); mg = be.marginalize(wg); tm = time.perf_counter() - t0
    t0 = time.perf_counter(); mo = orc.marginalize(wo); tmo = time.perf_counter() - t0
    print("config %d K=%d L=%d Fv=%d Np=%d Ne=%d prior n=%d | GPU %d it %.3f ms (%.0f it/s) | CPU %d it %.1f ms (%.0f it/s) | speedup %.1fx | cost %.6f vs %.6f | dpos %.2e dquat %.2e | marg GPU %.2f ms CPU %.2f ms dA %.1e"
          % (cid, wg.K, wg.L, len(wg.vis_i), len(wg.plane_pose), len(wg.edge_pose), wg.prior.n, sg.iterations, 1e3 * tg, sg.iterations / tg, so.iterations, 1e3 * to, so.iterations / to,
             (sg.iterations / tg) / (so.iterations / to), sg.final_cost, so.final_cost, dp, dq, 1e3 * tm, 1e3 * tmo, np.abs(mg.A_matrix() - mo.A_matrix()).max() / np.abs(mo.A_matrix()).max()), flush=True)
