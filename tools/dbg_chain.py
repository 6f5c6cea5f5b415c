"""Debug aid: first trust-region iterations of small windows, chain step kernel vs dense step kernel (VIL_DENSE_STEP=1) vs oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
from mvil_fusion_amd.abi import Window
import oracle_lib
orc = oracle_lib.open_oracle()
be = lib.open_vilsolve()

def k2():
    base = synth.make_config(1)
    w = Window(2, 12)
    w.pose, w.speedbias = base.pose[:2].copy(), base.speedbias[:2].copy()
    w.ex_pose, w.td = base.ex_pose.copy(), base.td.copy()
    w.G, w.sqrt_info_px = base.G.copy(), base.sqrt_info_px
    w.imu_i, w.imu_j, w.imu_const = np.array([0], np.int32), np.array([1], np.int32), base.imu_const[:1].copy()
    sel = np.where((base.vis_i == 0) & (base.vis_j == 1))[0][:12]
    w.vis_i, w.vis_j, w.vis_l = base.vis_i[sel].copy(), base.vis_j[sel].copy(), np.arange(len(sel), dtype=np.int32)
    w.vis_const = base.vis_const[sel].copy()
    w.L = len(sel); w.inv_depth = base.inv_depth[base.vis_l[sel]].copy(); w.lm_const = np.zeros(w.L, np.uint8)
    return w

cases = [("K2", k2)] + [("K%d" % K, (lambda K=K: synth.make_config(1, K=K, L=40))) for K in (4, 5, 6, 7)]
for name, mk in cases:
    for it in (1, 3):
        wg, wo = mk(), mk()
        opts = abi.default_options(max_iterations=it)
        try:
            sg = be.solve(wg, opts)
        except Exception as e:
            print(name, it, "GPU error", e); continue
        so = orc.solve(wo, opts)
        print(name, "it", it, "cost %.9g vs %.9g" % (sg.final_cost, so.final_cost), "dpose %.2e dsb %.2e dlam %.2e" % (
            np.abs(wg.pose - wo.pose).max(), np.abs(wg.speedbias - wo.speedbias).max(), np.abs(wg.inv_depth - wo.inv_depth).max()),
            "sb diff per frame", np.abs(wg.speedbias - wo.speedbias).max(axis=1).round(10).tolist())
