import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, replay, synth, vgicp, mapreg, preint
free0 = torch.cuda.mem_get_info()[0]
# 1. create / destroy churn of every context type
so = lib.load_vilsolve()
for i in range(200):
    be = lib.open_vilsolve(); w = synth.make_config(1); be.solve(w); be.close()
    v = vgicp.Vgicp(so, "vgicp_"); v.close(); m = mapreg.MapReg(so, "vmap_"); m.close(); p = preint.Preint(so, "vpre_"); p.close()
torch.cuda.synchronize()
print("after churn: device memory delta %.1f MB" % ((free0 - torch.cuda.mem_get_info()[0]) / 1e6))
# 2. long replay on one context
be = lib.open_vilsolve()
rp = replay.Replay(K=10, n_frames=2400, L=1000, n_plane=24000, n_edge=6000, seed=7, max_iterations=8)
t0 = time.time(); recs = replay.run(be, rp, n_steps=2300); el = time.time() - t0
err = np.array([r["pos_err_newest"] for r in recs]); tot = np.array([r["solve_ms"] + r["marg_ms"] for r in recs])
print("replay %d frames in %.1f s: median %.3f ms, p99 %.3f ms, max %.3f ms; position error median %.4f max %.4f m; iterations mean %.2f; device memory delta %.1f MB" %
      (len(recs), el, np.median(tot), np.percentile(tot, 99), tot.max(), np.median(err), err.max(), np.mean([r["iterations"] for r in recs]), (free0 - torch.cuda.mem_get_info()[0]) / 1e6))
first, last = np.median(tot[50:250]), np.median(tot[-200:])
print("latency first 200 vs last 200 frames: %.3f vs %.3f ms" % (first, last))
be.close()
