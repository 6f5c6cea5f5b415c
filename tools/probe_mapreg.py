"""Where a scan-to-map registration spends its time on the GPU box (wall clock per phase)."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, mapreg
from mvil_fusion_amd.vgicp import _rot

cm, sm = mapreg.make_map(seed=20240607, n_surf=60000, n_corner=8000)
R, t = _rot(0.01, -0.015, 0.5), np.array([1.5, -1.0, 0.25])
sc, ss = mapreg.make_scan(cm, sm, R, t, seed=11, n_surf=6000, n_corner=800)
q0 = mapreg.quat_from_R(R @ _rot(0.004, -0.003, 0.008)); t0 = t + np.array([0.05, -0.04, 0.03])
so = lib.load_vilsolve(); be = lib.open_vilsolve(); m = mapreg.MapReg(so, "vmap_")
for i in range(4):
    a = time.perf_counter(); m.set_map(cm, sm); print("set_map %d: %.3f ms" % (i, 1e3 * (time.perf_counter() - a)))
for i in range(3):
    m.align(be.ctx, sc, ss, q0, t0)
N = 50
a = time.perf_counter()
for i in range(N): m.align(be.ctx, sc, ss, q0, t0)
print("align: %.3f ms" % (1e3 * (time.perf_counter() - a) / N))
a = time.perf_counter()
for i in range(N): e, p = m.associate(sc, ss, q0, t0)
print("associate: %.3f ms" % (1e3 * (time.perf_counter() - a) / N))
q, tt, sm_ = m.align(be.ctx, sc, ss, q0, t0)
print("inside align: associate %.3f ms, prepare %.3f ms, solve %.3f ms, iterations(last) %d" % (sm_.t_associate_ms, sm_.t_prepare_ms, sm_.t_solve_ms, sm_.iterations))
