import sys, os, time, ctypes as C
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, mapreg
from mvil_fusion_amd.vgicp import _rot
cm, sm = mapreg.make_map(seed=20240607, n_surf=60000, n_corner=8000)
R, t = _rot(0.01, -0.015, 0.5), np.array([1.5, -1.0, 0.25])
sc, ss = mapreg.make_scan(cm, sm, R, t, seed=11, n_surf=6000, n_corner=800)
q0 = mapreg.quat_from_R(R @ _rot(0.004, -0.003, 0.008)); t0 = t + np.array([0.05, -0.04, 0.03])
so = lib.load_vilsolve(); be = lib.open_vilsolve(); m = mapreg.MapReg(so, "vmap_"); m.set_map(cm, sm)
def morton(p, h=1.0):
    k = np.floor(p[:, :3] / h).astype(np.int64) + 512
    key = np.zeros(len(p), dtype=np.int64)
    for b in range(10):
        for a in range(3): key |= ((k[:, a] >> b) & 1) << (3 * b + a)
    return np.argsort(key, kind="stable")
for name, (c, s) in {"as generated": (sc, ss), "morton sorted": (sc[morton(sc)], ss[morton(ss)]), "shuffled": (sc[np.random.default_rng(1).permutation(len(sc))], ss[np.random.default_rng(2).permutation(len(ss))])}.items():
    c = np.ascontiguousarray(c); s = np.ascontiguousarray(s)
    for _ in range(5): m.align(be.ctx, c, s, q0, t0)
    m.lib.vmap_profile_enable(m.ctx, 1)
    a = time.perf_counter()
    for _ in range(200): m.align(be.ctx, c, s, q0, t0)
    el = (time.perf_counter() - a) / 200
    pn, pms = (C.c_int64 * 2)(), (C.c_double * 2)(); m.lib.vmap_profile_read(m.ctx, pn, pms); m.lib.vmap_profile_enable(m.ctx, 0)
    print("%-14s align %.1f us  search %.2f us  fit %.2f us" % (name, 1e6 * el, 1e3 * pms[0] / max(1, pn[0]), 1e3 * pms[1] / max(1, pn[1])))
