"""Seed sweep of the widened rows' parity (GPU vs CPU oracle): scan-to-scan VGICP, scan-to-map registration, IMU pre-integration.
python tools/fuzz_rows.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, vgicp, mapreg, preint
from mvil_fusion_amd.vgicp import _rot
import oracle_lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
so, orc, be = lib.load_vilsolve(), oracle_lib.open_oracle(), lib.open_vilsolve()
# ---- VGICP: alignment with device-estimated covariances, the three neighbour modes in turn
G, O = vgicp.Vgicp(so, "vgicp_"), vgicp.Vgicp(orc.lib, "orc_vgicp_")
bad, worst = 0, 0.0
for k in range(N):
    tx, tc, sx, sc, Tt = vgicp.make_pair(seed=100 + k, rings=6 + k % 5, az=300 + 37 * (k % 7))
    mode = (vgicp.DIRECT1, vgicp.DIRECT7, vgicp.DIRECT27)[k % 3]
    for r in (G, O):
        r.set_target(tx, None, 0.5); r.set_source(sx, None)
    Tg, sg = G.align(np.eye(4), G.default_options(neighbor_mode=mode, optimizer=vgicp.LM if k % 2 else vgicp.GN)); To, s_o = O.align(np.eye(4), O.default_options(neighbor_mode=mode, optimizer=vgicp.LM if k % 2 else vgicp.GN))
    d = np.abs(Tg - To).max(); worst = max(worst, d)
    bad += not (sg.iterations == s_o.iterations and sg.converged == s_o.converged and d < 1e-8)
print("vgicp: %d pairs, mismatches %d, worst |dT| %.2e" % (N, bad, worst))
# ---- scan-to-map
Gm, Om = mapreg.MapReg(so, "vmap_"), mapreg.MapReg(orc.lib, "orc_vmap_")
bad, worst = 0, 0.0
for k in range(N):
    cm, sm = mapreg.make_map(seed=200 + k, n_surf=4000 + 500 * (k % 9), n_corner=600 + 100 * (k % 5))
    R, t = _rot(0.01 * (k % 3), -0.01 * (k % 4), 0.1 * k), np.array([0.3 * (k % 7) - 1, 0.2 * (k % 5) - 0.5, 0.1])
    sc, ss = mapreg.make_scan(cm, sm, R, t, seed=300 + k, n_surf=700 + 50 * (k % 6), n_corner=150 + 10 * (k % 4))
    q0 = mapreg.quat_from_R(R @ _rot(0.004, -0.003, 0.008)); t0 = t + np.array([0.05, -0.04, 0.03])
    for r in (Gm, Om):
        r.set_map(cm, sm)
    qg, tg, sg = Gm.align(be.ctx, sc, ss, q0, t0); qo, to, s_o = Om.align(None, sc, ss, q0, t0)
    d = max(np.abs(tg - to).max(), np.abs(qg - qo).max()); worst = max(worst, d)
    bad += not ((sg.n_edge, sg.n_plane, sg.iterations) == (s_o.n_edge, s_o.n_plane, s_o.iterations) and d < 1e-8)
print("mapreg: %d scans, mismatches %d, worst |dpose| %.2e" % (N, bad, worst))
# ---- pre-integration
Gp, Op = preint.Preint(so, "vpre_"), preint.Preint(orc.lib, "orc_vpre_")
worst = 0.0
for k in range(N):
    s = preint.make_stream(n_intervals=1 + k % 19, samples=(1 + k % 5, 10 + 13 * (k % 11)), seed=400 + k)
    rg, jg = Gp.integrate(*s); ro, jo = Op.integrate(*s)
    worst = max(worst, np.abs(rg[:, 62:] - ro[:, 62:]).max() / np.abs(ro[:, 62:]).max(), np.abs(jg - jo).max() / np.abs(jo).max(), np.abs(rg[:, :17] - ro[:, :17]).max())
print("preint: %d streams, worst relative difference %.2e" % (N, worst))
