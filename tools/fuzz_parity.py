"""Seed sweep of the window-level parity: N differently seeded scenes per shape, GPU vs CPU oracle (iterations, termination,
final cost, state after the gauge fix, marginalised information).  python tools/fuzz_parity.py [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
import oracle_lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
be, orc = lib.open_vilsolve(), oracle_lib.open_oracle()
pf = lambda pre: orc.marginalize(pre).to_prior()
shapes = [(1, dict(L=60)), (2, dict(L=100, n_plane=1500, n_edge=500)), (2, dict(K=7, L=80, n_plane=600, n_edge=200)), (4, dict(L=200))]
bad, worst = [], dict(dp=0.0, dc=0.0, dA=0.0)
for cid, kw in shapes:
    for k in range(N):
        a = synth.make_config(cid, prior_fn=pf, seed_offset=1000 * k, **kw); b = synth.make_config(cid, prior_fn=pf, seed_offset=1000 * k, **kw)
        p0 = a.pose[0].copy()
        try:
            sa, sb = be.solve(a), orc.solve(b)
        except lib.VilError as e:
            bad.append((cid, k, "error %s" % e)); continue
        be.gauge_fix(p0, a); orc.gauge_fix(p0, b)
        dp = np.abs(a.pose[:, :3] - b.pose[:, :3]).max(); dc = abs(sa.final_cost - sb.final_cost) / max(sb.final_cost, 1e-300)
        ok = (sa.iterations, sa.termination) == (sb.iterations, sb.termination) and dc < (1e-5 if a.prior.n == 0 else 1e-8) and dp < (1e-4 if a.prior.n == 0 else 1e-6)
        ma, mb = be.marginalize(a, abi.MARGIN_OLD), orc.marginalize(b, abi.MARGIN_OLD)
        dA = np.abs(ma.A_matrix() - mb.A_matrix()).max() / np.abs(mb.A_matrix()).max() if ma.c.n == mb.c.n else 1.0
        ok = ok and dA < (1e-4 if a.prior.n == 0 else 1e-7)
        worst["dp"] = max(worst["dp"], dp if a.prior.n else 0.0); worst["dc"] = max(worst["dc"], dc if a.prior.n else 0.0); worst["dA"] = max(worst["dA"], dA if a.prior.n else 0.0)
        if not ok:
            bad.append((cid, kw.get("K"), k, sa.iterations, sb.iterations, sa.termination, sb.termination, dc, dp, dA))
print("windows: %d, mismatches: %d; worst with a prior: position %.2e m, cost %.2e rel, information %.2e rel" % (len(shapes) * N, len(bad), worst["dp"], worst["dc"], worst["dA"]))
for r in bad[:20]:
    print("  ", r)
