"""Phases of the marginalisation kernels from their own clock stamps (vil_debug_marg_stamps), K = 10 (n = 70) and K = 20 (n = 130); first call of the process vs warm."""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g; g.load_package()
import numpy as np
from mvil_fusion_amd import abi, lib, synth
NAMES = ["k_marg entered", "dropped block gathered", "15x15 Cholesky inverse", "kept x dropped blocks staged", "T = A_kd A_dd^-1", "A = A_kk - T A_dk, b", "symmetrised copies out",
         "k_marg_fast entered", "tiles loaded", "n x n factorisation", "J0 / r0 out", "k_marg phase 1 entered", "pivoted square root done"]
for cfg, kw in ((2, {}), (4, {}), (4, dict(n_plane=24000, n_edge=6000))):
    be = lib.open_vilsolve()
    w = synth.make_config(cfg, prior_fn=lambda pre: be.marginalize(pre).to_prior(), **kw)
    be.solve(w)
    ts = []
    for k in range(6):
        t0 = time.perf_counter(); out = be.marginalize(w); ts.append(1e3 * (time.perf_counter() - t0))
    st = (C.c_uint64 * 16)(); assert be.lib.vil_debug_marg_stamps(be.ctx, st) == 0
    v = [int(x) for x in st]
    print("config %d %s: K = %d, kept n = %d; vil_marginalize host ms: %s" % (cfg, kw, w.K, out.c.n, " ".join("%.3f" % t for t in ts)))
    prev = v[0]
    for k in range(1, 13):
        if v[k] == 0 or v[k] < prev: continue
        print("   %-32s +%7.2f us  (at %7.2f)" % (NAMES[k], (v[k] - prev) * 0.01, (v[k] - v[0]) * 0.01)); prev = v[k]
    be.close()
