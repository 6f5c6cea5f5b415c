"""Bitwise run-to-run reproducibility under stress: the same window solved N times through one upload (graph replay) and through fresh uploads; reports mismatches per launch mode."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as g; g.load_package()
from mvil_fusion_amd import abi, lib, synth
N = int(os.environ.get("N", "200"))
for mode in [int(v) for v in os.environ.get("MODES", "0,3").split(",")]:
    for cid, kw in ((1, {}), (2, dict(L=120, n_plane=1600, n_edge=480)), (2, {})):
        be = lib.open_vilsolve()
        if mode: be.lib.vil_debug_set_launch_mode(be.ctx, mode)
        ref = None; bad = 0; bad_cost = 0
        for i in range(N):
            w = synth.make_config(cid, **kw)
            s = be.solve(w, abi.default_options(max_iterations=12))        # fresh upload each time (direct launches, generous first chunk)
            key = (w.pose.tobytes(), w.inv_depth.tobytes(), w.speedbias.tobytes(), s.final_cost, s.iterations)
            if ref is None: ref = key
            elif key != ref:
                bad += 1; bad_cost += key[3] != ref[3]
        w = synth.make_config(cid, **kw); be.upload(w); ref2 = None; bad2 = 0
        for i in range(N):
            be.reset_state(); s = be.solve_resident(abi.default_options(max_iterations=12)); be.download_state(w)
            key = (w.pose.tobytes(), w.inv_depth.tobytes(), s.final_cost, s.iterations)
            if ref2 is None: ref2 = key
            elif key != ref2: bad2 += 1
        print("mode %d cfg %d %s: fresh uploads %d/%d mismatching (cost differs in %d), resident re-solves %d/%d" % (mode, cid, kw, bad, N, bad_cost, bad2, N), flush=True)
        be.close()
