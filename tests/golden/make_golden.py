#!/usr/bin/env python3
"""Generates the golden fixtures tests/golden/*.npz.

The reference ships no golden vectors and cannot be compiled or imported here (C++ needing
Eigen/Ceres/ROS -- SURVEY.md 8c), so the fixtures are produced by the CPU restatement in oracle/
(itself pinned by the finite-difference / identity tests) on small deterministic windows.  Each file
holds the INPUT window (every table of vil_problem / vil_state) and the EXPECTED outputs:
per-class residuals and Jacobians, the Schur-reduced normal equations, the solve summary and the
solved + gauge-fixed state, the marginalisation result.   Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

g.load_package()
from mvil_fusion_amd import abi, synth  # noqa: E402
import oracle_lib  # noqa: E402

CASES = {
    "c1_mini": (1, dict(L=24)),                                                  # IMU + visual only, no prior (config 1 shape)
    "c2_mini": (2, dict(L=40, n_plane=160, n_edge=60)),                          # VIL window with prior, ICP, LPS, LiDAR points
    "c4_mini": (4, dict(L=36)),                                                  # K = 20 window, prior n = 130
}


def main():
    orc = oracle_lib.open_oracle()
    pf = lambda pre: orc.marginalize(pre).to_prior()
    for name, (cid, kw) in CASES.items():
        w = synth.make_config(cid, prior_fn=pf, **kw)
        out = {"in_" + k: v for k, v in w.to_dict().items()}
        for cls in range(7):
            r, J = orc.eval_factors(w, cls)
            out["r%d" % cls], out["J%d" % cls] = r, J
        cost, S, gv = orc.linearize(w)
        out["lin_cost"], out["lin_S"], out["lin_g"] = np.array([cost]), S, gv
        w2 = abi.Window.from_dict(w.to_dict())
        p0 = w2.pose[0].copy()
        sm = orc.solve(w2)
        orc.gauge_fix(p0, w2)
        out["sol_summary"] = np.array([sm.iterations, sm.successful_steps, sm.termination, sm.initial_cost, sm.final_cost])
        out["sol_trace"] = np.array(sm.cost_trace[:sm.iterations])
        for k, v in w2.state_copy().items():
            out["sol_" + k] = v
        mo = orc.marginalize(w2, abi.MARGIN_OLD)
        out["marg_meta"] = np.array([mo.c.n, mo.c.nblk, mo.c.m]); out["marg_A"], out["marg_b"] = mo.A_matrix(), mo.b_vector()
        out["marg_kind"], out["marg_index"], out["marg_col"] = mo.blk_kind[:mo.c.nblk].copy(), mo.blk_index[:mo.c.nblk].copy(), mo.blk_col[:mo.c.nblk].copy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "K=%d L=%d Fv=%d" % (w.K, w.L, len(w.vis_i)), "%.0f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
