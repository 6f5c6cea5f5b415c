"""An independent optimiser on the same cost: scipy.optimize.least_squares (trust-region-reflective, finite-difference Jacobian)
minimises 1/2 sum rho(|r_f|^2) built from the oracle's RAW per-factor residuals, on small windows with and without a prior.
The restated Ceres-style dogleg loop (oracle_solver.cpp) must reach the same minimum value -- a pin of the solver restatement
that does not share a line with it (the trajectory through the iterations is implementation-defined and is not compared)."""
import numpy as np
import pytest
from scipy.optimize import least_squares

from mvil_fusion_amd import abi, synth


def _plus_pose(p7, d6):
    out = p7.copy()
    out[:3] += d6[:3]
    q = synth.qmul(p7[3:7], np.array([0.5 * d6[3], 0.5 * d6[4], 0.5 * d6[5], 1.0]))
    out[3:7] = q / np.linalg.norm(q)
    return out


class Problem:
    def __init__(self, oracle, w):
        self.o, self.w = oracle, w
        self.base = {k: v.copy() for k, v in w.state_copy().items()}
        K, L = w.K, w.L
        self.nfull = 6 * K + 9 * K + 6 + 1 + L
        # parameter blocks the problem holds constant (SetParameterBlockConstant, estimator.cpp:1154-1166, 1217-1221, 1354-1370) stay put
        free = np.ones(self.nfull, bool)
        for k in range(K):
            if w.pose_const[k]: free[6 * k:6 * k + 6] = False
            if w.sb_const[k]: free[6 * K + 9 * k:6 * K + 9 * k + 9] = False
        if w.ex_const: free[15 * K:15 * K + 6] = False
        if w.td_const or not w.use_td: free[15 * K + 6] = False
        free[15 * K + 7:] = ~np.asarray(w.lm_const, bool)
        self.free = np.where(free)[0]
        self.n = len(self.free)
        self.classes = [c for c in (abi.FACTOR_IMU, abi.FACTOR_VISUAL, abi.FACTOR_PRIOR, abi.FACTOR_ICP, abi.FACTOR_LPS, abi.FACTOR_EDGE, abi.FACTOR_PLANE) if w.eval_sizes(c)[0] > 0]

    def set_state(self, dfree):
        d = np.zeros(self.nfull); d[self.free] = dfree
        w, b, K, L = self.w, self.base, self.w.K, self.w.L
        for k in range(K):
            w.pose[k] = _plus_pose(b["pose"][k], d[6 * k:6 * k + 6])
        w.speedbias[:] = b["speedbias"] + d[6 * K:15 * K].reshape(K, 9)
        w.ex_pose[:] = _plus_pose(b["ex_pose"].ravel(), d[15 * K:15 * K + 6]).reshape(b["ex_pose"].shape)
        w.td[:] = b["td"] + d[15 * K + 6]
        w.inv_depth[:] = b["inv_depth"] + d[15 * K + 7:]

    def residual(self, d):
        self.set_state(d)
        out = []
        for c in self.classes:
            r, _ = self.o.eval_factors(self.w, c, jac=False)
            nr = abi.NR.get(c)
            if c == abi.FACTOR_VISUAL or c in (abi.FACTOR_ICP, abi.FACTOR_LPS):            # CauchyLoss(1.0): rho(s) = log(1 + s)
                r = r.reshape(-1, nr); s = (r * r).sum(axis=1)
                r = (r * np.sqrt(np.where(s > 0, np.log1p(s) / np.maximum(s, 1e-300), 1.0))[:, None]).ravel()
            elif c in (abi.FACTOR_EDGE, abi.FACTOR_PLANE):                                 # HuberLoss(0.1)
                r = r.reshape(-1, nr); s = (r * r).sum(axis=1); a = 0.1
                rho = np.where(s > a * a, 2 * a * np.sqrt(np.maximum(s, 1e-300)) - a * a, s)
                r = (r * np.sqrt(np.where(s > 0, rho / np.maximum(s, 1e-300), 1.0))[:, None]).ravel()
            out.append(r)
        return np.concatenate(out)


@pytest.mark.parametrize("cid,kw", [(1, dict(L=24)), (2, dict(L=30, n_plane=120, n_edge=40))])
def test_dogleg_restatement_stops_at_a_minimum_scipy_agrees_with(oracle, cid, kw):
    pf = lambda pre: oracle.marginalize(pre).to_prior()
    w = synth.make_config(cid, prior_fn=pf, **kw)
    # the prior-less window has a flat valley (4 gauge directions held only by mu): the dogleg loop creeps along it, so give it
    # iterations; termination tolerances off
    # autodiff_quirk = 0: the ICP / LPS factors then use the true tangent Jacobian.  (The reference pairs AutoDiff over the raw
    # quaternion coordinates with VINS' identity-like local parameterisation; Gauss-Newton with that Jacobian settles a hair
    # away from the minimum -- 0.05 % in cost on this window -- which the default options reproduce on purpose.)
    opts = abi.default_options(max_iterations=2000, function_tolerance=1e-16, parameter_tolerance=1e-16, gradient_tolerance=1e-16, autodiff_quirk=0)
    # the same cost function on both sides, at the initial state ...
    pr0 = Problem(oracle, w)
    r0 = pr0.residual(np.zeros(pr0.n)); pr0.set_state(np.zeros(pr0.n))
    s = oracle.solve(w, opts)                                                     # w now holds the restated solver's answer
    assert abs(0.5 * r0 @ r0 - s.initial_cost) <= 1e-9 * s.initial_cost
    # ... and at the solution
    pr = Problem(oracle, w)
    rs = pr.residual(np.zeros(pr.n))
    assert abs(0.5 * rs @ rs - s.final_cost) <= 1e-9 * s.final_cost
    # (i) an independent trust-region optimiser started AT that answer cannot improve on it
    kw_ls = dict(method="trf", jac="3-point", x_scale="jac", xtol=1e-14, ftol=1e-14, gtol=1e-12, max_nfev=80)
    sol = least_squares(pr.residual, np.zeros(pr.n), **kw_ls)
    assert sol.cost >= s.final_cost * (1 - 1e-4) and sol.cost <= s.final_cost * (1 + 1e-12), (sol.cost, s.final_cost)
    # (ii) started from a perturbed state it comes back to the same cost
    rng = np.random.default_rng(0)
    d0 = (rng.normal(0, 1.0, pr.nfull) * np.concatenate([np.tile([2e-3] * 3 + [5e-4] * 3, w.K), np.tile([2e-3] * 3 + [1e-4] * 6, w.K), [1e-4] * 6, [1e-5], [2e-3] * w.L]))[pr.free]
    assert 0.5 * pr.residual(d0) @ pr.residual(d0) > 1.5 * s.final_cost
    sol2 = least_squares(pr.residual, d0, **dict(kw_ls, max_nfev=300))
    assert abs(sol2.cost - s.final_cost) <= 2e-4 * s.final_cost, (sol2.cost, s.final_cost, sol2.nfev, sol2.status)
