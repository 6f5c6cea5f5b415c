"""The trust-region TRAJECTORY of the restated solver (oracle_solver.cpp), re-derived independently: Levenberg-regularised Gauss-Newton / Cauchy / traditional-dogleg
iterations as SURVEY.md Appendix B describes ceres' TrustRegionMinimizer + DoglegStrategy, written here from that description on the FULL (camera + landmark) normal
equations -- assembled in numpy from the oracle's raw per-factor residuals and Evaluate()-layout Jacobians (tests/numpy_ref.py: loss corrector, local columns) --
with every linear solve and every dogleg scalar in 50-digit mpmath.  No Schur complement, no shared line with oracle_solver.cpp: eliminating the landmarks first
(DENSE_SCHUR) is exact algebra on the same system, so the accepted / rejected decisions, the cost after every iteration and the trust-region radius must agree.
tests/test_oracle_solver_scipy.py pins the optimum; this pins how the solver gets there, including a REJECTED step (radius halved, the same Gauss-Newton and Cauchy
vectors blended again)."""
import numpy as np
import pytest

mp = pytest.importorskip("mpmath")

from mvil_fusion_amd import abi, synth
import numpy_ref as nr


def _perturbed_window(oracle):
    """configs[0]-shaped mini window (K = 5, 24 landmarks, no prior) with the orientations of frames 1.. disturbed by ~3 degrees: the fifth iteration's step is rejected."""
    w = synth.make_config(1, prior_fn=lambda pre: oracle.marginalize(pre).to_prior(), L=24)
    rng = np.random.default_rng(3)
    w.inv_depth[:] *= np.exp(rng.normal(0, 0.8, w.L))          # (draws of the exploration that fixed the case: kept so that the stream below is the same)
    w = synth.make_config(1, prior_fn=lambda pre: oracle.marginalize(pre).to_prior(), L=24)
    for k in range(1, w.K):
        q = synth.qmul(w.pose[k, 3:], np.concatenate([rng.normal(0, 0.05, 3), [1.0]]))
        w.pose[k, 3:] = q / np.linalg.norm(q)
    return w


def _index(w):
    idx = nr.camera_index(w)
    obs = np.bincount(w.vis_l, minlength=w.L)
    for l in range(w.L):
        if not w.lm_const[l] and obs[l] > 0:
            idx[("lam", l)] = w.D + l
    return idx


def _plus(w, base, idx, step):
    """x (+) step on a copy of the base state: translation additive, rotation q * [1, dtheta / 2] normalised (pose_local_parameterization.cpp:3-17), the rest additive."""
    w.set_state(base)
    for key, i0 in idx.items():
        if key[0] == "pose" or key[0] == "ex":
            tgt = w.pose[key[1]] if key[0] == "pose" else w.ex_pose.reshape(-1)
            d = step[i0:i0 + 6]
            tgt[:3] += d[:3]
            q = synth.qmul(tgt[3:7].copy(), np.array([0.5 * d[3], 0.5 * d[4], 0.5 * d[5], 1.0]))
            tgt[3:7] = q / np.linalg.norm(q)
        elif key[0] == "sb":
            w.speedbias[key[1]] += step[i0:i0 + 9]
        elif key[0] == "td":
            w.td[0] += step[i0]
        else:
            w.inv_depth[key[1]] += step[i0]


def _system(oracle, w, opts, idx, N):
    cost, H, g = nr.normal_equations(nr.factor_list(oracle, w, opts), idx)
    H = np.pad(H, ((0, N - H.shape[0]), (0, N - H.shape[1]))); g = np.pad(g, (0, N - g.shape[0]))
    return cost, H, g


def test_first_iterations_of_the_dogleg_loop_rederived_in_mpmath(oracle):
    mp.mp.dps = 50
    NIT = 6
    opts = abi.default_options(max_iterations=NIT, function_tolerance=0.0, parameter_tolerance=0.0, gradient_tolerance=0.0)
    wo = _perturbed_window(oracle)
    so = oracle.solve(wo, opts)
    assert so.iterations == NIT and so.termination == abi.TERM_NAMES.index("max_iterations")
    tr_o, rad_o = list(so.cost_trace)[:NIT], list(so.radius_trace)[:NIT]
    acc_o = [tr_o[0] < so.initial_cost] + [tr_o[i] < tr_o[i - 1] for i in range(1, NIT)]
    assert acc_o.count(False) >= 1 and acc_o[0], acc_o                     # the case holds a rejected step

    w = _perturbed_window(oracle)
    idx = _index(w)
    N = w.D + w.L
    live = np.zeros(N, bool)
    for key, i0 in idx.items():
        live[i0:i0 + {"pose": 6, "sb": 9, "ex": 6, "td": 1, "lam": 1}[key[0]]] = True
    cols = np.where(live)[0]
    n = len(cols)
    cost, H, g = _system(oracle, w, opts, idx, N)
    assert abs(cost - so.initial_cost) <= 1e-12 * so.initial_cost
    radius, mu, S = mp.mpf(opts.initial_radius), mp.mpf(opts.min_mu), None
    reuse, trace, radii, accepted = False, [], [], []
    gd = gn = alpha = d = None
    for it in range(NIT):
        if not reuse:
            Hm = mp.matrix(H[np.ix_(cols, cols)].tolist()); gm = mp.matrix(g[cols].tolist())
            if S is None:                                                   # Jacobi scaling, fixed at the first linearisation: 1 / (1 + |J_col|)
                S = [1 / (1 + mp.sqrt(Hm[i, i])) for i in range(n)]
            Hs = mp.matrix(n, n)
            for i in range(n):
                for j in range(n):
                    Hs[i, j] = S[i] * Hm[i, j] * S[j]
            gs = mp.matrix([S[i] * gm[i] for i in range(n)])
            d = [mp.sqrt(min(max(Hs[i, i], mp.mpf("1e-6")), mp.mpf("1e32"))) for i in range(n)]
            gd = [gs[i] / d[i] for i in range(n)]                           # gradient in dogleg space
            u = mp.matrix([gd[i] / d[i] for i in range(n)])
            alpha = sum(v * v for v in gd) / (u.T * Hs * u)[0]
            A = Hs.copy()
            for i in range(n):
                A[i, i] += mu * d[i] * d[i]
            x = mp.lu_solve(A, gs)                                          # (H_s + mu D^2) x = g_s ; Gauss-Newton step = -x (.) d in dogleg space
            mu = max(mp.mpf(opts.min_mu), 2 * mu / 10)
            gn = [-x[i] * d[i] for i in range(n)]
        gn_norm = mp.sqrt(sum(v * v for v in gn)); g_norm = mp.sqrt(sum(v * v for v in gd))
        if gn_norm <= radius:
            cg, cn, step_norm = mp.mpf(0), mp.mpf(1), gn_norm
        elif g_norm * alpha >= radius:
            cg, cn, step_norm = -(radius / g_norm), mp.mpf(0), radius
        else:                                                               # the point where the segment Cauchy -> Gauss-Newton leaves the trust region
            a = [-alpha * v for v in gd]
            ba = [gn[i] - a[i] for i in range(n)]
            a2, ba2, aba = sum(v * v for v in a), sum(v * v for v in ba), sum(a[i] * ba[i] for i in range(n))
            beta = (-aba + mp.sqrt(aba * aba + ba2 * (radius * radius - a2))) / ba2
            cg, cn, step_norm = -alpha * (1 - beta), beta, radius
        step = np.zeros(N)
        sm = [S[i] * (cg * gd[i] + cn * gn[i]) / d[i] for i in range(n)]
        step[cols] = [float(v) for v in sm]
        smv = mp.matrix(sm)
        model_change = -((smv.T * Hm * smv)[0] / 2 + sum(gm[i] * sm[i] for i in range(n)))
        assert model_change > 0
        base = w.state_copy()
        _plus(w, base, idx, step)
        cand_cost, Hc, gc = _system(oracle, w, opts, idx, N)
        rel = (mp.mpf(cost) - mp.mpf(cand_cost)) / model_change
        if rel > mp.mpf(opts.min_relative_decrease):
            cost, H, g = cand_cost, Hc, gc
            if rel < 0.25: radius = radius / 2
            if rel > 0.75: radius = max(radius, 3 * step_norm)
            radius = min(mp.mpf(opts.max_radius), radius)
            reuse = False; accepted.append(True)
        else:
            w.set_state(base)
            radius = radius / 2
            reuse = True; accepted.append(False)
        trace.append(cost); radii.append(float(radius))
    assert accepted == acc_o, (accepted, acc_o)
    # the radius the oracle records for iteration i is the one it WORKED with (before the update): compare shifted by one
    for i in range(NIT):
        assert abs(trace[i] - tr_o[i]) <= 1e-7 * max(1.0, abs(tr_o[i])), (i, trace[i], tr_o[i])
    for i in range(1, NIT):
        assert abs(radii[i - 1] - rad_o[i]) <= 1e-7 * rad_o[i], (i, radii, rad_o)
