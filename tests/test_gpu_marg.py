"""GPU marginalisation (vil_marginalize) vs the CPU oracle (MarginalizationInfo restatement).

The eigenvector basis of linearized_jacobians is implementation-defined in the reference
(SURVEY App. C #12), so parity is on the reduced information matrix A, vector b, the identities
J0^T J0 = A, J0^T r0 = b (marginalization_factor.cpp:313-314) and the block metadata.
"""
import numpy as np
import pytest

from mvil_fusion_amd import abi, synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def check(outg, outo, tol=1e-8):
    assert outg.c.n == outo.c.n and outg.c.nblk == outo.c.nblk and outg.c.m == outo.c.m
    nb = outo.c.nblk
    assert np.array_equal(outg.blk_kind[:nb], outo.blk_kind[:nb])
    assert np.array_equal(outg.blk_index[:nb], outo.blk_index[:nb])
    assert np.array_equal(outg.blk_col[:nb], outo.blk_col[:nb])
    pg, po = outg.to_prior(), outo.to_prior()
    assert np.array_equal(pg.x0, po.x0)
    Ag, Ao, bg, bo = outg.A_matrix(), outo.A_matrix(), outg.b_vector(), outo.b_vector()
    assert rel(Ag, Ao) < tol, rel(Ag, Ao)
    assert rel(bg, bo) < tol, rel(bg, bo)
    Jg = pg.J_matrix()
    As = 0.5 * (Ag + Ag.T)
    assert rel(Jg.T @ Jg, As) < 1e-7, rel(Jg.T @ Jg, As)          # eps-truncation of tiny eigenvalues only
    assert rel(Jg.T @ pg.r0, bg) < 1e-6, rel(Jg.T @ pg.r0, bg)
    Jo = po.J_matrix()
    assert rel(Jg.T @ Jg, Jo.T @ Jo) < 1e-7
    assert rel(Jg.T @ pg.r0, Jo.T @ po.r0) < 1e-6


@pytest.fixture(scope="module")
def wsolved(oracle):
    w = synth.make_config(2, L=150, n_plane=3000, n_edge=800, prior_fn=lambda pre: oracle.marginalize(pre).to_prior())
    p0 = w.pose[0].copy()
    oracle.solve(w)
    oracle.gauge_fix(p0, w)     # estimator.cpp:1419 double2vector precedes the marginalisation (App. C #9)
    return w


def test_margin_old_parity(hip, oracle, wsolved):
    check(hip.marginalize(wsolved, abi.MARGIN_OLD), oracle.marginalize(wsolved, abi.MARGIN_OLD))


def test_margin_old_with_icp_lps(hip, oracle, wsolved):
    w = wsolved
    icp = int(np.where(w.icp_ids[:, 0] == 0)[0][0])
    # make one LPS constraint touch frame 0 for this test
    w2 = synth.make_config(2, L=150, n_plane=0, n_edge=0)
    w2.set_state(w.state_copy()); w2.prior = w.prior
    w2.lps_ids = w2.lps_ids.copy(); w2.lps_ids[0] = [0, 1]
    w2.lps_const = w2.lps_const.copy(); w2.lps_const[0, 0:3] = [0.0, 0.1, 0.04]
    check(hip.marginalize(w2, abi.MARGIN_OLD, icp_marg=icp, lps_marg=0), oracle.marginalize(w2, abi.MARGIN_OLD, icp_marg=icp, lps_marg=0))


def test_margin_second_new_parity(hip, oracle, wsolved):
    og, oo = hip.marginalize(wsolved, abi.MARGIN_SECOND_NEW), oracle.marginalize(wsolved, abi.MARGIN_SECOND_NEW)
    check(og, oo)
    assert og.c.n == wsolved.prior.n - 6


def test_margin_without_prior(hip, oracle):
    w = synth.make_config(1)
    check(hip.marginalize(w, abi.MARGIN_OLD), oracle.marginalize(w, abi.MARGIN_OLD))
    og = hip.marginalize(w, abi.MARGIN_SECOND_NEW)
    assert og.c.n == -1       # no prior -> nothing to do (estimator.cpp:1620)


def test_prior_chain_solve(hip, oracle):
    """prior produced by the GPU feeds the next window's GPU solve and matches the oracle chain."""
    pfg = lambda pre: hip.marginalize(pre).to_prior()
    pfo = lambda pre: oracle.marginalize(pre).to_prior()
    wg = synth.make_config(2, L=150, n_plane=3000, n_edge=800, prior_fn=pfg)
    wo = synth.make_config(2, L=150, n_plane=3000, n_edge=800, prior_fn=pfo)
    sg, so = hip.solve(wg), oracle.solve(wo)
    assert sg.iterations == so.iterations
    assert abs(sg.final_cost - so.final_cost) <= 1e-7 * so.final_cost
    assert np.abs(wg.pose - wo.pose).max() < 1e-6


# ---- independent of the oracle: GPU factors -> numpy normal equations -> 60-digit Schur complement (tests/numpy_ref.py) ----------
import numpy_ref as nr


@pytest.mark.parametrize("flag", [abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW])
def test_marginal_equals_exact_schur_complement(hip, wsolved, flag):
    """vil_marginalize's A, b against the reference's rule (marginalization_factor.cpp:273-290) evaluated with 60-digit
    arithmetic on normal equations assembled in numpy from vil_eval_factors' own r / J -- nothing of the oracle is involved.
    The reference's fp64 eigen route only reaches ~1e-6 (diagonally scaled) on these matrices (cond(A_mm) ~ 1e7, cancellation
    against the 1e10 bias information); the library is asserted to sit inside that same floor."""
    opts = abi.default_options()
    ref = nr.marg_numpy(hip, wsolved, opts, flag, lidar=True)
    Ax, bx, _ = nr.exact_schur(ref["A_full"], ref["b_full"], ref["m"])
    out = hip.marginalize(wsolved, flag)
    assert out.c.n == Ax.shape[0] and out.c.m == ref["m"]
    e, floor = nr.scaled_err(out.A_matrix(), Ax), max(nr.scaled_err(ref["A"], Ax), 1e-9)
    print("GPU marginal vs exact: %.2e (numpy fp64 eigen route: %.2e)" % (e, floor))
    assert e <= 5e-6 and e <= 20 * floor, (e, floor)          # inside the noise floor of the reference's own fp64 algorithm
    assert np.abs(out.b_vector() - bx).max() <= 1e-5 * np.abs(bx).max()


def test_lidar_factors_of_dropped_pose_reach_the_prior(hip, oracle, wsolved):
    """Extended mode: the edge / plane point factors attached to frame 0 are part of the marginalised information
    (marginalization_factor.cpp:176-316 folds every factor touching a dropped block)."""
    w = wsolved
    assert (w.plane_pose == 0).sum() > 0 and (w.edge_pose == 0).sum() > 0
    w0 = synth.make_config(2, L=150, n_plane=0, n_edge=0)
    w0.set_state(w.state_copy()); w0.prior = w.prior
    a_with, a_without = hip.marginalize(w, abi.MARGIN_OLD).A_matrix(), hip.marginalize(w0, abi.MARGIN_OLD).A_matrix()
    assert nr.scaled_err(a_with, a_without) > 1e-3            # they carry real information about the kept poses
    check(hip.marginalize(w, abi.MARGIN_OLD), oracle.marginalize(w, abi.MARGIN_OLD))


def test_rank_deficient_landmark_follows_reference_rule(hip, oracle):
    """Landmarks of frame 0 without parallax (h_ll <= eps = 1e-8): the reference's pseudo inverse zeroes those directions
    (marginalization_factor.cpp:277); the library drops the same pivots (k_sweep, lin_mode 2).  Compared with the rule
    evaluated at 60 digits, and shown to differ materially from the plain inverse."""
    w, weak = nr.rank_deficient_window(oracle)
    opts = abi.default_options()
    ref = nr.marg_numpy(hip, w, opts, abi.MARGIN_OLD, lidar=True)
    Ax, bx, E = nr.exact_schur(ref["A_full"], ref["b_full"], ref["m"])
    assert (E <= 1e-8).sum() == len(weak) >= 1
    out = hip.marginalize(w, abi.MARGIN_OLD)
    e = nr.scaled_err(out.A_matrix(), Ax)
    assert e <= 1e-6, e
    A0, _, _ = nr.exact_schur(ref["A_full"], ref["b_full"], ref["m"], eps=0.0)
    assert nr.scaled_err(A0, Ax) > 1e-3
