"""Config 5 (synthetic replay): the HIP library drives the per-image chain solve -> gauge fix -> marginalise -> slide;
on every frame the CPU oracle is given the SAME input window and must produce the same solved state and the same new
prior (information form).  fp64 tolerances of DESIGN.md section 2."""
import numpy as np
import pytest

from mvil_fusion_amd import abi, replay
from mvil_fusion_amd.abi import Window

pytestmark = pytest.mark.gpu


def _rot_angle(qa, qb):
    d = abs(float(np.dot(qa, qb)) / (np.linalg.norm(qa) * np.linalg.norm(qb)))
    return 2.0 * np.arccos(min(1.0, d))


def test_replay_lockstep(hip, oracle):
    K = 8
    rp = replay.Replay(K=K, n_frames=40, L=150, n_plane=2400, n_edge=800, seed=21, second_new_every=4)
    flags = set()
    for step in range(12):
        w = rp.window()
        flag = rp.margin_flag(); flags.add(flag)
        wo = Window.from_dict(w.to_dict())
        p0 = w.pose[0].copy()
        sg = hip.solve(w, rp.opts); hip.gauge_fix(p0, w)
        so = oracle.solve(wo, rp.opts); oracle.gauge_fix(p0, wo)
        assert sg.iterations == so.iterations and sg.termination == so.termination
        assert abs(sg.final_cost - so.final_cost) <= 1e-7 * max(1.0, abs(so.final_cost)) * (100.0 if w.prior.n == 0 else 1.0)
        assert np.abs(w.pose[:, :3] - wo.pose[:, :3]).max() < 1e-6
        assert max(_rot_angle(w.pose[k, 3:], wo.pose[k, 3:]) for k in range(K)) < 1e-7
        assert np.abs(w.speedbias - wo.speedbias).max() < 1e-6
        assert np.abs(w.inv_depth - wo.inv_depth).max() < 1e-6
        pg = hip.marginalize(w, flag, w._icp_marg, w._lps_marg, rp.opts)
        po = oracle.marginalize(wo, flag, w._icp_marg, w._lps_marg, rp.opts)
        assert pg.c.n == po.c.n and pg.c.nblk == po.c.nblk
        if pg.c.n > 0:
            nb = pg.c.nblk
            assert np.array_equal(pg.blk_kind[:nb], po.blk_kind[:nb]) and np.array_equal(pg.blk_index[:nb], po.blk_index[:nb])
            Ag, Ao = pg.A_matrix(), po.A_matrix()
            sc = np.sqrt(np.outer(np.abs(np.diag(Ao)) + 1e-300, np.abs(np.diag(Ao)) + 1e-300))
            assert (np.abs(Ag - Ao) / sc).max() < 2e-5          # states agree to ~1e-9; the pseudo-inverse of the dropped block amplifies that
            Jg = pg.to_prior().J_matrix()
            assert np.abs(Jg.T @ Jg - 0.5 * (Ag + Ag.T)).max() < 1e-8 * np.abs(Ag).max()   # J0^T J0 = A (marginalization_factor.cpp:313), normwise: pivots below 1e-10 max(diag) are noise and dropped
        assert rp.absorb(w, pg, flag)
    assert flags == {abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW}
